import sys, os, torch, time
sys.path.insert(0, "/root/repo")
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
B = int(sys.argv[1])
wav = 0.1 * torch.randn(B, 64000, device=dev, generator=g)
ref = torch.cat([m(wav[i:i + 256], is_eval=True)[1] for i in range(0, B, 256)])
torch.cuda.synchronize()
t0 = time.perf_counter()
big = m(wav, is_eval=True)[1]
torch.cuda.synchronize()
print("B", B, "ms", (time.perf_counter() - t0) * 1e3, "equal", bool(torch.equal(big, ref)), "maxdiff", float((big - ref).abs().max()), flush=True)
