"""Regression probe: the front-end on a side stream beside (a) the whole bf16 forward, (b) the truncated trunk; many trials (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIDEKIT_AMD_LANES"] = "1"
import torch
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m1 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m2 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m3 = Xtractor(64, model_archi="xvector", loss="aam", seed=4321).to(dev).eval()
g = torch.Generator(device="cuda").manual_seed(0)
wav = 0.1 * torch.randn(256, 64000, device="cuda", generator=g)
a, b = wav[:128].contiguous(), wav[128:].contiguous()
ref2, ref3 = m2.features(b), m3.features(b)
feats_a = m1.features(a)
m1.compute_dtype = "bf16"
for w in range(2): m1.forward_features(feats_a); m1(a, is_eval=True)
torch.cuda.synchronize()
s2 = torch.cuda.Stream()
for name, aggr in (("bf16 trunk", lambda: m1.forward_features(feats_a)), ("bf16 forward", lambda: m1(a, is_eval=True))):
    bad2 = bad3 = 0
    for trial in range(30):
        aggr()
        with torch.cuda.stream(s2):
            f2 = m2.features(b); f3 = m3.features(b)
        torch.cuda.synchronize()
        bad2 += int(((f2 - ref2).abs().amax(dim=(1, 2)) > 0).sum()); bad3 += int(((f3 - ref3).abs().amax(dim=(1, 2)) > 0).sum())
    print(f"beside the {name}: utterances whose features differed over 30 trials: log-mel front-end {bad2}, MFCC front-end {bad3} (of 3840 each)", flush=True)
for model, x in ((m2, b), (m3, b)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): model.features(x)
    torch.cuda.synchronize()
    print(f"{model.model_archi} front-end alone: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per 128 x 4 s", flush=True)
