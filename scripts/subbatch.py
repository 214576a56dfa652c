"""Does a batch small enough for its activations to live in the 256-MB Infinity Cache run the HBM-bound layers faster?
ms per 256 utterances when the benchmark batch is forwarded as 256 / B sequential sub-batches of B (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
wav = 0.1 * torch.randn(256, 64000, device=dev, generator=g)
for B in (256, 128, 64, 32, 16):
    parts = [wav[i:i + B].contiguous() for i in range(0, 256, B)]
    def run():
        for p in parts: m(p, is_eval=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    m.set_profile(True)
    run(); torch.cuda.synchronize()
    prof = m.get_profile()
    m.set_profile(False)
    top = {k: round(v[0], 3) for k, v in prof.items() if v[0] > 0.05} if isinstance(prof, dict) else prof
    print(f"B={B:4d}: {ms:.3f} ms per 256   {top}", flush=True)
