import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
for B in (256, 384, 512):
    wavs = [0.1 * torch.randn(B, 64000, device=dev, generator=g) for _ in range(3)]
    for lanes in (1, 2, 3, 4):
        m.set_lanes(lanes)
        for _ in range(3): m(wavs[0], is_eval=True)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(10): m(wavs[i % 3], is_eval=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        print(f"B={B} lanes={lanes}: {best:.3f} ms = {B / best:.2f} k x-vec/s", flush=True)
    del wavs
