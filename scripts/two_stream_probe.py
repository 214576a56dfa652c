"""Probe: does running consecutive batches on two streams (two handles) raise throughput?  (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor
B, L, K = 256, 64000, 20
g = torch.Generator(device="cuda").manual_seed(0)
wav = [0.1 * torch.randn(B, L, device="cuda", generator=g) for _ in range(2)]
models = [Xtractor(7205, "halfresnet34", "aam", seed=1).to("cuda").eval() for _ in range(2)]
for m in models:
    m.compute_dtype = "bf16"
streams = [torch.cuda.Stream() for _ in range(2)]
def run(n_streams):
    for i in range(4):
        with torch.cuda.stream(streams[i % n_streams]):
            models[i % n_streams](wav[i % 2], is_eval=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % n_streams]):
            models[i % n_streams](wav[i % 2], is_eval=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * K / dt
for n in (1, 2, 1, 2):
    print(n, "stream(s):", round(run(n)), "x-vec/s", flush=True)
