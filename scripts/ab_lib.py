"""Same-box A/B of two builds of the library on the benchmark batch (GPU box): `python scripts/ab_lib.py [other/libsidekit_amd.so]`,
best of 5 x 20 steps.  Boxes differ by +-3 %, a change worth keeping must win on ONE box."""
import sys, os, time, ctypes
sys.path.insert(0, "/root/repo")
import torch
from sidekit_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
wavs = [0.1 * torch.randn(256, 64000, device=dev, generator=g) for _ in range(5)]
for _ in range(5): m(wavs[0], is_eval=True)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for i in range(20): m(wavs[i % 5], is_eval=True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
print(sys.argv[1:] or "HEAD", "best of 5 x 20 steps: %.3f ms" % best, flush=True)
