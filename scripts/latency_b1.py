"""Batch-1 latency of Xtractor.forward (the reference's extract_xvectors.py calls the model one file at a time): wall time per
call with the queue kept full, and per call when the caller synchronises after each (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd import _lib
if os.environ.get("SK_LIB"):    # A/B against another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["SK_LIB"])
from sidekit_amd.nnet import Xtractor

dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
for dt in ("bf16", "fp32"):
    m.compute_dtype = dt
    for B, sec in ((1, 4), (1, 10), (8, 4)):
        wav = 0.1 * torch.randn(B, sec * 16000, device=dev)
        for _ in range(5): m(wav, is_eval=True)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n): m(wav, is_eval=True)
        torch.cuda.synchronize()
        q = (time.perf_counter() - t0) / n * 1e3
        t0 = time.perf_counter()
        for _ in range(n):
            m(wav, is_eval=True)
            torch.cuda.synchronize()
        s = (time.perf_counter() - t0) / n * 1e3
        print(f"{dt} B={B} {sec}s: {q:.3f} ms/call queued, {s:.3f} ms/call synchronised", flush=True)
