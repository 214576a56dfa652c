"""Experiment: batch-1 forward replayed from a HIP graph (captured through torch.cuda.graph) against the eager forward (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
for dt in ("bf16", "fp32"):
    m.compute_dtype = dt
    for B in (1, 8):
        wav = 0.1 * torch.randn(B, 64000, device=dev)
        for _ in range(3): ref = m(wav, is_eval=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(wav, is_eval=True)
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(out[1], ref[1]) and torch.equal(out[0], ref[0]))
        n = 300
        t0 = time.perf_counter()
        for _ in range(n): m(wav, is_eval=True)
        torch.cuda.synchronize(); e = (time.perf_counter() - t0) / n * 1e3
        t0 = time.perf_counter()
        for _ in range(n): g.replay()
        torch.cuda.synchronize(); r = (time.perf_counter() - t0) / n * 1e3
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay(); torch.cuda.synchronize()
        rs = (time.perf_counter() - t0) / n * 1e3
        print(f"{dt} B={B}: eager {e:.3f} ms/call, graph replay {r:.3f} ms/call queued, {rs:.3f} synchronised, bit-identical: {same}", flush=True)
