"""Experiment (GPU box): does running the benchmark batch as sub-batches on several HIP streams overlap the HBM-bound
and the MFMA-bound kernel classes?  Prints ms per 256 utterances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor

dev = torch.device("cuda", 0)
def mk():
    m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
    m.compute_dtype = "bf16"
    return m
L = 64000
g = torch.Generator(device=dev).manual_seed(0)
wav = 0.1 * torch.randn(512, L, device=dev, generator=g)
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
m0 = mk()
print("one stream, B=256: %.3f ms per 256" % timeit(lambda: m0(wav[:256], is_eval=True)), flush=True)
def multi(nstreams, per):
    ms = [mk() for _ in range(nstreams)]
    ss = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    ws = [wav[i * per:(i + 1) * per].contiguous() for i in range(nstreams)]
    def run():
        for k in range(nstreams):
            with torch.cuda.stream(ss[k]): ms[k](ws[k], is_eval=True)
    t = timeit(run)
    print("%d streams x B=%d: %.3f ms per 256" % (nstreams, per, t * 256 / (nstreams * per)), flush=True)
for ns, per in ((2, 128), (2, 256), (3, 86), (2, 192), (4, 128)):
    multi(ns, per)
