"""Front-end alone (xt_features: STFT -> power -> mel -> log [-> DCT] -> CMVN) at the bench shapes: time per batch and a digest /
dump of the features, to A/B two builds.

usage: python scripts/frontend_bench.py [halfresnet34|xvector] [dump.npy]     (GPU box)
With a dump path: if the file exists the features are compared with it (max abs / rel difference), else it is written.
"""
import os
import sys

import numpy
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sidekit_amd import _lib   # noqa: E402
if os.environ.get("SK_LIB"):    # A/B against another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["SK_LIB"])
from sidekit_amd.nnet.xvector import Xtractor   # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "halfresnet34"
dump = sys.argv[2] if len(sys.argv) > 2 else None
B, L = (256, 64000) if arch == "halfresnet34" else (512, 96000)
dev = torch.device("cuda:0")
m = Xtractor(100, model_archi=arch, loss="aam" if arch == "halfresnet34" else "cce", seed=0).to(dev).eval()
g = torch.Generator().manual_seed(0)
x = (0.1 * torch.randn(B, L, generator=g)).to(dev)
lens = [L] * B
lens[1], lens[2], lens[3] = L - 777, L // 2 + 1, 16000         # ragged rows exercise the reflect / beyond-the-end paths
f = m.features(x, lengths=lens)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    m.features(x, lengths=lens)
ev[0].record()
N = 20
for _ in range(N):
    m.features(x, lengths=lens)
ev[1].record()
torch.cuda.synchronize()
fa = f.cpu().numpy()
print(f"{arch} features B={B} L={L}: {ev[0].elapsed_time(ev[1]) / N * 1e3:.1f} us per batch  "
      f"sum={fa.astype(numpy.float64).sum():.6f} abs={numpy.abs(fa).astype(numpy.float64).sum():.3f} finite={bool(numpy.isfinite(fa).all())}")
if dump:
    if os.path.exists(dump):
        ref = numpy.load(dump)
        d = numpy.abs(fa - ref)
        print(f"  vs {dump}: max abs diff {d.max():.3e}, mean abs diff {d.mean():.3e}, identical {bool((fa == ref).all())}")
    else:
        numpy.save(dump, fa)
