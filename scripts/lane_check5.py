"""Diagnostic: single convolution shapes (sk_bench_conv, default stream) beside the front-end of a model on a side stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIDEKIT_AMD_LANES"] = "1"
import torch
from sidekit_amd.nnet import Xtractor
from sidekit_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda", 0)
m2 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
g = torch.Generator(device="cuda").manual_seed(0)
b = 0.1 * torch.randn(128, 64000, device="cuda", generator=g)
s2 = torch.cuda.Stream()
ref_feat = m2.features(b)
torch.cuda.synchronize()
Ts = {0: 401, 2: 401, 4: 201, 7: 101, 10: 51, 11: 401}
for shape, variant in ((0, 0), (0, 8), (0, 16), (11, 8), (2, 8), (4, 8), (4, 16), (7, 8), (10, 8)):
    nbad = 0; worst = 0.0
    for trial in range(6):
        with torch.cuda.stream(s2):
            fs = [m2.features(b) for _ in range(4)]
        ms = ctypes.c_float(0)
        _lib.check(lib.sk_bench_conv(shape, 1, 128, Ts[shape], 6, variant, ctypes.byref(ms), None))
        torch.cuda.synchronize()
        for f in fs:
            d = (f - ref_feat).abs()
            nbad += int((d.amax(dim=(1, 2)) > 0).sum()); worst = max(worst, float(d.max()))
    print(f"aggressor conv shape {shape} variant {variant}: utterances whose features differed: {nbad} of {6 * 4 * 128}, max abs diff {worst:.3e}", flush=True)
