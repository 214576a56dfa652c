"""Diagnostic: does a kernel of one stream change the RESULT of the front-end running on another stream? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIDEKIT_AMD_LANES"] = "1"
import torch
from sidekit_amd.nnet import Xtractor

dev = torch.device("cuda", 0)
m1 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m2 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
g = torch.Generator(device="cuda").manual_seed(0)
wav = 0.1 * torch.randn(256, 64000, device="cuda", generator=g)
a, b = wav[:128].contiguous(), wav[128:].contiguous()
s2 = torch.cuda.Stream()
ref_feat = m2.features(b)
m1.compute_dtype = "bf16"
feats_a = m1.features(a)
torch.cuda.synchronize()

def aggressor(kind):
    if kind == "bf16 forward":
        m1.compute_dtype = "bf16"; m1(a, is_eval=True)
    elif kind == "fp32 forward":
        m1.compute_dtype = "fp32"; m1(a, is_eval=True)
    elif kind == "features only":
        m1.features(a)
    elif kind == "bf16 trunk only":
        m1.compute_dtype = "bf16"; m1.forward_features(feats_a)
    elif kind == "fp32 trunk only":
        m1.compute_dtype = "fp32"; m1.forward_features(feats_a)

for kind in ("bf16 forward", "fp32 forward", "features only", "bf16 trunk only", "fp32 trunk only", "nothing"):
    for w in range(2): aggressor(kind)
    torch.cuda.synchronize()
    nbad = 0; worst = 0.0
    for trial in range(10):
        aggressor(kind)                       # main stream
        with torch.cuda.stream(s2):
            f = m2.features(b)                # side stream, beside it
        torch.cuda.synchronize()
        d = (f - ref_feat).abs()
        nbad += int((d.amax(dim=(1, 2)) > 0).sum()); worst = max(worst, float(d.max()))
    print(f"aggressor = {kind:16s}: utterances whose features differed from the solo run: {nbad} of 1280, max abs diff {worst:.3e}", flush=True)
