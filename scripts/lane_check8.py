"""[needs the SIDEKIT_AMD_STOP_STAGE / SIDEKIT_AMD_STOP_LAUNCH hooks that commit afd6222 carried in xt_api.hip; removed afterwards] Diagnostic: does the truncated bf16 trunk, run ALONE, change memory it does not own (torch tensors allocated around it)?"""
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
feats_a = m1.features(a)
m1.compute_dtype = "bf16"
for w in range(2): m1.forward_features(feats_a)
ref = m2.features(b)
torch.cuda.synchronize()
# 1) aggressor alone: does anything the victim owns change?
f = m2.features(b); torch.cuda.synchronize()
fc, bc, refc = f.clone(), b.clone(), ref.clone()
for _ in range(5): m1.forward_features(feats_a)
torch.cuda.synchronize()
print("aggressor alone: victim output changed:", int((f != fc).sum()), " victim input changed:", int((b != bc).sum()), " ref changed:", int((ref != refc).sum()), flush=True)
f2 = m2.features(b); torch.cuda.synchronize()
print("victim re-run alone after the aggressor: differs from ref:", int((f2 != ref).sum()), flush=True)
# 2) concurrent, but the victim's three kernels separated: is it the FFT+cmvn phase or the copy-out phase?
s2 = torch.cuda.Stream()
for trial in range(4):
    m1.forward_features(feats_a)
    with torch.cuda.stream(s2):
        fa = m2.features(b)
    torch.cuda.synchronize()
    fb = m2.features(b); torch.cuda.synchronize()        # alone again, right after
    print(f"trial {trial}: concurrent run differs from ref in {int((fa != ref).sum())} elements; solo run right after: {int((fb != ref).sum())}", flush=True)
