"""[needs the SIDEKIT_AMD_STOP_STAGE / SIDEKIT_AMD_STOP_LAUNCH hooks that commit afd6222 carried in xt_api.hip; removed afterwards] Diagnostic: the bf16 trunk truncated at SIDEKIT_AMD_STOP_STAGE beside the front-end of another model on another stream."""
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
feats_a = m1.features(a)
m1.compute_dtype = "bf16"
for w in range(2): m1.forward_features(feats_a)
torch.cuda.synchronize()
nbad = 0; worst = 0.0
for trial in range(10):
    m1.forward_features(feats_a)
    with torch.cuda.stream(s2):
        f = m2.features(b)
    torch.cuda.synchronize()
    d = (f - ref_feat).abs()
    nbad += int((d.amax(dim=(1, 2)) > 0).sum()); worst = max(worst, float(d.max()))
tag = "stage %s launch %s" % (os.environ.get("SIDEKIT_AMD_STOP_STAGE", "-"), os.environ.get("SIDEKIT_AMD_STOP_LAUNCH", "-"))
print(f"stop {tag}: utterances whose features differed: {nbad} of 1280, max abs diff {worst:.3e}", flush=True)
# pattern of the last differing utterance: which (mel, frame) elements differ, how much
d = (f - ref_feat).abs()
bad = torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten().tolist()
for u in bad[:4]:
    du = d[u]                                  # (80, T)
    mels = torch.nonzero(du.amax(dim=1) > 0).flatten().tolist()
    frames = torch.nonzero(du.amax(dim=0) > 1e-2).flatten().tolist()
    print(f"  utt {u}: mel rows touched {len(mels)} {mels[:10]}, frames with |diff| > 1e-2: {len(frames)} {frames[:12]}, max {float(du.max()):.3f}", flush=True)
    if frames:
        t = frames[0]
        print("    frame", t, "diff over mels:", [round(float(x), 2) for x in du[:, t][:16]], "...", flush=True)
