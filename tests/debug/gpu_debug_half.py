"""Stage-wise parity of the HalfResNet34 path against the oracle (debug helper, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from sidekit_amd.nnet import Xtractor
from sidekit_amd.nnet.weights import seeded_state_dict
from oracle import xvector as oxv

def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return ((a - b).norm() / b.norm()).item()

def bf16_bytes_to_f32(buf):
    u = buf.view(numpy.uint16).astype(numpy.uint32) << 16
    return torch.from_numpy(u.view(numpy.float32).copy())

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 51)
n_spk = 16
sd = seeded_state_dict("halfresnet34", n_spk, seed=1234)
m = Xtractor(n_spk, "halfresnet34", "aam", seed=1234).to("cuda").eval()
m.compute_dtype = dtype
g = torch.Generator().manual_seed(11)
feats = torch.randn(B, 80, T, generator=g)
taps = {}
with torch.no_grad():
    o_logits, o_emb = oxv.halfresnet34_from_feats(feats, sd, taps=taps)
m.set_debug(True)
logits, emb = m.forward_features(feats.cuda())
torch.cuda.synchronize()
names = ["stem", "layer1", "layer2", "layer3", "layer4", "pooled", "pre_norm"]
raw = m.debug_taps(names)
H = T
for li, n in enumerate(["stem", "layer1", "layer2", "layer3", "layer4"]):
    ref = taps[n]  # (B,C,H,W)
    Bc, C, Hh, W = ref.shape
    buf = raw[n]
    x = bf16_bytes_to_f32(buf) if dtype == "bf16" else torch.from_numpy(buf.view(numpy.float32).copy())
    x = x.reshape(Bc, Hh, W, C).permute(0, 3, 1, 2)
    print(f"{n:8s} rel={rel(x, ref):.3e} max|ref|={ref.abs().max():.3f}")
pooled = torch.from_numpy(raw["pooled"].view(numpy.float32).copy()).reshape(B, 2, 10, 256).permute(0, 1, 3, 2).reshape(B, 5120)
print(f"pooled   rel={rel(pooled, taps['pooled']):.3e}")
pre = torch.from_numpy(raw["pre_norm"].view(numpy.float32).copy()).reshape(B, 256)
print(f"pre_norm rel={rel(pre, taps['pre_norm']):.3e}")
print(f"emb      rel={rel(emb.cpu(), o_emb):.3e}")
print(f"logits   rel={rel(logits.cpu(), o_logits):.3e}")
