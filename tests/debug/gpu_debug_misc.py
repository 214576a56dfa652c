"""Front-end + TDNN + ragged-batch parity against the oracle (debug helper, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from sidekit_amd.nnet import Xtractor
from oracle import xvector as oxv, frontend as ofe

def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return ((a - b).norm() / b.norm()).item()

torch.manual_seed(0)
# ---- mel front-end
m = Xtractor(16, "halfresnet34", "aam", seed=1234).to("cuda").eval()
m.compute_dtype = "fp32"
sd = m.state_dict()
wav = 0.1 * torch.randn(3, 16000 * 2 + 77)
f_ref = ofe.melspec_frontend(wav)
f = m.features(wav.cuda()).cpu()
print("melspec feats rel", rel(f, f_ref), "max abs", (f - f_ref).abs().max().item())
f64 = ofe.melspec_frontend(wav.double())
print("  oracle f32 vs f64", rel(f_ref, f64), " gpu vs f64", rel(f, f64))
lo, e = m(wav.cuda(), is_eval=True)
with torch.no_grad():
    rlo, re = oxv.halfresnet34_forward(wav, sd)
print("half wav->emb rel", rel(e.cpu(), re), "logits", rel(lo.cpu(), rlo))
# ragged batch
lens = [16000 * 2 + 77, 20000, 9000]
with torch.no_grad():
    _, rr = oxv.forward_ragged([wav[i, :lens[i]] for i in range(3)], sd)
_, er = m(wav.cuda(), is_eval=True, lengths=lens)
print("half ragged rel per utt", [rel(er[i].cpu(), rr[i]) for i in range(3)])
# ---- TDNN
t = Xtractor(16, "xvector", "aam", seed=4321).to("cuda").eval()
sdt = t.state_dict()
wav = 0.1 * torch.randn(3, 16000 * 4)
ft_ref = ofe.mfcc_frontend(wav)
ft = t.features(wav.cuda()).cpu()
print("mfcc feats rel", rel(ft, ft_ref))
g = torch.Generator().manual_seed(21)
feats = torch.randn(2, 80, 63, generator=g)
taps = {}
with torch.no_grad():
    rl, remb = oxv.tdnn_from_feats(feats, sdt, "aam", taps=taps)
t.set_debug(True)
l, emb = t.forward_features(feats.cuda())
print("tdnn feats->emb rel", rel(emb.cpu(), remb), "logits", rel(l.cpu(), rl))
raw = t.debug_taps(["conv1", "conv5", "pooled", "pre_norm"])
c1 = torch.from_numpy(raw["conv1"].view(numpy.float32).copy()).reshape(2, 63, 512)[:, :59].permute(0, 2, 1)
print("  conv1 rel", rel(c1, taps["conv1"]))
pooled = torch.from_numpy(raw["pooled"].view(numpy.float32).copy()).reshape(2, 3072)
print("  pooled rel", rel(pooled, taps["pooled"]))
with torch.no_grad():
    rl2, remb2 = oxv.tdnn_forward(wav, sdt)
l2, emb2 = t(wav.cuda(), is_eval=True)
print("tdnn wav->emb rel", rel(emb2.cpu(), remb2))
lens = [64000, 40000, 33000]
with torch.no_grad():
    _, rr = oxv.forward_ragged([wav[i, :lens[i]] for i in range(3)], sdt, arch="xvector")
_, er = t(wav.cuda(), is_eval=True, lengths=lens)
print("tdnn ragged rel per utt", [rel(er[i].cpu(), rr[i]) for i in range(3)])
c = Xtractor(16, "xvector", "cce", seed=4321).to("cuda").eval()
with torch.no_grad():
    rc = oxv.tdnn_from_feats(feats, c.state_dict(), "cce")
print("tdnn cce rel", rel(c.forward_features(feats.cuda()).cpu(), rc))
