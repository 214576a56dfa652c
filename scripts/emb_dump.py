"""x-vectors of a fixed seeded ragged batch (bf16 and fp32 paths) from one build of the library, dumped or compared bit for bit with an
earlier dump: `SK_LIB=other.so python scripts/emb_dump.py out.npz` then `python scripts/emb_dump.py out.npz` (GPU box).  A change of a
kernel's tiling / lane order that leaves every output's arithmetic alone must give `identical True`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from sidekit_amd import _lib
if os.environ.get("SK_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SK_LIB"])
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
g = torch.Generator().manual_seed(5)
B, L = 48, 64000
wav = (0.1 * torch.randn(B, L, generator=g)).to(dev)
lens = [L - 1237 * (i % 13) - 160 * (i % 7) for i in range(B)]
out = {}
for dt in ("bf16", "fp32"):
    m.compute_dtype = dt
    logits, emb = m(wav, is_eval=True, lengths=lens)
    out[dt] = emb.float().cpu().numpy()
    out[dt + "_logits"] = logits.float().cpu().numpy()
m.compute_dtype = "bf16"
m.set_debug(True)
m(wav[:8], is_eval=True, lengths=lens[:8])
for k, v in m.debug_taps(["stem", "layer1", "layer2", "layer3", "layer4"]).items():   # bf16 NHWC activations, as bytes
    out["tap_" + k] = v.view(numpy.uint16)
m.set_debug(False)
path = sys.argv[1]
if os.path.exists(path):
    ref = numpy.load(path)
    for k, v in out.items():
        if k.startswith("tap_"):
            a = (v.astype(numpy.uint32) << 16).view(numpy.float32); b = (ref[k].astype(numpy.uint32) << 16).view(numpy.float32)
            ne = a != b
            print(f"{k}: {int(ne.sum())} of {a.size} bf16 values differ ({ne.mean():.2e}), max abs diff {numpy.abs(a - b).max():.3e}, max |value| {numpy.abs(b).max():.2f}")
            if k == "tap_layer1" and ne.any():
                idx = numpy.nonzero(ne.reshape(8, -1, 80, 32))
                for nm, ix in zip("btwc", idx):
                    u, c = numpy.unique(ix, return_counts=True)
                    print("   ", nm, dict(zip(u.tolist()[:24], c.tolist()[:24])), "..." if len(u) > 24 else "")
            continue
        print(f"{k}: identical {bool((v == ref[k]).all())}  max abs diff {numpy.abs(v - ref[k]).max():.3e}  finite {bool(numpy.isfinite(v).all())}")
else:
    numpy.savez(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
