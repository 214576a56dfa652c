"""Soak: the benchmark batch (and a ragged one) forwarded repeatedly must give bit-identical x-vectors every time (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
m = Xtractor(7205, "halfresnet34", "aam", seed=3).to("cuda").eval()
g = torch.Generator(device="cuda").manual_seed(1)
wav = 0.1 * torch.randn(256, 64000, device="cuda", generator=g)
lens = [int(x) for x in torch.randint(9000, 64000, (256,), generator=torch.Generator().manual_seed(2))]
bad = 0
for dtype in ("bf16", "fp32"):
    m.compute_dtype = dtype
    for name, kw in (("uniform", {}), ("ragged", {"lengths": lens})):
        ref = m(wav, is_eval=True, **kw)[1].clone()
        assert bool(torch.isfinite(ref).all())
        for i in range(n if dtype == "bf16" else max(4, n // 10)):
            e = m(wav, is_eval=True, **kw)[1]
            if not torch.equal(e, ref):
                bad += 1
                print("MISMATCH", dtype, name, i, float((e - ref).abs().max()), flush=True)
        print(dtype, name, "ok", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
