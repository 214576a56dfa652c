"""Determinism of the two-lane forward against the serial one over many forwards (GPU box).
SIDEKIT_AMD_LANE_DIAG: 1 = second lane on the caller's stream, 2 = second lane starts after the first has finished."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor

dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
g = torch.Generator(device="cuda").manual_seed(0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
m.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for B, L in ((256, 64000), (200, 48000)):
    wav = 0.1 * torch.randn(B, L, device="cuda", generator=g)
    m.set_lanes(1)
    lref, ref = m(wav, is_eval=True)
    m.set_lanes(2)
    rows = set(); lrows = set()
    for trial in range(trials):
        lg, emb = m(wav, is_eval=True)
        rows.update(torch.nonzero((emb - ref).abs().amax(dim=1) > 0).flatten().tolist())
        lrows.update(torch.nonzero((lg - lref).abs().amax(dim=1) > 0).flatten().tolist())
    print(f"diag={os.environ.get('SIDEKIT_AMD_LANE_DIAG', '0')} {m.compute_dtype} B={B}: rows that ever differed from the serial forward over {trials} two-lane forwards: "
          f"x-vectors {sorted(rows)[:20]} ({len(rows)}), logits ({len(lrows)})", flush=True)
