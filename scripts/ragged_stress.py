"""Randomised ragged batches: the bf16 tilings against the fp32 path (different kernels' configurations), and batched vs
single-utterance results bit for bit in bf16 (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
m = Xtractor(64, "halfresnet34", "aam", seed=9).to("cuda").eval()
gen = torch.Generator().manual_seed(11)
worst, bad = 1.0, 0
for rd in range(rounds):
    B = int(torch.randint(1, 40, (1,), generator=gen))
    lens = [int(x) for x in torch.randint(2000, 100000, (B,), generator=gen)]
    if rd % 3 == 0:   # frame counts right at tile edges
        lens = [(int(t) - 1) * 160 + 5 for t in torch.randint(12, 620, (B,), generator=gen)]
    wav = 0.1 * torch.randn(B, max(lens), generator=gen)
    for i, n in enumerate(lens):
        wav[i, n:] = 5.0
    x = wav.cuda()
    m.compute_dtype = "fp32"
    e32 = m(x, is_eval=True, lengths=lens)[1]
    m.compute_dtype = "bf16"
    e16 = m(x, is_eval=True, lengths=lens)[1]
    cos = torch.nn.functional.cosine_similarity(e16, e32)
    ok = bool(torch.isfinite(e16).all()) and float(cos.min()) > 0.999
    worst = min(worst, float(cos.min()))
    for i in (0, B // 2, B - 1):
        one = m(x[i, :lens[i]], is_eval=True)[1]
        if not torch.equal(one[0], e16[i]):
            ok = False
            print("single != batched", rd, i, lens[i])
    bad += 0 if ok else 1
    print(rd, B, "min cos", round(float(cos.min()), 6), "ok" if ok else "FAIL", flush=True)
print("worst cosine", worst, "failures", bad)
sys.exit(1 if bad else 0)
