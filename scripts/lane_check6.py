"""[needs the SIDEKIT_AMD_STOP_STAGE / SIDEKIT_AMD_STOP_LAUNCH hooks that commit afd6222 carried in xt_api.hip; removed afterwards] Diagnostic: torch kernels as victims on a side stream beside the bf16 trunk truncated after its first convolution."""
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
x = torch.randn(64, 1 << 18, device=dev, generator=g)
victims = {
    "elementwise x*1.5+2": lambda: x * 1.5 + 2.0,
    "sort (LDS)": lambda: torch.sort(x[:16, :8192], dim=1).values,
    "softmax (LDS reductions)": lambda: torch.softmax(x[:, :4096], dim=1),
    "stft (rocFFT)": lambda: torch.stft(b[:32], 1024, 160, 400, window=torch.hann_window(400, device=dev), return_complex=True).abs(),
    "m2.features": lambda: m2.features(b),
    "m2 cmvn-free spectrum (features of zeros+wav)": lambda: m2.features(b[:, :32000]),
}
refs = {k: v() for k, v in victims.items()}
torch.cuda.synchronize()
s2 = torch.cuda.Stream()
for name, fn in victims.items():
    nbad = 0; worst = 0.0
    for trial in range(8):
        m1.forward_features(feats_a)
        with torch.cuda.stream(s2):
            outs = [fn() for _ in range(3)]
        torch.cuda.synchronize()
        for o in outs:
            d = (o - refs[name]).abs()
            nbad += int((d > 0).sum()); worst = max(worst, float(d.max()))
    print(f"victim {name:45s}: differing elements {nbad}, max abs diff {worst:.3e}", flush=True)
