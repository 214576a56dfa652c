"""Kernel timeline of ONE batch-1 forward (the reference driver's call shape), for `rocprofv3 --kernel-trace -- python3 scripts/trace_b1.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to("cuda").eval()
m.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
wav = 0.1 * torch.randn(1, 64000, device="cuda")
for _ in range(5): m(wav, is_eval=True)
torch.cuda.synchronize()
