"""Batch-1 bf16 forwards for a rocprofv3 kernel trace (GPU box): `rocprofv3 --kernel-trace --stats -d out -- python3 scripts/b1_forward.py`;
SK_LIB=<other build> selects the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd import _lib
if os.environ.get("SK_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SK_LIB"])
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
wav = 0.1 * torch.randn(1, 64000, device=dev)
for _ in range(100):
    m(wav, is_eval=True)
torch.cuda.synchronize()
