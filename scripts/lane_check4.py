"""[needs the SIDEKIT_AMD_STOP_STAGE / SIDEKIT_AMD_STOP_LAUNCH hooks that commit afd6222 carried in xt_api.hip; removed afterwards] Diagnostic: LDS canary (scripts/diag/libcanary.so) on a side stream beside the bf16 trunk truncated at SIDEKIT_AMD_STOP_STAGE."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIDEKIT_AMD_LANES"] = "1"
import torch
from sidekit_amd.nnet import Xtractor
from sidekit_amd import _lib
_lib.lib()
can = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libcanary.so"))
can.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
can.canary4_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda", 0)
m1 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
g = torch.Generator(device="cuda").manual_seed(0)
a = 0.1 * torch.randn(128, 64000, device="cuda", generator=g)
feats_a = m1.features(a)
m1.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for w in range(2): m1.forward_features(feats_a)
torch.cuda.synchronize()
s2 = torch.cuda.Stream()
err = torch.zeros(1, dtype=torch.int64, device=dev)
for lds in (8192, 16384, 32768):
    err.zero_(); torch.cuda.synchronize()
    for trial in range(5):
        can.canary_launch(ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(err.data_ptr()), 256 * 4, 3000, lds)
        m1.forward_features(feats_a)
        torch.cuda.synchronize()
    print(f"stop stage {os.environ.get('SIDEKIT_AMD_STOP_STAGE', '-')} {m1.compute_dtype}: canary LDS {lds} B: words overwritten {int(err.item())}", flush=True)

for lds in (26752, 16384, 40960):
    err.zero_(); torch.cuda.synchronize()
    for trial in range(5):
        can.canary4_launch(ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(err.data_ptr()), 256 * 6, 1500, lds)
        m1.forward_features(feats_a)
        torch.cuda.synchronize()
    print(f"stop stage {os.environ.get('SIDEKIT_AMD_STOP_STAGE', '-')} {m1.compute_dtype}: 4-wave canary, LDS {lds} B per workgroup: words that read back wrong {int(err.item())}", flush=True)
err.zero_(); torch.cuda.synchronize()
for trial in range(3):
    can.canary4_launch(ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(err.data_ptr()), 256 * 6, 1500, 26752)
    torch.cuda.synchronize()
print(f"4-wave canary alone: words that read back wrong {int(err.item())}", flush=True)
