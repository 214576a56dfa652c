"""Probe: the stem kernel (explicit v_pk_fma_f32) and the whole trunk on a side stream beside the synthetic MFMA co-runner."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIDEKIT_AMD_LANES"] = "1"
import numpy, torch
from sidekit_amd.nnet import Xtractor
ag = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libaggressor.so"))
ag.aggressor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
m2 = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
g = torch.Generator(device="cuda").manual_seed(0)
b = 0.1 * torch.randn(128, 64000, device="cuda", generator=g)
feats = m2.features(b)
sink = torch.zeros(4, device=dev)
main = torch.cuda.current_stream()
s2 = torch.cuda.Stream()
for dtype in ("bf16", "fp32"):
    m2.compute_dtype = dtype
    m2.set_debug(True)
    m2.forward_features(feats); torch.cuda.synchronize()
    ref = {k: v.copy() for k, v in m2.debug_taps(["stem", "layer1", "layer4", "pooled"]).items()}
    bad = {k: 0 for k in ref}
    for trial in range(20):
        ag.aggressor_launch(ctypes.c_void_p(main.cuda_stream), 2, 1024, 6000 if dtype == "bf16" else 30000, ctypes.c_void_p(sink.data_ptr()))
        with torch.cuda.stream(s2):
            m2.forward_features(feats)
        torch.cuda.synchronize()
        got = m2.debug_taps(list(ref))
        for k in ref:
            bad[k] += int((got[k] != ref[k]).sum())
    m2.set_debug(False)
    print(f"{dtype} trunk beside the register-operand MFMA co-runner, 20 trials: differing bytes per tap {bad}", flush=True)
