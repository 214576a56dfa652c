"""Diagnostic: synthetic co-runners (scripts/diag/libaggressor.so) on the default stream beside m2.features on a side stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIDEKIT_AMD_LANES"] = "1"
import torch
from sidekit_amd.nnet import Xtractor
ag = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libaggressor.so"))
ag.aggressor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
m2 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
g = torch.Generator(device="cuda").manual_seed(0)
b = 0.1 * torch.randn(128, 64000, device="cuda", generator=g)
ref = m2.features(b)
sink = torch.zeros(4, device=dev)
torch.cuda.synchronize()
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()
for mode, name in ((3, "random-operand MFMA fed from LDS"), (2, "MFMA, register operands"), (1, "ds_read_b128 spam")):
    nbad = 0; worst = 0.0
    for trial in range(12):
        ag.aggressor_launch(ctypes.c_void_p(main.cuda_stream), mode, 1024, 4000, ctypes.c_void_p(sink.data_ptr()))
        with torch.cuda.stream(s2):
            outs = [m2.features(b) for _ in range(3)]
        torch.cuda.synchronize()
        for o in outs:
            d = (o - ref).abs()
            nbad += int((d.amax(dim=(1, 2)) > 0).sum()); worst = max(worst, float(d.max()))
    print(f"co-runner {name:20s}: utterances whose features differed {nbad} of {12 * 3 * 128}, max abs diff {worst:.3e}", flush=True)
