"""Where does the exact-f32 GEMM lose its matrix-pipe cycles?  Times sc_cosine (E . T^T, 16384 x 16384, K = 1536: the TDNN
layer shape) for the 64 x 64 and 128 x 128 tilings and with the operand prefetch removed (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd import iv_scoring

N, D = 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 1536
g = torch.Generator(device="cuda").manual_seed(0)
E = torch.nn.functional.normalize(torch.randn(N, D, device="cuda", generator=g), dim=1)
T = torch.nn.functional.normalize(torch.randn(N, D, device="cuda", generator=g), dim=1)
def run(tag, **env):
    for k in ("SIDEKIT_AMD_GEMM64", "SIDEKIT_AMD_GEMM_DBG"): os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(2): iv_scoring.cosine_matrix_device(E, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): iv_scoring.cosine_matrix_device(E, T)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{tag:40s} {ms:8.3f} ms  {2.0 * N * N * D / ms * 1e-9:7.1f} TFLOP/s", flush=True)
run("128 x 128 tiles")
run("64 x 64 tiles", SIDEKIT_AMD_GEMM64="1")
run("128 x 128, no operand prefetch", SIDEKIT_AMD_GEMM_DBG="1")
