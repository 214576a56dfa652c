"""Trial-scoring kernels against their roofs (GPU box): cosine (f32 MFMA, 157 TFLOP/s = the f32 matrix/vector peak), fast PLDA
(f64 MFMA), and the matrix-free cosine histogram path.  Inputs resident in HBM, HIP events on the launch stream.

    python scripts/scoring_bench.py > gpurun_out/scoring_bench.json
"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from sidekit_amd import _lib, iv_scoring

dev = torch.device("cuda", 0)
lib = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
F32_PEAK, F64_PEAK, HBM = 157.3, 78.6, 8000.0      # TFLOP/s (MI355X_MICROARCH.md; f64 matrix = vector peak on CDNA4), GB/s


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


out = []
torch.manual_seed(0)
for N in (1000, 16384):
    e = torch.nn.functional.normalize(torch.randn(N, 256, device=dev), dim=1)
    t = torch.nn.functional.normalize(torch.randn(N, 256, device=dev), dim=1)
    o = torch.empty(N, N, device=dev)
    s = timed(lambda: _lib.check(lib.sc_cosine(e.data_ptr(), N, t.data_ptr(), N, 256, o.data_ptr(), st)), 20)
    fl, by = 2.0 * N * N * 256, (2 * N * 256 + N * N) * 4.0
    out.append({"kernel": "sc_cosine (gemm_kernel<LoadPlain>, f32 MFMA 32x32x2)", "Ne=Nt": N, "trials": N * N, "us": s * 1e6, "TFLOP/s": fl / s / 1e12,
                "frac_f32_mfma_peak": fl / s / 1e12 / F32_PEAK, "GB/s": by / s / 1e9, "frac_hbm": by / s / 1e9 / HBM, "trials_per_s": N * N / s})
    ed, td = e.double(), t.double()
    phi = torch.randn(256, 256, device=dev, dtype=torch.float64) / 256
    psi = torch.randn(256, 256, device=dev, dtype=torch.float64) / 256
    od = torch.empty(N, N, device=dev, dtype=torch.float64)
    s = timed(lambda: _lib.check(lib.sc_plda_fast(ed.data_ptr(), N, td.data_ptr(), N, 256, phi.data_ptr(), psi.data_ptr(), 0.5, 1.0, od.data_ptr(), st)), 10)
    fl, by = 2.0 * N * N * 256 + 3 * 2.0 * N * 256 * 256, (2 * N * 256 + N * N) * 8.0
    out.append({"kernel": "sc_plda_fast (plda_prep_kernel + dgemm_nt_kernel, v_mfma_f64_16x16x4_f64; 2 launches, cached workspace)", "Ne=Nt": N, "trials": N * N, "us": s * 1e6,
                "TFLOP/s": fl / s / 1e12, "frac_f64_peak": fl / s / 1e12 / F64_PEAK, "GB/s": by / s / 1e9, "frac_hbm": by / s / 1e9 / HBM,
                "trials_per_s": N * N / s})
for N in (16384, 65536):
    e = torch.nn.functional.normalize(torch.randn(N, 256, device=dev), dim=1)
    lab = torch.randint(0, 1000, (N,), device=dev, dtype=torch.int32)
    ht = torch.empty(8192, dtype=torch.int64, device=dev); hn = torch.empty(8192, dtype=torch.int64, device=dev)
    s = timed(lambda: _lib.check(lib.sc_cosine_hist(e.data_ptr(), N, e.data_ptr(), N, 256, lab.data_ptr(), lab.data_ptr(), 0, -1.0, 1.0, 8192,
                                                     ht.data_ptr(), hn.data_ptr(), st)), 3)
    fl = 2.0 * N * N * 256
    out.append({"kernel": "sc_cosine_hist (persistent f32 MFMA tiles -> LDS histograms, no score matrix)", "N": N, "trials": N * (N - 1), "ms": s * 1e3,
                "TFLOP/s": fl / s / 1e12, "frac_f32_mfma_peak": fl / s / 1e12 / F32_PEAK, "score_matrix_bytes_avoided": N * N * 4, "trials_per_s": N * (N - 1) / s})
print(json.dumps(out, indent=1))
