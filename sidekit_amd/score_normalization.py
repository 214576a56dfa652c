"""Adaptive symmetric score normalisation -- mirror of ``sidekit.score_normalization.asnorm``
(``sidekit/score_normalization.py:120-140``), the normalisation behind the reference's "norm EER"
(``sidekit/nnet/xvector.py:261``).

All-vs-all cosine scores of the enrolment x-vectors, cohort scores against the L2-normalised cohort,
mean / std of each row's 200 best cohort scores, then the symmetric normalisation
``0.5 (s - m_i)/sd_i + 0.5 (s - m_j)/sd_j``.  On the GPU: two f32 MFMA GEMMs (``sc_cosine``), an exact
radix-select top-k statistics kernel (``sc_topk_stats``) and an elementwise pass (``sc_snorm_apply``).
"""
import ctypes

import numpy
import torch

from . import _lib


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def asnorm(enrol_xv, cohort_xv, ndx=None, topk=200, device=None):
    """Same arguments as the reference (``ndx`` is unused there too); returns the (N, N) float32 numpy matrix."""
    if not torch.cuda.is_available():
        raise RuntimeError("sidekit_amd computes on the GPU only (no CPU fallback) and no GPU is visible")
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    e = torch.as_tensor(enrol_xv, dtype=torch.float32).to(device).contiguous()
    from .iv_scoring import normalize_rows_device
    c = normalize_rows_device(torch.as_tensor(cohort_xv, dtype=torch.float32), device)     # F.normalize of the cohort, on the device
    n, d = e.shape
    if d % 4 or c.shape[1] != d:
        raise ValueError("x-vector dimension must match and be a multiple of 4")
    lib = _lib.lib()
    scores = torch.empty((n, n), dtype=torch.float32, device=device)
    calib = torch.empty((n, c.shape[0]), dtype=torch.float32, device=device)
    mean = torch.empty(n, dtype=torch.float32, device=device)
    std = torch.empty(n, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        st = _stream(device)
        _lib.check(lib.sc_cosine(e.data_ptr(), n, e.data_ptr(), n, d, scores.data_ptr(), st))
        _lib.check(lib.sc_cosine(e.data_ptr(), n, c.data_ptr(), c.shape[0], d, calib.data_ptr(), st))
        _lib.check(lib.sc_topk_stats(calib.data_ptr(), n, c.shape[0], int(topk), mean.data_ptr(), std.data_ptr(), st))
        _lib.check(lib.sc_snorm_apply(scores.data_ptr(), n, n, mean.data_ptr(), std.data_ptr(), mean.data_ptr(), std.data_ptr(), st))
    return scores.cpu().numpy()
