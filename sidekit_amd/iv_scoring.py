"""Trial scoring -- mirror of ``sidekit/iv_scoring.py``: ``cosine_scoring`` (:63-113), ``PLDA_scoring``
(:215-269), ``full_PLDA_scoring`` (:272-368), ``fast_PLDA_scoring`` (:370-477).

The trial matrix is computed on the GPU through the C ABI (``sc_cosine``: f32 MFMA GEMM;
``sc_plda_fast``: float64 tiled GEMM with the quadratic terms and the constant fused in the
epilogue).  As in the reference the 256x256 float64 algebra (inverses, slogdet) stays on the host.
``full_PLDA_scoring`` maps onto the same device entry point: with e' = B e, t' = B t the reference's
per-model loop is  0.5 (e'+t')' K2 (e'+t') - 0.5 t' K1 t - 0.5 e' K1 e  =  0.5 e'(K2-K1)e' +
0.5 t'(K2-K1)t' + e' sym(K2) t'.  There is no CPU fallback.
"""
import copy
import ctypes
import logging

import numpy
import scipy.linalg
import torch

from . import _lib
from .bosaris import Ndx, Scores
from .statserver import StatServer


def _check_missing_model(enroll, test, ndx):
    """Drop trials whose model / segment has no vector, align both StatServers to the cleaned Ndx."""
    clean_ndx = ndx.filter(enroll.modelset, test.segset, True)
    enroll.align_models(clean_ndx.modelset)
    test.align_segments(clean_ndx.segset)
    return clean_ndx


def _device(device):
    if not torch.cuda.is_available():
        raise RuntimeError("sidekit_amd.iv_scoring computes on the GPU only (no CPU fallback) and no GPU is visible")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"sidekit_amd.iv_scoring computes on the GPU only; got device={device}")
    return device if device.index is not None else torch.device("cuda", torch.cuda.current_device())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _to_device(x, dtype, device):
    """numpy array or torch tensor (any device) -> contiguous tensor of `dtype` on `device`; a tensor already there is used as is."""
    if torch.is_tensor(x):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(numpy.ascontiguousarray(x, dtype=numpy.float32 if dtype == torch.float32 else numpy.float64)).to(device)


def cosine_matrix_device(enroll_vectors, test_vectors, device=None):
    """(Ne, D) x (Nt, D) already-normalised vectors -> (Ne, Nt) float32 **device tensor**: x-vectors that are already on the GPU
    (fresh from ``Xtractor.forward`` or an all-gather) are scored where they are, nothing crosses PCIe."""
    device = _device(device if device is not None else (enroll_vectors.device if torch.is_tensor(enroll_vectors) and enroll_vectors.is_cuda else None))
    e, t = _to_device(enroll_vectors, torch.float32, device), _to_device(test_vectors, torch.float32, device)
    D = e.shape[1]
    if D % 4:  # the GEMM wants K % 4 == 0: zero columns do not change a dot product
        pad = 4 - D % 4
        e = torch.nn.functional.pad(e, (0, pad))
        t = torch.nn.functional.pad(t, (0, pad))
        D += pad
    out = torch.empty((e.shape[0], t.shape[0]), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().sc_cosine(e.data_ptr(), e.shape[0], t.data_ptr(), t.shape[0], D, out.data_ptr(), _stream(device)),
                   AssertionError)
    return out


def cosine_matrix(enroll_vectors, test_vectors, device=None):
    """(Ne, D) x (Nt, D) already-normalised float vectors -> (Ne, Nt) float32 numpy matrix (computed on the GPU)."""
    return cosine_matrix_device(enroll_vectors, test_vectors, device).cpu().numpy()


def plda_matrix_device(enroll_vectors, test_vectors, Phi, Psi, cst, scaling_factor=1., device=None):
    """scaling * (0.5 e'Phi e + 0.5 t'Phi t + cst + e'Psi t) for all pairs, float64 **device tensor** (f64 MFMA GEMM)."""
    device = _device(device if device is not None else (enroll_vectors.device if torch.is_tensor(enroll_vectors) and enroll_vectors.is_cuda else None))
    e, t = _to_device(enroll_vectors, torch.float64, device), _to_device(test_vectors, torch.float64, device)
    phi, psi = _to_device(Phi, torch.float64, device), _to_device(Psi, torch.float64, device)
    out = torch.empty((e.shape[0], t.shape[0]), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().sc_plda_fast(e.data_ptr(), e.shape[0], t.data_ptr(), t.shape[0], e.shape[1], phi.data_ptr(),
                                           psi.data_ptr(), float(cst), float(scaling_factor), out.data_ptr(), _stream(device)),
                   AssertionError)
    return out


def plda_matrix(enroll_vectors, test_vectors, Phi, Psi, cst, scaling_factor=1., device=None):
    """Same, returned as a float64 numpy matrix."""
    return plda_matrix_device(enroll_vectors, test_vectors, Phi, Psi, cst, scaling_factor, device).cpu().numpy()


HIST_BINS = 8192


def cosine_histograms(enroll_vectors, test_vectors, enroll_labels, test_labels, self_offset=None, lo=-1.0, hi=1.0, device=None):
    """Target / non-target score histograms of ALL (enrol, test) pairs without materialising the (Ne, Nt) score matrix
    (100k x 100k cosine trials are 40 GB).  A trial is a target when the two integer labels are equal;
    ``self_offset=k`` drops the self-trials ``j == i + k`` when the enrolment side is rows ``[k, k + Ne)`` of the test side (``0`` for
    a set scored against itself, a shard's first row for one rank's block of it); ``None`` keeps every pair.  Returns two uint64 arrays of
    ``HIST_BINS`` equal bins over ``[lo, hi)``; ``bosaris.detplot.eer_from_histograms`` turns them into the ROCCH EER."""
    device = _device(device if device is not None else (enroll_vectors.device if torch.is_tensor(enroll_vectors) and enroll_vectors.is_cuda else None))
    e, t = _to_device(enroll_vectors, torch.float32, device), _to_device(test_vectors, torch.float32, device)
    if e.shape[1] % 4 or e.shape[1] != t.shape[1]:
        raise AssertionError("x-vector dimensions must match and be a multiple of 4")
    le = torch.as_tensor(enroll_labels).to(device=device, dtype=torch.int32).contiguous()
    lt = torch.as_tensor(test_labels).to(device=device, dtype=torch.int32).contiguous()
    assert le.shape == (e.shape[0],) and lt.shape == (t.shape[0],), "one label per vector"
    ht = torch.empty(HIST_BINS, dtype=torch.int64, device=device)
    hn = torch.empty(HIST_BINS, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().sc_cosine_hist(e.data_ptr(), e.shape[0], t.data_ptr(), t.shape[0], e.shape[1], le.data_ptr(), lt.data_ptr(),
                                             -1 if self_offset is None else int(self_offset), float(lo), float(hi), HIST_BINS, ht.data_ptr(), hn.data_ptr(),
                                             _stream(device)), AssertionError)
    return ht.cpu().numpy().astype(numpy.uint64), hn.cpu().numpy().astype(numpy.uint64)


def plda_parameters(mu, F, Sigma, scaling_factor=1.):
    """The 256 x 256 float64 algebra of ``fast_PLDA_scoring`` (``iv_scoring.py:428-446``): ``(Phi, Psi, plda_cst)`` such that
    ``score(e, t) = scaling * (0.5 e'Phi e + 0.5 t'Phi t + plda_cst + e'Psi t)`` for centred vectors."""
    Phi, Psi, plda_cst = plda_parameters(mu, F, Sigma, scaling_factor)
    score = Scores()
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    score.scoremat = plda_matrix(enroll_ctr.stat1, test_ctr.stat1, Phi, Psi, plda_cst, scaling_factor, device)
    if p_known != 0:
        score.scoremat = _open_set(score.scoremat, p_known)
    return score
