"""Trial scoring -- mirror of ``sidekit/iv_scoring.py``: ``cosine_scoring`` (:63-113), ``PLDA_scoring``
(:215-269), ``full_PLDA_scoring`` (:272-368), ``fast_PLDA_scoring`` (:370-477), and -- beyond SURVEY 8's rows, because they are
the same device entry point with other matrices -- ``mahalanobis_scoring`` (:116-156) and ``two_covariance_scoring`` (:159-213).

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


def normalize_rows_device(vectors, device=None):
    """``torch.nn.functional.normalize(x, dim=1)`` on the GPU (``sc_normalize_rows``): (N, D) -> float32 **device tensor** of unit rows."""
    device = _device(device if device is not None else (vectors.device if torch.is_tensor(vectors) and vectors.is_cuda else None))
    x = _to_device(vectors, torch.float32, device)
    out = torch.empty_like(x)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().sc_normalize_rows(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _stream(device)), AssertionError)
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


def cosine_histograms(enroll_vectors, test_vectors, enroll_labels, test_labels, self_offset=None, lo=-1.0, hi=1.0, device=None, bins=None):
    """Target / non-target score histograms of ALL (enrol, test) pairs without materialising the (Ne, Nt) score matrix
    (100k x 100k cosine trials are 40 GB).  A trial is a target when the two integer labels are equal;
    ``self_offset=k`` drops the self-trials ``j == i + k`` when the enrolment side is rows ``[k, k + Ne)`` of the test side (``0`` for
    a set scored against itself, a shard's first row for one rank's block of it); ``None`` keeps every pair.  Returns two uint64 arrays of
    ``bins`` equal bins over ``[lo, hi)`` (scores outside land in the end bins); ``bosaris.detplot.eer_from_histograms`` turns them into the
    ROCCH EER.  ``bins``: ``HIST_BINS`` (8192, the kernel's LDS histograms: one pass over the pairs) or a multiple of ``HIST_BINS - 2``: that
    many finer bins from ``bins / (HIST_BINS - 2)`` passes, each over a slice of the range with one guard bin either side."""
    device = _device(device if device is not None else (enroll_vectors.device if torch.is_tensor(enroll_vectors) and enroll_vectors.is_cuda else None))
    e, t = _to_device(enroll_vectors, torch.float32, device), _to_device(test_vectors, torch.float32, device)
    if e.shape[1] % 4 or e.shape[1] != t.shape[1]:
        raise AssertionError("x-vector dimensions must match and be a multiple of 4")
    le = torch.as_tensor(enroll_labels).to(device=device, dtype=torch.int32).contiguous()
    lt = torch.as_tensor(test_labels).to(device=device, dtype=torch.int32).contiguous()
    assert le.shape == (e.shape[0],) and lt.shape == (t.shape[0],), "one label per vector"
    bins = HIST_BINS if bins is None else int(bins)
    inner = HIST_BINS - 2
    if bins != HIST_BINS and (bins <= 0 or bins % inner):
        raise AssertionError(f"bins must be {HIST_BINS} or a multiple of {inner}")
    if not float(hi) > float(lo):
        raise AssertionError("histogram range: hi must exceed lo")

    def one_pass(a, b):
        ht = torch.empty(HIST_BINS, dtype=torch.int64, device=device)
        hn = torch.empty(HIST_BINS, dtype=torch.int64, device=device)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().sc_cosine_hist(e.data_ptr(), e.shape[0], t.data_ptr(), t.shape[0], e.shape[1], le.data_ptr(), lt.data_ptr(),
                                                 -1 if self_offset is None else int(self_offset), float(a), float(b), HIST_BINS, ht.data_ptr(), hn.data_ptr(),
                                                 _stream(device)), AssertionError)
        return ht.cpu().numpy().astype(numpy.uint64), hn.cpu().numpy().astype(numpy.uint64)

    if bins == HIST_BINS:
        return one_pass(lo, hi)
    w = (float(hi) - float(lo)) / bins
    out_t, out_n = numpy.zeros(bins, dtype=numpy.uint64), numpy.zeros(bins, dtype=numpy.uint64)
    passes = bins // inner
    for k in range(passes):
        a = float(lo) + k * inner * w
        ht, hn = one_pass(a - w, a + (inner + 1) * w)          # bins 0 and HIST_BINS - 1 of a pass: everything below / above its slice
        for full, part in ((out_t, ht), (out_n, hn)):
            full[k * inner:(k + 1) * inner] = part[1:-1]
            if k == 0:
                full[0] += part[0]
            if k == passes - 1:
                full[-1] += part[-1]
    return out_t, out_n


def _speaker_posterior_terms(K):
    """For ``K`` = speaker-subspace precision gained from ONE observation: the posterior covariances after one and after two observations
    of a speaker, ``(I + K)^-1`` and ``(I + 2K)^-1``, and the constant of the log-likelihood ratio they leave,
    ``log|I + K| - 0.5 log|I + 2K|``."""
    eye = numpy.eye(K.shape[0])
    one, two = eye + K, eye + 2.0 * K
    return numpy.linalg.inv(one), numpy.linalg.inv(two), numpy.linalg.slogdet(one)[1] - 0.5 * numpy.linalg.slogdet(two)[1]


def plda_parameters(mu, F, Sigma, scaling_factor=1.):
    """The D x D float64 algebra of two-covariance PLDA scoring, kept on the host as the reference keeps it (``iv_scoring.py:428-446``):
    ``(Phi, Psi, plda_cst)`` such that ``score(e, t) = scaling * (0.5 e'Phi e + 0.5 t'Phi t + plda_cst + e'Psi t)`` for centred vectors.

    A same-speaker pair ``(e, t)`` is jointly Gaussian with covariance ``[[T, A], [A, T]]`` (``A = F F'`` across speakers, ``T = A + Sigma``
    total), a different-speaker pair with ``[[T, 0], [0, T]]``; the log-likelihood ratio's quadratic form is the difference of the two
    precisions, whose blocks follow from the Schur complement ``S = T - A T^-1 A``: diagonal ``T^-1 - S^-1``, off-diagonal ``T^-1 A S^-1``."""
    F = numpy.asarray(F, dtype=numpy.float64)
    within = numpy.asarray(Sigma, dtype=numpy.float64)
    across = F @ F.T
    total = across + within
    total_inv = numpy.linalg.inv(total)
    schur_inv = numpy.linalg.inv(total - across @ total_inv @ across)
    Phi = total_inv - schur_inv
    Psi = total_inv @ across @ schur_inv
    _, _, plda_cst = _speaker_posterior_terms(scaling_factor * (F.T @ numpy.linalg.solve(within, F)))
    return Phi, Psi, plda_cst


def full_plda_parameters(F, G, Sigma, scaling_factor=1.):
    """Host algebra of PLDA with a channel sub-space (``iv_scoring.py:299-330``): ``(B, Phi, Psi, constant)`` with which the reference's
    per-model loop becomes the two-covariance kernel's form on projected vectors ``e' = B e``, ``t' = B t`` (module docstring).

    Precision of an observation once the channel factor is integrated out (Woodbury on ``Sigma + G G'``), then everything lives in the
    speaker sub-space: ``B`` projects a centred vector there and ``B F`` is what one observation adds to the speaker's posterior precision."""
    F, G = numpy.asarray(F, dtype=numpy.float64), numpy.asarray(G, dtype=numpy.float64)
    prec = scaling_factor * numpy.linalg.inv(numpy.asarray(Sigma, dtype=numpy.float64))
    PG = prec @ G
    prec_marg = prec - PG @ numpy.linalg.inv(numpy.eye(G.shape[1]) + G.T @ PG) @ PG.T
    B = F.T @ prec_marg
    K1, K2, constant = _speaker_posterior_terms(B @ F)
    return B, K2 - K1, 0.5 * (K2 + K2.T), constant


def _open_set(scoremat, p_known):
    """Open-set identification term (iv_scoring.py:467-475): impostor mass = mean of the other models' likelihoods."""
    N = scoremat.shape[0]
    tmp = numpy.exp(scoremat)
    others = tmp.sum(axis=0)[numpy.newaxis, :] - tmp
    return scoremat - numpy.log(p_known * others / (N - 1) + (1 - p_known))


def cosine_scoring(enroll, test, ndx, wccn=None, check_missing=True, device=None):
    """Cosine similarity of every (model, segment) pair of ``ndx``; returns a ``Scores`` (float32 matrix)."""
    assert isinstance(enroll, StatServer), 'First parameter should be a StatServer'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    enroll_copy = copy.deepcopy(enroll)
    test_copy = copy.deepcopy(test)
    clean_ndx = _check_missing_model(enroll_copy, test_copy, ndx) if check_missing else ndx
    if wccn is not None:
        enroll_copy.rotate_stat1(wccn)
        test_copy.rotate_stat1(wccn)
    enroll_copy.norm_stat1()
    test_copy.norm_stat1()
    score = Scores()
    score.scoremat = cosine_matrix(enroll_copy.stat1, test_copy.stat1, device)
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    return score


def PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, test_uncertainty=None, Vtrans=None, p_known=0.0, scaling_factor=1.,
                 full_model=False):
    """PLDA log-likelihood ratios; dispatches to the two-covariance form unless ``full_model``."""
    assert isinstance(enroll, StatServer), 'First parameter should be a StatServer'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    assert enroll.stat1.shape[1] == test.stat1.shape[1], 'I-vectors dimension mismatch'
    assert enroll.stat1.shape[1] == F.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    assert enroll.stat1.shape[1] == G.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    if not full_model:
        return fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, test_uncertainty, Vtrans, p_known=p_known,
                                 scaling_factor=scaling_factor, check_missing=True)
    return full_PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, p_known=p_known, scaling_factor=scaling_factor)


def full_PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, p_known=0.0, scaling_factor=1., check_missing=True, device=None):
    """PLDA with a channel sub-space G."""
    enroll_copy = copy.deepcopy(enroll)
    test_copy = copy.deepcopy(test)
    clean_ndx = _check_missing_model(enroll_copy, test_copy, ndx) if check_missing else ndx
    enroll_copy.center_stat1(mu)
    test_copy.center_stat1(mu)
    B, Phi, Psi, constant = full_plda_parameters(F, G, Sigma, scaling_factor)
    enroll_tmp = enroll_copy.stat1 @ B.T      # (Ne, rank): speaker-subspace projections
    test_tmp = test_copy.stat1 @ B.T
    score = Scores()
    score.scoremat = plda_matrix(enroll_tmp, test_tmp, Phi, Psi, constant, scaling_factor, device)
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    if p_known != 0:
        score.scoremat = _open_set(score.scoremat, p_known)
    return score


def fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, test_uncertainty=None, Vtrans=None, p_known=0.0, scaling_factor=1.,
                      check_missing=True, device=None):
    """Two-covariance PLDA scoring of all trials of ``ndx`` (float64)."""
    enroll_ctr = copy.deepcopy(enroll)
    test_ctr = copy.deepcopy(test)
    if not numpy.unique(enroll_ctr.modelset).shape == enroll_ctr.modelset.shape:
        logging.warning("Enrollment models are not unique, average i-vectors")
        enroll_ctr = enroll_ctr.mean_stat_per_model()
    clean_ndx = _check_missing_model(enroll_ctr, test_ctr, ndx) if check_missing else ndx
    enroll_ctr.center_stat1(mu)
    test_ctr.center_stat1(mu)
    Phi, Psi, plda_cst = plda_parameters(mu, F, Sigma, scaling_factor)
    score = Scores()
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    score.scoremat = plda_matrix(enroll_ctr.stat1, test_ctr.stat1, Phi, Psi, plda_cst, scaling_factor, device)
    if p_known != 0:
        score.scoremat = _open_set(score.scoremat, p_known)
    return score


def _prepared(enroll, test, ndx, check_missing):
    """Shared head of the two functions below (iv_scoring.py:134-143,183-192): duplicate models averaged with a warning, missing
    models / segments dropped.  The reference works on the caller's objects (its alignment reorders them in place); here copies."""
    enroll, test = copy.deepcopy(enroll), copy.deepcopy(test)
    if not numpy.unique(enroll.modelset).shape == enroll.modelset.shape:
        logging.warning("Enrollment models are not unique, average i-vectors")
        enroll = enroll.mean_stat_per_model()
    clean_ndx = _check_missing_model(enroll, test, ndx) if check_missing else ndx
    return enroll, test, clean_ndx


def _scores(scoremat, clean_ndx):
    score = Scores()
    score.scoremat = scoremat
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    return score


def mahalanobis_scoring(enroll, test, ndx, m, check_missing=True, device=None):
    """``-0.5 (e - t)' M (e - t)`` for every trial (iv_scoring.py:116-156; the reference loops over models on the host).  Expanded it is
    the PLDA kernel's form with ``Phi = -sym(M)``, ``Psi = sym(M)``, no constant: one ``sc_plda_fast`` call, float64."""
    assert isinstance(enroll, StatServer), 'First parameter should be a StatServer'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    assert enroll.stat1.shape[1] == test.stat1.shape[1], 'I-vectors dimension mismatch'
    assert enroll.stat1.shape[1] == m.shape[0], 'I-vectors and Mahalanobis matrix dimension mismatch'
    enroll, test, clean_ndx = _prepared(enroll, test, ndx, check_missing)
    ms = 0.5 * (numpy.asarray(m, dtype=numpy.float64) + numpy.asarray(m, dtype=numpy.float64).T)
    return _scores(plda_matrix(enroll.stat1, test.stat1, -ms, ms, 0.0, 1.0, device), clean_ndx)


def two_covariance_scoring(enroll, test, ndx, W, B, check_missing=True, device=None):
    """Two-covariance scores (iv_scoring.py:159-213): ``(e + t)' G (e + t) - t' H t - e' H e`` with ``G = iW (iB + 2 iW)^-1 iW``,
    ``H = iW (iB + iW)^-1 iW`` from the host's float64 algebra, i.e. ``Phi = 2 sym(G - H)``, ``Psi = G + G'`` in the PLDA kernel's form."""
    assert isinstance(enroll, StatServer), 'First parameter should be a directory'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    assert enroll.stat1.shape[1] == test.stat1.shape[1], 'I-vectors dimension mismatch'
    assert enroll.stat1.shape[1] == W.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    assert enroll.stat1.shape[1] == B.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    enroll, test, clean_ndx = _prepared(enroll, test, ndx, check_missing)
    iW, iB = scipy.linalg.inv(W), scipy.linalg.inv(B)
    G = iW.dot(scipy.linalg.inv(iB + 2 * iW)).dot(iW)
    H = iW.dot(scipy.linalg.inv(iB + iW)).dot(iW)
    GH = G - H
    return _scores(plda_matrix(enroll.stat1, test.stat1, GH + GH.T, G + G.T, 0.0, 1.0, device), clean_ndx)
