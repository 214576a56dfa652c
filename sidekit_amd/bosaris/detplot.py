"""ROC convex hull and EER: mirror of ``sidekit/bosaris/detplot.py`` ``pavx`` (:289-351), ``rocch2eer``
(:354-387), ``rocch`` (:390-436), ``sigmoid`` (:439-451), ``fast_minDCF`` (:454-511) and the prior
helpers.  The two sequential loops (pool-adjacent-violators, vertex walk) run in the library's
host code (``csrc/host_metrics.cpp``) with the reference's own floating-point operations, so the
bins and vertices are bit-identical to it; plotting (``DetPlot``) is out of scope."""
import ctypes

import numpy

from .. import _lib


def pavx(y):
    """Non-decreasing least-squares fit of ``y``: returns ``(ghat, width, height)``."""
    assert y.ndim == 1, 'Argument should be a 1-D array'
    assert y.shape[0] > 0, 'Input array is empty'
    y = numpy.ascontiguousarray(y, dtype=numpy.float64)
    n = y.shape[0]
    ghat = numpy.empty(n)
    width = numpy.empty(n, dtype=numpy.int64)
    height = numpy.empty(n)
    nbins = ctypes.c_int64(0)
    _lib.check(_lib.lib().sk_pavx(y.ctypes.data, n, ghat.ctypes.data, width.ctypes.data, height.ctypes.data, ctypes.byref(nbins)))
    nb = nbins.value
    return ghat, width[:nb].copy(), height[:nb].copy()


def rocch(tar_scores, nontar_scores):
    """Vertices ``(pmiss, pfa)`` of the ROC convex hull of the two score sets."""
    Nt, Nn = tar_scores.shape[0], nontar_scores.shape[0]
    scores = numpy.concatenate((tar_scores, nontar_scores))
    pideal = numpy.concatenate((numpy.ones(Nt), numpy.zeros(Nn)))
    order = numpy.argsort(scores, kind='mergesort')  # stable: equal scores must not be swapped
    pideal = numpy.ascontiguousarray(pideal[order])
    _, width, _ = pavx(pideal)
    nbins = width.shape[0]
    pmiss = numpy.zeros(nbins + 1)
    pfa = numpy.zeros(nbins + 1)
    width = numpy.ascontiguousarray(width, dtype=numpy.int64)
    _lib.check(_lib.lib().sk_rocch_vertices(pideal.ctypes.data, Nt + Nn, Nt, Nn, width.ctypes.data, nbins, pmiss.ctypes.data,
                                            pfa.ctypes.data))
    return pmiss, pfa


def rocch2eer(pmiss, pfa):
    """Equal error rate from the hull vertices (``detplot.py:354-387``).

    Every hull edge (pfa_i, pmiss_i) -> (pfa_i+1, pmiss_i+1) lies on a line ``a x + b y = 1``; that line crosses the
    diagonal ``x == y`` at ``1 / (a + b)`` and the EER is the largest crossing.  Axis-parallel edges contribute 0.  All
    edges go through ONE batched ``numpy.linalg.solve`` (the same LAPACK 2x2 factorisation per edge as a per-edge call,
    so the value is bit-identical to the reference's loop)."""
    pmiss = numpy.asarray(pmiss, dtype=numpy.float64)
    pfa = numpy.asarray(pfa, dtype=numpy.float64)
    if pfa.shape[0] < 2:
        return 0
    edges = numpy.stack((numpy.stack((pfa[:-1], pmiss[:-1]), axis=1), numpy.stack((pfa[1:], pmiss[1:]), axis=1)), axis=1)
    assert bool(numpy.all(edges[:, 1, 0] <= edges[:, 0, 0]) and numpy.all(edges[:, 0, 1] <= edges[:, 1, 1])), \
        'pmiss and pfa have to be sorted'
    slanted = (edges[:, 0, 0] != edges[:, 1, 0]) & (edges[:, 0, 1] != edges[:, 1, 1])
    if not slanted.any():
        return 0
    coef = numpy.linalg.solve(edges[slanted], numpy.ones((int(slanted.sum()), 2, 1)))
    crossing = 1 / (coef[:, 0, 0] + coef[:, 1, 0])
    return max(0, crossing.max())


def rocch_from_histograms(hist_tar, hist_non):
    """ROC convex hull vertices ``(pmiss, pfa)`` of scores known only as target / non-target counts per ascending score bin
    (``iv_scoring.cosine_histograms``): the bins are tied-score groups with multiplicities, so the hull is that of
    ``rocch`` on the binned scores -- a weighted pool-adjacent-violators fit of the per-bin target fraction."""
    ht = numpy.asarray(hist_tar, dtype=numpy.float64)
    hn = numpy.asarray(hist_non, dtype=numpy.float64)
    keep = (ht + hn) > 0
    ht, hn = ht[keep], hn[keep]
    n_tar, n_non = ht.sum(), hn.sum()
    assert n_tar > 0 and n_non > 0, "both histograms must hold trials"
    # weighted PAV (non-decreasing): blocks of (targets, total) merged while a block's fraction is not above its predecessor's
    bt, bw = [], []
    for t, w in zip(ht, ht + hn):
        bt.append(t)
        bw.append(w)
        while len(bt) > 1 and bt[-2] * bw[-1] >= bt[-1] * bw[-2]:      # frac[-2] >= frac[-1], without dividing
            t2, w2 = bt.pop(), bw.pop()
            bt[-1] += t2
            bw[-1] += w2
    bt, bw = numpy.array(bt), numpy.array(bw)
    left_tar = numpy.concatenate(([0.0], numpy.cumsum(bt)))             # targets at or below each block edge = misses
    left_all = numpy.concatenate(([0.0], numpy.cumsum(bw)))
    pmiss = left_tar / n_tar
    pfa = ((n_tar + n_non) - left_all - (n_tar - left_tar)) / n_non      # non-targets above the edge = false alarms
    return pmiss, pfa


def eer_from_histograms(hist_tar, hist_non):
    """Equal error rate (ROCCH) of binned scores: ``rocch2eer(*rocch_from_histograms(...))``."""
    return rocch2eer(*rocch_from_histograms(hist_tar, hist_non))


def sigmoid(log_odds):
    return 1 / (1 + numpy.exp(-log_odds))


def logit(p):
    p = numpy.asarray(p, dtype=float)
    with numpy.errstate(divide='ignore'):
        return numpy.log(p) - numpy.log(1 - p)


def effective_prior(Ptar, cmiss, cfa):
    p = Ptar * cmiss / (Ptar * cmiss + (1 - Ptar) * cfa)
    return p


def logit_effective_prior(Ptar, cmiss, cfa):
    p = Ptar * cmiss / (Ptar * cmiss + (1 - Ptar) * cfa)
    return float(logit(p))


def fast_minDCF(tar, non, plo, normalize=False):
    """``(minDCF, Pmiss, Pfa, prbep, eer)`` at prior log-odds ``plo`` (``detplot.py:454-511``): the detection cost
    ``P_tar Pmiss + (1 - P_tar) Pfa`` is linear between hull vertices, so its minimum over thresholds is attained at a
    vertex; ``prbep`` is the break-even point of the hull in counts (misses == false alarms)."""
    p_miss, p_fa = rocch(tar, non)
    n_tar, n_non = tar.shape[0], non.shape[0]
    prior_tar, prior_non = sigmoid(plo), sigmoid(-plo)
    cost = numpy.dot(numpy.array([[prior_tar, prior_non]]), numpy.vstack((p_miss, p_fa)))[0]
    best = int(numpy.argmin(cost))
    dcf = cost[best] / min(prior_tar, prior_non) if normalize else cost[best]
    return dcf, p_miss[best], p_fa[best], rocch2eer(p_miss * n_tar, p_fa * n_non), rocch2eer(p_miss, p_fa)
