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
    """Equal error rate from the hull vertices: the highest intersection of a hull segment with the
    diagonal pmiss == pfa."""
    eer = 0
    for i in range(pfa.shape[0] - 1):
        xx, yy = pfa[i:i + 2], pmiss[i:i + 2]
        assert (xx[1] <= xx[0]) & (yy[0] <= yy[1]), 'pmiss and pfa have to be sorted'
        XY = numpy.column_stack((xx, yy))
        dd = numpy.dot(numpy.array([1, -1]), XY)
        if numpy.min(numpy.abs(dd)) == 0:
            eerseg = 0
        else:
            # the segment's line a*x + b*y = 1 meets x == y at 1 / (a + b)
            seg = numpy.linalg.solve(XY, numpy.array([[1], [1]]))
            eerseg = 1 / (numpy.sum(seg))
        eer = max([eer, eerseg])
    return eer


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
    """``(minDCF, Pmiss, Pfa, prbep, eer)`` at prior log-odds ``plo``."""
    Pmiss, Pfa = rocch(tar, non)
    Nmiss = Pmiss * tar.shape[0]
    Nfa = Pfa * non.shape[0]
    prbep = rocch2eer(Nmiss, Nfa)
    eer = rocch2eer(Pmiss, Pfa)
    Ptar, Pnon = sigmoid(plo), sigmoid(-plo)
    cdet = numpy.dot(numpy.array([[Ptar, Pnon]]), numpy.vstack((Pmiss, Pfa)))
    ii = numpy.argmin(cdet, axis=1)
    minDCF = cdet[0, ii][0]
    if normalize:
        minDCF = minDCF / min([Ptar, Pnon])
    return minDCF, Pmiss[ii][0], Pfa[ii][0], prbep, eer
