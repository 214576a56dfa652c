"""Oracle trial scoring and ROCCH-EER on plain numpy arrays.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Restates the arithmetic of ``sidekit/iv_scoring.py:98-109`` (cosine), ``:428-475`` (fast PLDA),
``:303-366`` (full PLDA), ``sidekit/statserver.py:797-817`` (norm / centre) and
``sidekit/bosaris/detplot.py:289-436`` (pavx, rocch2eer, rocch), ``sidekit/score_normalization.py:120-140`` (asnorm).  The trial bookkeeping
(``Ndx.filter``, ``align_*``) is not restated here: it is pinned directly by the fixtures in
``tests/golden/scoring.npz`` made with the imported reference.
"""
import numpy
import scipy.linalg


def norm_rows(x):
    """StatServer.norm_stat1, statserver.py:797-800 (float64, norms clipped at 1e-8)."""
    x = numpy.asarray(x, dtype=numpy.float64)
    n = numpy.clip(numpy.linalg.norm(x, axis=1), 1e-08, numpy.inf)
    return (x.transpose() / n).transpose()


def cosine_scores(enroll, test):
    """iv_scoring.py:98-109: rows normalised in float64, product in float32."""
    e = norm_rows(enroll).astype(numpy.float32)
    t = norm_rows(test).astype(numpy.float32)
    return numpy.einsum('ij,kj', e, t)


def _open_set(scoremat, p_known):
    """iv_scoring.py:467-475 / :356-364."""
    N = scoremat.shape[0]
    out = numpy.empty(scoremat.shape)
    tmp = numpy.exp(scoremat)
    for ii in range(N):
        out[ii, :] = scoremat[ii, :] - numpy.log(p_known * tmp[~(numpy.arange(N) == ii)].sum(axis=0) / (N - 1) + (1 - p_known))
    return out


def fast_plda_matrices(F, Sigma, scaling_factor=1.0):
    """iv_scoring.py:428-446 -> (Phi, Psi, plda_cst)."""
    invSigma = scipy.linalg.inv(Sigma)
    I_spk = numpy.eye(F.shape[1], dtype='float')
    K = F.T.dot(invSigma * scaling_factor).dot(F)
    K1 = scipy.linalg.inv(K + I_spk)
    K2 = scipy.linalg.inv(2 * K + I_spk)
    alpha1 = numpy.linalg.slogdet(K1)[1]
    alpha2 = numpy.linalg.slogdet(K2)[1]
    plda_cst = alpha2 / 2.0 - alpha1
    Sigma_ac = numpy.dot(F, F.T)
    Sigma_tot = Sigma_ac + Sigma
    Sigma_tot_inv = scipy.linalg.inv(Sigma_tot)
    Tmp = numpy.linalg.inv(Sigma_tot - Sigma_ac.dot(Sigma_tot_inv).dot(Sigma_ac))
    Phi = Sigma_tot_inv - Tmp
    Psi = Sigma_tot_inv.dot(Sigma_ac).dot(Tmp)
    return Phi, Psi, plda_cst


def fast_plda_scores(enroll, test, mu, F, Sigma, p_known=0.0, scaling_factor=1.0):
    """iv_scoring.py:421-475 on already aligned (Ne, D) / (Nt, D) float64 vectors (stat0 == 1)."""
    e = numpy.asarray(enroll, dtype=numpy.float64) - mu
    t = numpy.asarray(test, dtype=numpy.float64) - mu
    Phi, Psi, cst = fast_plda_matrices(F, Sigma, scaling_factor)
    model_part = 0.5 * numpy.einsum('ij, ji->i', e.dot(Phi), e.T)
    seg_part = 0.5 * numpy.einsum('ij, ji->i', t.dot(Phi), t.T)
    s = model_part[:, numpy.newaxis] + seg_part + cst
    s += e.dot(Psi).dot(t.T)
    s *= scaling_factor
    return _open_set(s, p_known) if p_known != 0 else s


def full_plda_scores(enroll, test, mu, F, G, Sigma, p_known=0.0, scaling_factor=1.0):
    """iv_scoring.py:299-366 (per-model loop kept as in the reference)."""
    e = numpy.asarray(enroll, dtype=numpy.float64) - mu
    t = numpy.asarray(test, dtype=numpy.float64) - mu
    invSigma = scipy.linalg.inv(Sigma)
    I_iv = numpy.eye(mu.shape[0], dtype='float')
    I_ch = numpy.eye(G.shape[1], dtype='float')
    I_spk = numpy.eye(F.shape[1], dtype='float')
    A = numpy.linalg.inv(G.T.dot(invSigma * scaling_factor).dot(G) + I_ch)
    B = F.T.dot(invSigma * scaling_factor).dot(I_iv - G.dot(A).dot(G.T).dot(invSigma * scaling_factor))
    K = B.dot(F)
    K1 = scipy.linalg.inv(K + I_spk)
    K2 = scipy.linalg.inv(2 * K + I_spk)
    constant = numpy.linalg.slogdet(K2)[1] / 2.0 - numpy.linalg.slogdet(K1)[1]
    test_tmp = B.dot(t.T)
    enroll_tmp = B.dot(e.T)
    tmp1 = test_tmp.T.dot(K1)
    S1 = numpy.array([tmp1[i, :].dot(test_tmp[:, i]) / 2. for i in range(t.shape[0])])
    S2 = numpy.empty(e.shape[0])
    s = numpy.zeros((e.shape[0], t.shape[0]))
    for m in range(e.shape[0]):
        both = test_tmp + numpy.atleast_2d(enroll_tmp[:, m]).T
        tmp2 = both.T.dot(K2)
        S2[m] = enroll_tmp[:, m].dot(K1).dot(enroll_tmp[:, m]) / 2.
        s[m, :] = numpy.einsum("ij, ji->i", tmp2, both) / 2.
    s += constant - (S1 + S2[:, numpy.newaxis])
    s *= scaling_factor
    return _open_set(s, p_known) if p_known != 0 else s


def mahalanobis_scores(enroll, test, m):
    """iv_scoring.py:145-149, the reference's own loop: per model, ``-0.5 * sum((t - e) M * (t - e))`` over the test rows."""
    s = numpy.zeros((enroll.shape[0], test.shape[0]))
    for i in range(enroll.shape[0]):
        d = test - enroll[i]
        s[i] = -0.5 * numpy.sum(d.dot(m) * d, axis=1)
    return s


def two_covariance_scores(enroll, test, W, B):
    """iv_scoring.py:194-205 (column vectors there, rows here): ``(e + t)' G (e + t) - t' H t - e' H e``."""
    iW, iB = scipy.linalg.inv(W), scipy.linalg.inv(B)
    G = iW.dot(scipy.linalg.inv(iB + 2 * iW)).dot(iW)
    H = iW.dot(scipy.linalg.inv(iB + iW)).dot(iW)
    s2 = numpy.sum(test.dot(H) * test, axis=1)
    s3 = numpy.sum(enroll.dot(H) * enroll, axis=1)
    s = numpy.zeros((enroll.shape[0], test.shape[0]))
    for i in range(enroll.shape[0]):
        a = test + enroll[i]
        s[i] = numpy.sum(a.dot(G) * a, axis=1) - s2 - s3[i]
    return s


def asnorm(enrol_xv, cohort_xv, topk=200):
    """score_normalization.py:120-140 (torch, float32)."""
    import torch
    enrol_xv = torch.as_tensor(enrol_xv, dtype=torch.float32)
    cohort_xv = torch.nn.functional.normalize(torch.as_tensor(cohort_xv, dtype=torch.float32), dim=1)
    enrol_test_scores = torch.einsum('ij,kj', enrol_xv, enrol_xv).numpy()
    calib_scores = torch.einsum('ij,kj', enrol_xv, cohort_xv)
    topk_cohort = calib_scores.topk(topk, dim=1).values
    calib_mean = topk_cohort.mean(dim=1).numpy()
    calib_std = topk_cohort.std(dim=1).numpy()
    return 0.5 * ((enrol_test_scores.T - calib_mean) / calib_std).T + 0.5 * (enrol_test_scores - calib_mean) / calib_std


def pavx(y):
    """detplot.py:289-351 pool-adjacent-violators: returns (ghat, width, height)."""
    assert y.ndim == 1 and y.shape[0] > 0
    n = y.shape[0]
    ghat = numpy.zeros(n)
    length = numpy.zeros(n, dtype=int)
    ci = 0
    length[0] = 1
    ghat[0] = y[0]
    for j in range(1, n):
        ci += 1
        length[ci] = 1
        ghat[ci] = y[j]
        while ci >= 1 and ghat[ci - 1] >= ghat[ci]:
            nw = length[ci - 1] + length[ci]
            ghat[ci - 1] = ghat[ci - 1] + (length[ci] / nw) * (ghat[ci] - ghat[ci - 1])
            length[ci - 1] = nw
            ci -= 1
    height = ghat[:ci + 1].copy()
    width = length[:ci + 1].copy()
    full = numpy.repeat(height, width)
    # reference quirk (detplot.py:343-349): the fill loop of the FIRST bin starts at j = index[0] = 0 and so also
    # writes ghat[-1] = height[0]; kept so the restatement returns what the reference returns (rocch only uses width)
    full[-1] = height[0]
    return full, width, height


def rocch(tar_scores, nontar_scores):
    """detplot.py:390-436: ROC convex hull vertices (pmiss, pfa)."""
    Nt, Nn = tar_scores.shape[0], nontar_scores.shape[0]
    N = Nt + Nn
    scores = numpy.concatenate((tar_scores, nontar_scores))
    Pideal = numpy.concatenate((numpy.ones(Nt), numpy.zeros(Nn)))
    perturb = numpy.argsort(scores, kind='mergesort')  # stable: ties keep target-before-nontarget order
    Pideal = Pideal[perturb]
    _, width, _ = pavx(Pideal)
    nbins = width.shape[0]
    pmiss = numpy.zeros(nbins + 1)
    pfa = numpy.zeros(nbins + 1)
    left, fa, miss = 0, Nn, 0
    for i in range(nbins):
        pmiss[i] = miss / Nt
        pfa[i] = fa / Nn
        left = int(left + width[i])
        miss = numpy.sum(Pideal[:left])
        fa = N - left - numpy.sum(Pideal[left:])
    pmiss[nbins] = miss / Nt
    pfa[nbins] = fa / Nn
    return pmiss, pfa


def rocch2eer(pmiss, pfa):
    """detplot.py:354-387."""
    eer = 0
    for i in range(pfa.shape[0] - 1):
        xx = pfa[i:i + 2]
        yy = pmiss[i:i + 2]
        assert (xx[1] <= xx[0]) & (yy[0] <= yy[1]), 'pmiss and pfa have to be sorted'
        XY = numpy.column_stack((xx, yy))
        dd = numpy.dot(numpy.array([1, -1]), XY)
        if numpy.min(numpy.abs(dd)) == 0:
            eerseg = 0
        else:
            seg = numpy.linalg.solve(XY, numpy.array([[1], [1]]))
            eerseg = 1 / (numpy.sum(seg))
        eer = max([eer, eerseg])
    return eer


def eer(tar, non):
    return rocch2eer(*rocch(numpy.asarray(tar, dtype=numpy.float64), numpy.asarray(non, dtype=numpy.float64)))
