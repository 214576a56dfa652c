"""EER, Cllr (min / actual) and linkability from a score file and a key file -- ``tools/compute_metrics.py``.

The reference prints three lines (``tools/compute_metrics.py:43-45``) using ``anonymization_metrics.performance`` -- a
repository cloned at install time (``install.sh:98-105``), NOT in the reference tree and not fetchable here.  What this
file holds:

* ``eer_from_files`` -- the EER from the in-tree ``rocch`` / ``rocch2eer`` (``sidekit/bosaris/detplot.py:354-436``), pinned by
  reference fixtures (``tests/test_host_logic.py``);
* ``cllr``, ``min_cllr``, ``linkability`` -- **parity unpinned**: restated from the published definitions that package
  implements, not checked against its code:
    Cllr      = 1/2 [ mean_tar log2(1 + e^-llr) + mean_non log2(1 + e^llr) ]            (Bruemmer & du Preez 2006)
    min Cllr  = Cllr of the PAV-calibrated scores: isotonic fit of the target indicator over the score-sorted trials
                (``pavx``), posterior log-odds minus the log prior odds log(N_tar / N_non), plus a vanishing ramp that
                keeps ties ordered; its ROC convex hull is the one ``rocch`` walks, so the EER that comes with it is the
                ``rocch2eer`` EER                                                        (Bruemmer & de Villiers, BOSARIS)
    D_sys     = integral over scores of D(s) p(s | mated), D(s) = 2 w LR / (1 + w LR) - 1 where w LR > 1 and 0 elsewhere,
                LR = p(s | mated) / p(s | non-mated) from normalised histograms over the common score range
                (min(N_mated / 10, 100) bins by default), D = 1 where only mated scores fall, trapezoid rule over the
                bin centres                                                              (Gomez-Barrero et al. 2017)
  Scores are read as log-likelihood ratios for the actual Cllr, exactly as the reference feeds raw cosine scores to it.

Files: ``enrol test score`` and ``enrol test target|nontarget`` per line.
"""
import argparse

import numpy

from ..bosaris import Key, Scores, pavx, rocch, rocch2eer


def tar_non_from_files(score_file, key_file):
    scores = Scores.read_txt(score_file)
    key = Key.read_txt(key_file)
    tar, non = scores.get_tar_non(key)
    return numpy.asarray(tar, dtype=numpy.float64), numpy.asarray(non, dtype=numpy.float64)


def eer_from_files(score_file, key_file):
    return rocch2eer(*rocch(*tar_non_from_files(score_file, key_file)))


def _neg_log_sigmoid(x):
    """-log(sigmoid(x)) = log(1 + e^-x), stable on both sides; +inf at x = -inf."""
    x = numpy.asarray(x, dtype=numpy.float64)
    with numpy.errstate(over="ignore", invalid="ignore"):
        return numpy.where(x >= 0, numpy.log1p(numpy.exp(-numpy.abs(x))), -x + numpy.log1p(numpy.exp(-numpy.abs(x))))


def cllr(tar_llrs, nontar_llrs):
    """Log-likelihood-ratio cost in bits (parity unpinned, see the module header)."""
    tar_llrs, nontar_llrs = numpy.asarray(tar_llrs, dtype=numpy.float64), numpy.asarray(nontar_llrs, dtype=numpy.float64)
    c1 = _neg_log_sigmoid(tar_llrs).mean() / numpy.log(2.0)
    c2 = _neg_log_sigmoid(-nontar_llrs).mean() / numpy.log(2.0)
    return float((c1 + c2) / 2.0)


def optimal_llr(tar, non, monotonicity_epsilon=1e-6):
    """PAV calibration: the non-decreasing LLR mapping that minimises Cllr on these very scores.  -> (tar_llrs, non_llrs)."""
    tar, non = numpy.asarray(tar, dtype=numpy.float64), numpy.asarray(non, dtype=numpy.float64)
    scores = numpy.concatenate([non, tar])
    pideal = numpy.concatenate([numpy.zeros(non.shape[0]), numpy.ones(tar.shape[0])])
    order = numpy.argsort(scores, kind="mergesort")
    _, width, height = pavx(pideal[order])
    popt = numpy.repeat(height, width)      # the fit itself from its bins (the in-tree pavx's own expanded vector carries the reference's last-element quirk, detplot.py:343-349)
    with numpy.errstate(divide="ignore"):
        llrs = numpy.log(popt) - numpy.log1p(-popt) - numpy.log(tar.shape[0] / non.shape[0])
    n = scores.shape[0]
    llrs = llrs + numpy.arange(n) * monotonicity_epsilon / n        # ties keep their order
    back = numpy.empty(n, dtype=numpy.int64)
    back[order] = numpy.arange(n)
    llrs = llrs[back]
    return llrs[non.shape[0]:], llrs[:non.shape[0]]


def min_cllr(tar, non, compute_eer=False):
    """Cllr after PAV calibration (parity unpinned); with ``compute_eer`` also the ROCCH EER of the same hull."""
    t, n = optimal_llr(tar, non)
    cmin = cllr(t, n)
    if compute_eer:
        return cmin, rocch2eer(*rocch(numpy.asarray(tar, dtype=numpy.float64), numpy.asarray(non, dtype=numpy.float64)))
    return cmin


def linkability(mated, non_mated, omega=1.0, bins=-1):
    """Global linkability D_sys (parity unpinned).  -> (Dsys, D, bin_centers, bin_edges)."""
    mated, non_mated = numpy.asarray(mated, dtype=numpy.float64), numpy.asarray(non_mated, dtype=numpy.float64)
    if bins < 0:
        bins = min(int(mated.shape[0] / 10), 100)
    bins = max(int(bins), 1)
    edges = numpy.linspace(min(mated.min(), non_mated.min()), max(mated.max(), non_mated.max()), num=bins + 1, endpoint=True)
    centers = (edges[1:] + edges[:-1]) / 2.0
    y1 = numpy.histogram(mated, bins=edges, density=True)[0]
    y2 = numpy.histogram(non_mated, bins=edges, density=True)[0]
    lr = numpy.divide(y1, y2, out=numpy.ones_like(y1), where=y2 != 0)
    d = 2.0 * (omega * lr / (1.0 + omega * lr)) - 1.0
    d[omega * lr <= 1.0] = 0.0
    d[(y2 == 0) & (y1 != 0)] = 1.0
    f = d * y1
    dsys = float(numpy.sum((centers[1:] - centers[:-1]) * (f[1:] + f[:-1]) / 2.0))   # trapezoid rule over the bin centres
    return dsys, d, centers, edges


def cli(argv=None):
    parser = argparse.ArgumentParser(description='EER, Cllr and linkability of a score file against a key file')
    parser.add_argument('-s', dest='score_file', type=str, required=True, help='path to score file')
    parser.add_argument('-k', dest='key_file', type=str, required=True, help='path to key file')
    parser.add_argument('--omega', dest='omega', type=float, default=1.0, help='prior ratio (default is 1)')
    parser.add_argument('--bins', dest='bins', type=int, default=-1, help='#bins of the linkability estimate (default min(len(mated) / 10, 100))')
    args = parser.parse_args(argv)
    tar, non = tar_non_from_files(args.score_file, args.key_file)
    cmin, eer = min_cllr(tar, non, compute_eer=True)
    dsys = linkability(tar, non, args.omega, args.bins)[0]
    print("EER: {:.2f}".format(eer * 100))                                  # tools/compute_metrics.py:43-45, the same three lines
    print("Cllr (min/act): %f %f" % (cmin, cllr(tar, non)))
    print("linkability: %f" % dsys)


if __name__ == '__main__':
    cli()
