"""Trial scoring on the GPU through the C ABI: golden reference matrices, oracle at the 1M-trial size, EER."""
import os

import numpy
import pytest
import torch

from oracle import scoring as osc
from sidekit_amd import iv_scoring
from sidekit_amd.bosaris import Key, Ndx, rocch, rocch2eer
from sidekit_amd.statserver import StatServer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx(golden_dir):
    return numpy.load(os.path.join(golden_dir, "scoring.npz"))


def _obj(a):
    return numpy.array([str(x) for x in a], dtype=object)


def _setup(fx):
    enroll = StatServer.from_arrays(_obj(fx["enr_ids"]), _obj(fx["enr_ids"]), fx["E"])
    test = StatServer.from_arrays(_obj(fx["tst_ids"]), _obj(fx["tst_ids"]), fx["T"])
    ndx = Ndx(models=_obj(fx["trial_models"]), testsegs=_obj(fx["trial_segs"]))
    return enroll, test, ndx


def test_cosine_scoring_golden(gpu, fx):
    enroll, test, ndx = _setup(fx)
    before = enroll.stat1.copy()
    sc = iv_scoring.cosine_scoring(enroll, test, ndx)
    assert list(sc.modelset) == list(fx["cos_modelset"]) and list(sc.segset) == list(fx["cos_segset"])
    assert numpy.array_equal(sc.scoremask, fx["cos_scoremask"])
    assert sc.scoremat.dtype == numpy.float32
    numpy.testing.assert_allclose(sc.scoremat, fx["cos_scoremat"], atol=2e-6)
    numpy.testing.assert_array_equal(enroll.stat1, before)                      # inputs are never mutated
    scw = iv_scoring.cosine_scoring(enroll, test, ndx, wccn=fx["wccn"])
    numpy.testing.assert_allclose(scw.scoremat, fx["cos_wccn_scoremat"], atol=2e-6)
    key = Key(models=_obj(fx["trial_models"]), testsegs=_obj(fx["trial_segs"]), trials=_obj(fx["trial_labels"]))
    tar, non = sc.get_tar_non(key)
    assert abs(rocch2eer(*rocch(tar.astype(float), non.astype(float))) - float(fx["cos_eer"])) < 5e-4   # +-0.05 % abs
    with pytest.raises(AssertionError):
        iv_scoring.cosine_scoring(enroll.stat1, test, ndx)
    raw = iv_scoring.cosine_scoring(enroll, test, ndx, check_missing=False)     # no filtering / alignment at all (:92-93)
    assert raw.scoremat.shape == (enroll.stat1.shape[0], test.stat1.shape[0]) and raw.modelset is ndx.modelset


def test_plda_scoring_golden(gpu, fx, caplog):
    enroll, test, ndx = _setup(fx)
    mu, F, G, Sigma = fx["mu"], fx["F"], fx["G"], fx["Sigma"]
    p = iv_scoring.fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma)
    assert p.scoremat.dtype == numpy.float64 and numpy.array_equal(p.scoremask, fx["plda_scoremask"])
    numpy.testing.assert_allclose(p.scoremat, fx["plda_scoremat"], rtol=1e-9, atol=1e-9)
    p = iv_scoring.fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, scaling_factor=0.7)
    numpy.testing.assert_allclose(p.scoremat, fx["plda_scaled_scoremat"], rtol=1e-9, atol=1e-9)
    p = iv_scoring.fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, p_known=0.3)
    numpy.testing.assert_allclose(p.scoremat, fx["plda_open_scoremat"], rtol=1e-9, atol=1e-9)
    p = iv_scoring.PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma)             # dispatcher -> fast
    numpy.testing.assert_allclose(p.scoremat, fx["plda_scoremat"], rtol=1e-9, atol=1e-9)
    p = iv_scoring.PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, full_model=True)
    numpy.testing.assert_allclose(p.scoremat, fx["plda_full_scoremat"], rtol=1e-8, atol=1e-8)
    p = iv_scoring.full_PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, p_known=0.2, scaling_factor=0.9)
    numpy.testing.assert_allclose(p.scoremat, fx["plda_full_open_scoremat"], rtol=1e-8, atol=1e-8)
    # duplicate enrolment models are averaged with a warning (iv_scoring.py:409-411)
    dup = StatServer.from_arrays(_obj(fx["dup_ids"]), _obj([f"e{i:03d}" for i in range(len(fx["dup_ids"]))]), fx["E"])
    with caplog.at_level("WARNING"):
        p = iv_scoring.fast_PLDA_scoring(dup, test, ndx, mu, F, Sigma)
    assert "not unique" in caplog.text
    assert list(p.modelset) == list(fx["plda_dup_modelset"])
    numpy.testing.assert_allclose(p.scoremat, fx["plda_dup_scoremat"], rtol=1e-9, atol=1e-9)
    with pytest.raises(AssertionError, match="dimension mismatch"):
        iv_scoring.PLDA_scoring(enroll, test, ndx, mu, F[:-1], G, Sigma)


def test_mahalanobis_and_two_covariance_golden(gpu, fx):
    """iv_scoring.py:116-213 through the same ``sc_plda_fast`` launch pair; scores of magnitude ~600, expanded quadratic form, so the
    tolerance is relative to that scale (float64 cancellation, not a precision choice)."""
    enroll, test, ndx = _setup(fx)
    e0, t0 = enroll.stat1.copy(), test.stat1.copy()
    m = iv_scoring.mahalanobis_scoring(enroll, test, ndx, fx["maha_M"])
    assert m.scoremat.dtype == numpy.float64 and list(m.modelset) == list(fx["cos_modelset"]) and list(m.segset) == list(fx["cos_segset"])
    assert numpy.array_equal(m.scoremask, fx["plda_scoremask"])
    numpy.testing.assert_allclose(m.scoremat, fx["maha_scoremat"], rtol=1e-11, atol=1e-9)
    t = iv_scoring.two_covariance_scoring(enroll, test, ndx, fx["twocov_W"], fx["twocov_B"])
    numpy.testing.assert_allclose(t.scoremat, fx["twocov_scoremat"], rtol=1e-10, atol=1e-8)
    assert numpy.array_equal(enroll.stat1, e0) and numpy.array_equal(test.stat1, t0)       # the caller's servers are left alone
    # against the oracle's per-model loops on a larger ragged problem, asymmetric M included (only sym(M) matters to the score)
    rs = numpy.random.RandomState(5)
    E, T, M = rs.randn(301, 96), rs.randn(517, 96), rs.randn(96, 96)
    ms = 0.5 * (M + M.T)
    got = iv_scoring.plda_matrix(E, T, -ms, ms, 0.0, 1.0, None)
    numpy.testing.assert_allclose(got, osc.mahalanobis_scores(E, T, M), rtol=1e-11, atol=1e-10)
    with pytest.raises(AssertionError, match="dimension mismatch"):
        iv_scoring.mahalanobis_scoring(enroll, test, ndx, fx["maha_M"][:-1, :-1])


def test_million_trial_matrix_and_eer(gpu):
    """BASELINE config 5 sized: 1000 x 1000 trials, D = 256, synthetic speakers; scores vs the oracle, EER +-0.05 %."""
    rs = numpy.random.RandomState(0)
    n_spk, D, Ne, Nt = 250, 256, 1000, 1000
    c = rs.randn(n_spk, D)
    spk_e, spk_t = rs.randint(0, n_spk, Ne), rs.randint(0, n_spk, Nt)
    norm = lambda x: x / numpy.linalg.norm(x, axis=1, keepdims=True)
    E = norm(c[spk_e] + 1.8 * rs.randn(Ne, D))
    T = norm(c[spk_t] + 1.8 * rs.randn(Nt, D))
    tar_mask = spk_e[:, None] == spk_t[None, :]
    cos = iv_scoring.cosine_matrix(E, T)
    ref = osc.cosine_scores(E, T)
    numpy.testing.assert_allclose(cos, ref, atol=3e-6)
    eer_gpu = rocch2eer(*rocch(cos[tar_mask].astype(float), cos[~tar_mask].astype(float)))
    eer_ref = osc.eer(ref[tar_mask], ref[~tar_mask])
    assert abs(eer_gpu - eer_ref) < 5e-4 and 0.01 < eer_ref < 0.05     # EER ~ 2.9 %
    # PLDA with a synthetic two-covariance model
    mu = 0.05 * rs.randn(D)
    F = rs.randn(D, 128) / numpy.sqrt(D)
    A = rs.randn(D, D) / numpy.sqrt(D)
    Sigma = A.dot(A.T) + 0.5 * numpy.eye(D)
    Phi, Psi, cst = osc.fast_plda_matrices(F, Sigma)
    got = iv_scoring.plda_matrix(E - mu, T - mu, Phi, Psi, cst)
    want = osc.fast_plda_scores(E, T, mu, F, Sigma)
    assert numpy.abs(got - want).max() / numpy.abs(want).max() < 1e-9
    assert abs(osc.eer(got[tar_mask], got[~tar_mask]) - osc.eer(want[tar_mask], want[~tar_mask])) < 5e-4


def test_config5_pinned_by_the_reference_at_full_size(gpu, golden_dir):
    """BASELINE config 5 as stated: 1000 enrolment models x 1000 test segments, PLDA (mu, F, Sigma) trained by the
    reference's FactorAnalyser.plda on the disjoint synthetic set; the fixture holds the REFERENCE's cosine_scoring /
    fast_PLDA_scoring matrices (strided sample, row / column sums, moments), ROCCH vertices and EERs (make_golden.py config5).
    The public API (StatServer + Ndx + Key in, Scores out) on the GPU must reproduce them: 1e-6 relative in f64 (measured 1e-12),
    f32 cosine 2e-6 absolute, EER +-0.05 % absolute (measured: identical)."""
    import sys
    sys.path.insert(0, golden_dir)
    import config5_inputs as c5
    fx = numpy.load(os.path.join(golden_dir, "config5.npz"))
    E, T, spk_e, spk_t = c5.trial_set()
    numpy.testing.assert_array_equal(c5.digest(E), fx["E_digest"])
    numpy.testing.assert_array_equal(c5.digest(T), fx["T_digest"])
    enr_ids, tst_ids = c5.ids("enr", c5.NE), c5.ids("tst", c5.NT)
    enroll, test = StatServer.from_arrays(enr_ids, enr_ids, E), StatServer.from_arrays(tst_ids, tst_ids, T)
    mm, ss = numpy.meshgrid(numpy.arange(c5.NE), numpy.arange(c5.NT), indexing="ij")
    models, segs = enr_ids[mm.ravel()], tst_ids[ss.ravel()]
    tar_mask = spk_e[:, None] == spk_t[None, :]
    ndx = Ndx(models=models, testsegs=segs)
    key = Key(models=models, testsegs=segs, trials=numpy.where(tar_mask.ravel(), "target", "nontarget").astype(object))
    assert ndx.trialmask.all() and numpy.array_equal(key.tar, tar_mask) and int(tar_mask.sum()) == int(fx["n_target"])
    cos = iv_scoring.cosine_scoring(enroll, test, ndx)
    plda = iv_scoring.fast_PLDA_scoring(enroll, test, ndx, fx["mu"], fx["F"], fx["Sigma"])
    for tag, sc in (("cos", cos), ("plda", plda)):
        m = sc.scoremat
        assert str(m.dtype) == str(fx[f"{tag}_dtype"]) and m.shape == (c5.NE, c5.NT) and sc.scoremask.all()
        assert list(sc.modelset) == list(enr_ids) and list(sc.segset) == list(tst_ids)
        m64 = m.astype(numpy.float64)
        if tag == "cos":
            numpy.testing.assert_allclose(m[::7, ::11], fx["cos_sample"], rtol=0, atol=2e-6)
            numpy.testing.assert_allclose(m64.sum(axis=1), fx["cos_row_sums"], rtol=0, atol=1e-4)
            numpy.testing.assert_allclose(m64.sum(axis=0), fx["cos_col_sums"], rtol=0, atol=1e-4)
        else:
            scale = numpy.abs(fx["plda_sample"]).max()
            assert numpy.abs(m[::7, ::11] - fx["plda_sample"]).max() / scale < 1e-9
            numpy.testing.assert_allclose(m64.sum(axis=1), fx["plda_row_sums"], rtol=1e-9, atol=1e-6)
            numpy.testing.assert_allclose(m64.sum(axis=0), fx["plda_col_sums"], rtol=1e-9, atol=1e-6)
        numpy.testing.assert_allclose([m64.mean(), m64.std(), m64.min(), m64.max()], fx[f"{tag}_moments"], rtol=1e-6, atol=2e-6)
        tar, non = sc.get_tar_non(key)
        assert tar.shape[0] == int(fx["n_target"]) and non.shape[0] == c5.NE * c5.NT - int(fx["n_target"])
        pmiss, pfa = rocch(tar.astype(float), non.astype(float))
        assert abs(rocch2eer(pmiss, pfa) - float(fx[f"{tag}_eer"])) < 5e-4          # +-0.05 % absolute (north star)
        if tag == "plda":                                                           # f64 end to end: the hull itself is the reference's
            numpy.testing.assert_allclose(pmiss, fx["plda_pmiss"], atol=1e-12)
            numpy.testing.assert_allclose(pfa, fx["plda_pfa"], atol=1e-12)
            assert abs(rocch2eer(pmiss, pfa) - float(fx["plda_eer"])) < 1e-9


def test_listed_trials_cosine(gpu):
    """compute_spk_cosine.py:18-26,50-55: speaker-mean enrolment, L2, cosine per listed trial (float64 maths)."""
    import ctypes
    from sidekit_amd import _lib
    rs = numpy.random.RandomState(1)
    E = rs.randn(50, 256).astype(numpy.float32)
    T = rs.randn(70, 256).astype(numpy.float32)
    ei = rs.randint(0, 50, 5000).astype(numpy.int32)
    ti = rs.randint(0, 70, 5000).astype(numpy.int32)
    dE, dT = torch.from_numpy(E).cuda(), torch.from_numpy(T).cuda()
    dei, dti = torch.from_numpy(ei).cuda(), torch.from_numpy(ti).cuda()
    out = torch.empty(5000, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().sc_cosine_trials(dE.data_ptr(), dT.data_ptr(), 256, dei.data_ptr(), dti.data_ptr(), 5000, out.data_ptr(),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    e64, t64 = E.astype(numpy.float64), T.astype(numpy.float64)
    ref = (e64[ei] * t64[ti]).sum(1) / (numpy.linalg.norm(e64[ei], axis=1) * numpy.linalg.norm(t64[ti], axis=1))
    numpy.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-12, atol=1e-14)


def test_asnorm_golden_and_large(gpu, golden_dir):
    """Adaptive s-norm (score_normalization.py:120-140): reference fixture, then a cohort with heavy ties at scale."""
    from sidekit_amd.score_normalization import asnorm
    fx = numpy.load(os.path.join(golden_dir, "asnorm.npz"))
    got = asnorm(torch.from_numpy(fx["enrol"]), torch.from_numpy(fx["cohort"]), None)
    assert got.dtype == numpy.float32 and got.shape == (64, 64)
    numpy.testing.assert_allclose(got, fx["snorm"], rtol=2e-5, atol=2e-5)
    rs = numpy.random.RandomState(9)
    e = rs.randn(700, 256).astype(numpy.float32)
    e /= numpy.linalg.norm(e, axis=1, keepdims=True)
    cohort = numpy.round(rs.randn(5000, 256), 1).astype(numpy.float32)       # coarse values -> tied cohort scores
    cohort[100:140] = cohort[100]                                            # 40 identical cohort speakers
    numpy.testing.assert_allclose(asnorm(e, cohort), osc.asnorm(e, cohort), rtol=5e-5, atol=5e-5)
    with pytest.raises(ValueError):
        asnorm(e[:4], cohort[:100], topk=200)                                # k > cohort size


def test_f64_mfma_gemm_ragged_shapes(gpu):
    """sc_plda_fast on sizes that are not multiples of the 64 x 64 x 16 tile (v_mfma_f64_16x16x4_f64 path) vs float64 numpy."""
    rs = numpy.random.RandomState(8)
    # (2900, 2950, 36) takes the 128 x 128-tile kernel (>= 512 such tiles), the others the 64 x 64 one; odd D = scalar loads
    for Ne, Nt, D in ((100, 77, 50), (1, 1, 4), (65, 130, 256), (64, 64, 17), (2900, 2950, 36), (2817, 2900, 33)):
        E, T = rs.randn(Ne, D), rs.randn(Nt, D)
        A = rs.randn(D, D)
        Phi, Psi = -(A @ A.T) / D, rs.randn(D, D) / D
        got = iv_scoring.plda_matrix(E, T, Phi, Psi, 0.37, 0.9)
        want = 0.9 * (0.5 * numpy.einsum("ik,kl,il->i", E, Phi, E)[:, None] + 0.5 * numpy.einsum("jk,kl,jl->j", T, Phi, T)[None, :]
                      + 0.37 + E @ Psi @ T.T)
        assert got.dtype == numpy.float64 and got.shape == (Ne, Nt)
        assert numpy.abs(got - want).max() <= 1e-11 * max(1.0, numpy.abs(want).max()), (Ne, Nt, D)
    # operands that start 8 bytes off a 16-byte boundary (a view into a larger buffer): the 16-byte loads must not be taken
    flat = torch.as_tensor(rs.randn(1 + 70 * 50 + 90 * 50), device=gpu)
    E, T = flat[1:1 + 70 * 50].view(70, 50), flat[1 + 70 * 50:].view(90, 50)
    assert E.data_ptr() % 16 == 8
    Ph, Ps = rs.randn(50, 50) / 50, rs.randn(50, 50) / 50
    got = iv_scoring.plda_matrix_device(E, T, Ph, Ps, 0.1, 1.0).cpu().numpy()
    En, Tn = E.cpu().numpy(), T.cpu().numpy()
    want = (0.5 * numpy.einsum("ik,kl,il->i", En, Ph, En)[:, None] + 0.5 * numpy.einsum("jk,kl,jl->j", Tn, Ph, Tn)[None, :] + 0.1 + En @ Ps @ Tn.T)
    assert numpy.abs(got - want).max() <= 1e-11 * max(1.0, numpy.abs(want).max())


def test_device_resident_scoring(gpu):
    """x-vectors that are already on the GPU are scored there: tensor in, tensor out, same numbers as the numpy entry points."""
    torch.manual_seed(9)
    e = torch.nn.functional.normalize(torch.randn(150, 256, device=gpu), dim=1)
    t = torch.nn.functional.normalize(torch.randn(333, 256, device=gpu), dim=1)
    s = iv_scoring.cosine_matrix_device(e, t)
    assert s.is_cuda and s.dtype == torch.float32 and s.shape == (150, 333)
    assert numpy.array_equal(s.cpu().numpy(), iv_scoring.cosine_matrix(e.cpu().numpy(), t.cpu().numpy()))
    assert float((s - e @ t.T).abs().max()) < 2e-6
    rs = numpy.random.RandomState(10)
    Phi, Psi = rs.randn(256, 256) / 256, rs.randn(256, 256) / 256
    p = iv_scoring.plda_matrix_device(e.double(), t.double(), Phi, Psi, 1.5, 1.0)
    assert p.is_cuda and p.dtype == torch.float64
    assert numpy.array_equal(p.cpu().numpy(), iv_scoring.plda_matrix(e.double().cpu().numpy(), t.double().cpu().numpy(), Phi, Psi, 1.5, 1.0))


def test_cosine_histograms_match_the_score_matrix(gpu):
    """The matrix-free path counts exactly the scores sc_cosine would have written (same MFMA arithmetic, same bins), drops the
    self-trials of a row shard, and its ROCCH EER is that of the binned scores -- within +-0.05 % absolute of the exact EER."""
    from sidekit_amd.bosaris import eer_from_histograms
    rs = numpy.random.RandomState(11)
    n_spk, N = 40, 1000
    lab = rs.randint(0, n_spk, N).astype(numpy.int32)
    c = rs.randn(n_spk, 256)
    X = c[lab] + 1.7 * rs.randn(N, 256)
    X = torch.nn.functional.normalize(torch.as_tensor(X, dtype=torch.float32), dim=1).to(gpu)
    S = iv_scoring.cosine_matrix_device(X, X).cpu().numpy()
    nb = iv_scoring.HIST_BINS
    bins = numpy.clip(numpy.floor((S - numpy.float32(-1.0)) * numpy.float32(nb / 2.0)).astype(numpy.int64), 0, nb - 1)
    tar = lab[:, None] == lab[None, :]
    off = ~numpy.eye(N, dtype=bool)
    ht, hn = iv_scoring.cosine_histograms(X, X, lab, lab, self_offset=0)
    assert ht.dtype == numpy.uint64 and int(ht.sum() + hn.sum()) == N * N - N
    assert numpy.array_equal(ht, numpy.bincount(bins[tar & off], minlength=nb)) and numpy.array_equal(hn, numpy.bincount(bins[~tar], minlength=nb))
    # a row shard of the same set: rows [300, 650) against everything, self-trials at j == i + 300
    a, b = 300, 650
    ht2, hn2 = iv_scoring.cosine_histograms(X[a:b], X, lab[a:b], lab, self_offset=a)
    assert numpy.array_equal(ht2, numpy.bincount(bins[a:b][(tar & off)[a:b]], minlength=nb))
    assert numpy.array_equal(hn2, numpy.bincount(bins[a:b][~tar[a:b]], minlength=nb))
    ht3, hn3 = iv_scoring.cosine_histograms(X[:77], X[100:], lab[:77], lab[100:])          # disjoint sets: nothing dropped
    assert int(ht3.sum() + hn3.sum()) == 77 * 900
    eer_h = eer_from_histograms(ht, hn)
    eer_x = rocch2eer(*rocch(S[tar & off].astype(float), S[~tar].astype(float)))
    assert 0.01 < eer_x < 0.4 and abs(eer_h - eer_x) < 5e-4, (eer_h, eer_x)
    # finer bins (round 6, tests/test_gpu_eer_dtype.py): `bins` a multiple of HIST_BINS - 2 = that many passes of the kernel over slices of [lo, hi)
    # with one guard bin either side; scores outside the range land in the end bins.  Against numpy on the score matrix with the same edges: a
    # score within float32 rounding of an edge may sit in the neighbouring bin (the kernel bins (s - lo_pass) * scale in float32; a bin is 4e-5 wide,
    # so about one score in 400 is that close to an edge) -- each such score moves ONE value of the cumulative counts by one, hence the bound on the
    # cumulative difference (observed: 4 of 10^6 pairs), not on the bins
    lo, hi, nf = float(S[off].min()) + 0.05, float(S[off].max()) - 0.02, 3 * (nb - 2)      # some scores below lo and above hi
    hft, hfn = iv_scoring.cosine_histograms(X, X, lab, lab, self_offset=0, lo=lo, hi=hi, bins=nf)
    assert hft.shape == hfn.shape == (nf,) and int(hft.sum() + hfn.sum()) == N * N - N
    want = numpy.clip(numpy.floor((S.astype(numpy.float64) - lo) * (nf / (hi - lo))).astype(numpy.int64), 0, nf - 1)
    for got, sel in ((hft, tar & off), (hfn, ~tar)):
        ref = numpy.bincount(want[sel], minlength=nf)
        assert int(numpy.abs(numpy.cumsum(got.astype(numpy.int64)) - numpy.cumsum(ref)).max()) <= 16 and int(got[0]) >= int((S[sel] < lo).sum()) - 16
    assert abs(eer_from_histograms(hft, hfn) - eer_x) < 5e-4
    with pytest.raises(AssertionError):
        iv_scoring.cosine_histograms(X, X, lab, lab, bins=10000)


def test_sharded_driver_single_rank(gpu, capsys):
    """configs 3 + 5 as one rank: synthetic-speaker corpus -> x-vectors -> gather -> cosine + fast PLDA on the gathered x-vectors'
    own trials, sharded by enrolment rows and device resident, plus the matrix-free all-pairs histogram path."""
    import json
    from sidekit_amd.bin import shard_extract_score
    shard_extract_score.main(["--utterances", "1600", "--batch", "160", "--seconds", "1", "--trials", "500", "--speakers", "40",
                              "--plda-rank", "32", "--all-pairs"])
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ranks"] == 1 and d["trials"] == 250000 and d["all_pairs"] == 1600 * 1599
    for k in ("cosine_eer", "plda_eer", "all_pairs_eer"):
        assert 0.0 <= d[k] < 0.3, (k, d[k])                        # the random-weight extractor separates the synthetic speakers
    assert d["plda_eer"] < d["cosine_eer"] + 0.02                  # PLDA trained on the held-out part of the same x-vectors is no worse
    assert abs(d["cosine_eer"] - d["all_pairs_eer"]) < 0.05        # same score distribution, different trial subsets
