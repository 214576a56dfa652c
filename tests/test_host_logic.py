"""Host-side mirror classes (bosaris / StatServer / weights / Xtractor surface) -- CPU only."""
import logging
import os

import numpy
import pytest
import torch

from sidekit_amd.bosaris import IdMap, Key, Ndx, Scores, pavx, rocch, rocch2eer, fast_minDCF
from sidekit_amd.nnet import Xtractor
from sidekit_amd.nnet import preprocessor as pp
from sidekit_amd.nnet.weights import seeded_state_dict, state_dict_spec
from sidekit_amd.statserver import StatServer
from oracle import frontend as ofe


@pytest.fixture(scope="module")
def fx(golden_dir):
    return numpy.load(os.path.join(golden_dir, "scoring.npz"))


def _obj(a):
    return numpy.array([str(x) for x in a], dtype=object)


def _sts(models, segs, X):
    return StatServer.from_arrays(_obj(models), _obj(segs), X)


def test_ndx_matches_reference(fx):
    ndx = Ndx(models=_obj(fx["trial_models"]), testsegs=_obj(fx["trial_segs"]))
    assert list(ndx.modelset) == list(fx["ndx_modelset"]) and list(ndx.segset) == list(fx["ndx_segset"])
    assert numpy.array_equal(ndx.trialmask, fx["ndx_trialmask"])
    clean = ndx.filter(_obj(fx["enr_ids"]), _obj(fx["tst_ids"]), True)
    assert list(clean.modelset) == list(fx["cos_modelset"]) and list(clean.segset) == list(fx["cos_segset"])
    assert numpy.array_equal(clean.trialmask, fx["cos_scoremask"])
    dropped = ndx.filter(["ghost_a", "ghost_b"], ["ghost_seg"], False)
    assert "ghost_a" not in dropped.modelset and "ghost_seg" not in dropped.segset and dropped.validate()


def test_ndx_text_roundtrip(tmp_path, fx):
    ndx = Ndx(models=_obj(fx["trial_models"]), testsegs=_obj(fx["trial_segs"]))
    p = tmp_path / "trials.ndx"
    ndx.save_txt(str(p))
    back = Ndx(str(p))
    assert list(back.modelset) == list(ndx.modelset) and numpy.array_equal(back.trialmask, ndx.trialmask)


def test_key_and_scores_match_reference(fx, tmp_path):
    key = Key(models=_obj(fx["trial_models"]), testsegs=_obj(fx["trial_segs"]), trials=_obj(fx["trial_labels"]))
    assert list(key.modelset) == list(fx["key_modelset"]) and list(key.segset) == list(fx["key_segset"])
    assert numpy.array_equal(key.tar, fx["key_tar"]) and numpy.array_equal(key.non, fx["key_non"])
    sc = Scores()
    sc.modelset, sc.segset = _obj(fx["cos_modelset"]), _obj(fx["cos_segset"])
    sc.scoremat, sc.scoremask = fx["cos_scoremat"], fx["cos_scoremask"]
    aligned = sc.align_with_ndx(key)
    assert numpy.array_equal(aligned.scoremask, fx["aligned_scoremask"])
    numpy.testing.assert_array_equal(aligned.scoremat, fx["aligned_scoremat"])
    tar, non = sc.get_tar_non(key)   # shapes differ -> align branch (works here; raises in the reference under numpy >= 1.25)
    numpy.testing.assert_array_equal(tar, fx["cos_tar"])
    numpy.testing.assert_array_equal(non, fx["cos_non"])
    ok = numpy.array([m in set(sc.modelset) and s in set(sc.segset) for m, s in zip(fx["trial_models"], fx["trial_segs"])])
    key2 = Key(models=_obj(fx["trial_models"][ok]), testsegs=_obj(fx["trial_segs"][ok]), trials=_obj(fx["trial_labels"][ok]))
    tar2, non2 = sc.get_tar_non(key2)
    numpy.testing.assert_array_equal(tar2, fx["cos_tar2"])
    numpy.testing.assert_array_equal(non2, fx["cos_non2"])
    # text formats of compute_spk_cosine.py / compute_metrics.py
    sc.write_txt(str(tmp_path / "scores.txt"))
    back = Scores.read_txt(str(tmp_path / "scores.txt"))
    assert numpy.array_equal(back.scoremask, sc.scoremask)
    numpy.testing.assert_allclose(back.scoremat[back.scoremask], sc.scoremat[sc.scoremask].astype(float), rtol=1e-7)
    key.write_txt(str(tmp_path / "key.txt"))
    kb = Key.read_txt(str(tmp_path / "key.txt"))
    assert numpy.array_equal(kb.tar, key.tar) and numpy.array_equal(kb.non, key.non)
    assert numpy.array_equal(key.to_ndx().trialmask, key.tar | key.non)


def test_statserver_algebra_matches_reference(fx, caplog):
    ids = fx["enr_ids"]
    s = _sts(ids, ids, fx["E"])
    assert s.validate()
    s.norm_stat1()
    numpy.testing.assert_array_equal(s.stat1, fx["norm_stat1"])
    s = _sts(ids, ids, fx["E"])
    s.center_stat1(fx["mu"])
    numpy.testing.assert_array_equal(s.stat1, fx["center_stat1"])
    s = _sts(ids, ids, fx["E"])
    s.whiten_stat1(fx["mu"], fx["Sigma"])
    numpy.testing.assert_allclose(s.stat1, fx["whiten_stat1"], rtol=1e-10, atol=1e-10)
    d = _sts(fx["dup_ids"], [f"e{i:03d}" for i in range(len(ids))], fx["E"]).mean_stat_per_model()
    assert list(d.modelset) == list(fx["mean_per_model_modelset"])
    numpy.testing.assert_allclose(d.stat1, fx["mean_per_model_stat1"], rtol=1e-15, atol=0)
    # alignment keeps the first occurrence and follows the requested order
    s = _sts(ids, ids, fx["E"])
    want = _obj(sorted(ids)[:5])
    s.align_models(want)
    assert list(s.modelset) == list(want)
    em = {m: i for i, m in enumerate(ids)}
    numpy.testing.assert_array_equal(s.stat1, fx["E"][[em[m] for m in want]])
    with pytest.raises(IndexError):
        s.align_segments(_obj(["nope"]))


def test_pav_and_rocch_host_code_bit_exact(fx):
    g, w, h = pavx(fx["pav_y"])
    assert numpy.array_equal(g, fx["pav_ghat"]) and numpy.array_equal(w, fx["pav_width"]) and numpy.array_equal(h, fx["pav_height"])
    pm, pf = rocch(fx["rocch_tar"], fx["rocch_non"])
    assert numpy.array_equal(pm, fx["rocch_pmiss"]) and numpy.array_equal(pf, fx["rocch_pfa"])
    assert rocch2eer(pm, pf) == float(fx["rocch_eer"])
    pm2, pf2 = rocch(fx["cos_tar"], fx["cos_non"])
    assert rocch2eer(pm2, pf2) == float(fx["cos_eer"])
    out = fast_minDCF(fx["rocch_tar"], fx["rocch_non"], 0.0)
    assert out[4] == float(fx["rocch_eer"]) and 0 <= out[0] <= 0.5
    # edge cases: perfectly separated, fully overlapping, single scores
    assert rocch2eer(*rocch(numpy.array([2.0, 3.0]), numpy.array([0.0, 1.0]))) == 0
    assert abs(rocch2eer(*rocch(numpy.array([0.0, 1.0]), numpy.array([0.0, 1.0]))) - 0.5) < 1e-12
    assert rocch2eer(*rocch(numpy.array([1.0]), numpy.array([1.0]))) in (0, 0.5, 1.0)
    with pytest.raises(AssertionError):
        pavx(numpy.zeros(0))


def test_idmap():
    im = IdMap()
    im.set(_obj(["a", "a", "b"]), _obj(["s1", "s2", "s3"]))
    assert im.validate()
    assert list(im.filter_on_left(["a"], True).rightids) == ["s1", "s2"]
    assert list(im.filter_on_right(["s1"], False).rightids) == ["s2", "s3"]
    st = StatServer(im, distrib_nb=1, feature_size=4)
    assert st.stat1.shape == (3, 4) and st.validate()


def test_checkpoint_layout_and_strict_loading():
    spec = state_dict_spec("halfresnet34", 7205)
    assert len(spec) == 273
    assert spec["stat_pooling.attention.0.weight"][0] == (128, 7680, 1)   # SURVEY F1'
    assert spec["sequence_network.layer1.0.shortcut.0.weight"][0] == (32, 32, 1, 1)
    n_par = sum(int(numpy.prod(s)) for k, (s, kind) in spec.items() if kind in ("conv", "linear", "bias", "bn_w", "bn_b"))
    assert n_par == 5363744 + 1313664 + 1311232 + 1844480           # trunk + pooling + embedding + AAM head
    m = Xtractor(16, model_archi="halfresnet34", loss="aam", seed=1)
    sd = seeded_state_dict("halfresnet34", 16, seed=2)
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.state_dict()["after_speaker_embedding.weight"], sd["after_speaker_embedding.weight"])
    assert torch.equal(m.after_speaker_embedding.weight, sd["after_speaker_embedding.weight"])
    bad = dict(sd)
    bad.pop("sequence_network.conv1.weight")
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["extra.key"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["before_speaker_embedding.lin_be.weight"] = torch.zeros(256, 61440)   # what AttentivePooling(256, 80) would imply
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad, strict=True)
    assert m.context_size() == 3 and Xtractor(4, "xvector", "cce", seed=0).context_size() == 15
    with pytest.raises(NotImplementedError):
        Xtractor(4, "xvector", loss="nope")
    with pytest.raises(NotImplementedError):
        Xtractor(4, "resnet34", loss="aam")
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(16000), is_eval=True)                          # model still on CPU: no fallback
    with pytest.raises(NotImplementedError):
        m.to("cpu")(torch.zeros(16000))                             # is_eval=False (training) is out of scope


def test_product_frontend_constants_equal_oracle():
    b = pp.MelSpecFrontEnd().buffers()
    assert torch.equal(b["MelSpec.mel_scale.fb"], ofe.mel_filterbank(513, 90, 7600, 80, 16000))
    assert torch.allclose(b["MelSpec.spectrogram.window"], ofe.hann_window(400), atol=1e-7)
    b = pp.MfccFrontEnd().buffers()
    assert torch.equal(b["MFCC.MelSpectrogram.mel_scale.fb"], ofe.mel_filterbank(1025, 133.333, 6855.4976, 100, 16000))
    assert torch.equal(b["MFCC.dct_mat"], ofe.dct_matrix(80, 100))


def test_histogram_eer_is_the_rocch_eer_of_the_binned_scores():
    from sidekit_amd.bosaris import eer_from_histograms, rocch_from_histograms
    rs = numpy.random.RandomState(0)
    tar = numpy.clip(rs.randn(3000) * 0.15 + 0.45, -1, 0.9999)
    non = numpy.clip(rs.randn(40000) * 0.12, -1, 0.9999)
    nb = 8192
    bt, bn = numpy.floor((tar + 1) * nb / 2).astype(int), numpy.floor((non + 1) * nb / 2).astype(int)
    ht, hn = numpy.bincount(bt, minlength=nb), numpy.bincount(bn, minlength=nb)
    eer_h = eer_from_histograms(ht, hn)
    assert abs(eer_h - rocch2eer(*rocch(bt.astype(float), bn.astype(float)))) < 1e-15       # ties with multiplicities == weighted bins
    assert abs(eer_h - rocch2eer(*rocch(tar, non))) < 5e-4                                    # +-0.05 % absolute of the exact EER
    pm, pf = rocch_from_histograms(ht, hn)
    assert pm[0] == 0 and pf[0] == 1 and pm[-1] == 1 and pf[-1] == 0 and numpy.all(numpy.diff(pm) >= 0) and numpy.all(numpy.diff(pf) <= 0)
    # perfectly separated and fully mixed histograms
    a, b = numpy.zeros(16), numpy.zeros(16)
    a[12], b[3] = 5, 7
    assert eer_from_histograms(a, b) == 0
    a[:], b[:] = 1, 1
    assert abs(eer_from_histograms(a, b) - 0.5) < 1e-12


def test_shard_extract_score_load_plda(tmp_path, golden_dir):
    """`bin/shard_extract_score.load_plda`: (mu, F, Sigma) from the config-5 fixture (.npz, the reference's FactorAnalyser.plda output) and
    from a SIDEKIT PLDA HDF5 file written by `sidekit_io.write_plda_hdf5` -- the same arrays either way."""
    import os
    from sidekit_amd import sidekit_io
    from sidekit_amd.bin.shard_extract_score import load_plda
    fx = numpy.load(os.path.join(golden_dir, "config5.npz"))
    mu, F, Sigma = load_plda(os.path.join(golden_dir, "config5.npz"))
    assert mu.shape == (256,) and F.shape == (256, 128) and Sigma.shape == (256, 256) and mu.dtype == numpy.float64
    numpy.testing.assert_array_equal(F, fx["F"])
    sidekit_io.write_plda_hdf5((mu, F, numpy.zeros((256, 0)), Sigma), str(tmp_path / "plda.h5"))
    mu2, F2, Sigma2 = load_plda(str(tmp_path / "plda.h5"))
    numpy.testing.assert_array_equal(mu2, mu)
    numpy.testing.assert_array_equal(F2, F)
    numpy.testing.assert_array_equal(Sigma2, Sigma)


def test_compute_metrics_three_lines(tmp_path, capsys):
    """tools/compute_metrics.py:43-45 prints EER, Cllr (min / act) and linkability.  The EER is the pinned in-tree ROCCH EER; Cllr and
    linkability restate anonymization_metrics (un-vendored: PARITY UNPINNED) and are checked against closed forms and invariants."""
    from sidekit_amd.bin import compute_metrics as cm
    rs = numpy.random.RandomState(3)
    # well-calibrated Gaussian LLRs: tar ~ N(+m, 2m), non ~ N(-m, 2m)  =>  actual Cllr ~ min Cllr, both < 1 bit
    m = 2.0
    tar, non = rs.normal(m, numpy.sqrt(2 * m), 4000), rs.normal(-m, numpy.sqrt(2 * m), 6000)
    c_act, (c_min, eer) = cm.cllr(tar, non), cm.min_cllr(tar, non, compute_eer=True)
    assert 0.0 < c_min <= c_act + 1e-12 < 1.0 and c_act - c_min < 0.02          # calibrated: PAV cannot gain much
    assert abs(cm.cllr(numpy.zeros(5), numpy.zeros(7)) - 1.0) < 1e-12               # llr = 0 everywhere: exactly one bit
    assert abs(cm.cllr([numpy.log(3.0)], [-numpy.log(3.0)]) - numpy.log2(4.0 / 3.0)) < 1e-12
    assert cm.cllr([-numpy.inf], [0.0]) == numpy.inf
    # min Cllr is invariant under any increasing map of the scores, the actual Cllr is not; the EER with it is the ROCCH EER
    c_min2, eer2 = cm.min_cllr(numpy.tanh(tar / 10) * 7 + 3, numpy.tanh(non / 10) * 7 + 3, compute_eer=True)
    assert abs(c_min2 - c_min) < 1e-9 and abs(eer2 - eer) < 1e-12
    from oracle import scoring as osc
    assert abs(eer - osc.eer(tar, non)) < 1e-12
    # separable scores: min Cllr = 0, EER = 0, every mated score fully linkable
    assert cm.min_cllr([2.0, 3.0, 4.0], [-1.0, 0.0, 1.0]) < 1e-5
    d_sep = cm.linkability(rs.normal(10, 1, 2000), rs.normal(-10, 1, 2000))[0]
    d_same = cm.linkability(rs.normal(0, 1, 20000), rs.normal(0, 1, 20000))[0]
    d_mid = cm.linkability(tar, non)[0]
    assert 0.95 < d_sep <= 1.0 + 1e-9 and 0.0 <= d_same < 0.12 and d_same < d_mid < d_sep
    # the CLI: the reference's text formats in, its three lines out
    models, segs = [f"m{i}" for i in range(20)], [f"s{j}" for j in range(30)]
    with open(tmp_path / "scores", "w") as fs, open(tmp_path / "key", "w") as fk:
        for i, e in enumerate(models):
            for j, t in enumerate(segs):
                is_tar = (i % 10) == (j % 10)
                fs.write(f"{e} {t} {(2.0 if is_tar else -2.0) + rs.randn():.6f}\n")
                fk.write(f"{e} {t} {'target' if is_tar else 'nontarget'}\n")
    cm.cli(["-s", str(tmp_path / "scores"), "-k", str(tmp_path / "key")])
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 3 and out[0].startswith("EER: ") and out[1].startswith("Cllr (min/act): ") and out[2].startswith("linkability: ")
    assert abs(float(out[0].split()[1]) / 100 - cm.eer_from_files(str(tmp_path / "scores"), str(tmp_path / "key"))) < 5e-5
    cmin_s, cact_s = (float(x) for x in out[1].split()[2:4])
    assert 0.0 <= cmin_s <= cact_s and 0.0 <= float(out[2].split()[1]) <= 1.0


def test_plda_host_algebra_against_the_oracle(golden_dir):
    """``iv_scoring.plda_parameters`` / ``full_plda_parameters`` (the D x D float64 algebra that stays on the host, ``sidekit/iv_scoring.py:299-330,
    428-446``) are written from the model (Schur complement of the pair covariance; Woodbury for the channel term), the oracle follows the
    reference's statements: the same matrices to float64 round-off on the reference-trained config-5 parameters, and the same scores when
    the kernel's form ``s (0.5 e'Phi e + 0.5 t'Phi t + c + e'Psi t)`` is evaluated in numpy."""
    from oracle import scoring as osc
    from sidekit_amd import iv_scoring
    z = numpy.load(os.path.join(golden_dir, "config5.npz"))
    mu, F, Sigma = (numpy.asarray(z[k], dtype=numpy.float64) for k in ("mu", "F", "Sigma"))
    for scaling in (1.0, 0.7):
        Phi, Psi, cst = iv_scoring.plda_parameters(mu, F, Sigma, scaling)
        rPhi, rPsi, rcst = osc.fast_plda_matrices(F, Sigma, scaling)
        assert numpy.abs(Phi - rPhi).max() <= 1e-10 * numpy.abs(rPhi).max()
        assert numpy.abs(Psi - rPsi).max() <= 1e-10 * numpy.abs(rPsi).max()
        assert abs(cst - rcst) <= 1e-10 * abs(rcst)
    rs = numpy.random.RandomState(3)
    G = 0.3 * rs.randn(F.shape[0], 24)
    e, t = rs.randn(17, F.shape[0]) + mu, rs.randn(23, F.shape[0]) + mu
    B, Phi, Psi, c = iv_scoring.full_plda_parameters(F, G, Sigma, 0.9)
    ep, tp = (e - mu) @ B.T, (t - mu) @ B.T
    got = 0.9 * (0.5 * numpy.einsum("ij,jk,ik->i", ep, Phi, ep)[:, None] + 0.5 * numpy.einsum("ij,jk,ik->i", tp, Phi, tp)[None, :] + c + ep @ Psi @ tp.T)
    want = osc.full_plda_scores(e, t, mu, F, G, Sigma, scaling_factor=0.9)
    assert numpy.abs(got - want).max() <= 1e-9 * numpy.abs(want).max()
