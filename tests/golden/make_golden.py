"""Generate the golden fixtures under tests/golden/ by running the REAL reference modules.

Runs only in the build container (needs /root/reference, never on the GPU box).  The reference
package cannot be imported whole (``import sidekit`` needs h5py / torchaudio / soundfile /
kaldiio, none installed), so -- as SURVEY.md 8(c) describes -- an empty ``sidekit`` package shell
is registered with its three constants, stand-in modules are registered for the missing
third-party packages, and the reference's own source files are then imported unchanged:
``sidekit.nnet.xvector`` (Xtractor, PreHalfResNet34, BasicBlock, SELayer, AttentivePooling,
MeanStdPooling, l2_norm, ArcMarginProduct), ``sidekit.iv_scoring``, ``sidekit.statserver``,
``sidekit.bosaris`` (Ndx, Key, Scores, IdMap, rocch, rocch2eer, pavx).

The torchaudio stand-in's MelSpectrogram / MFCC compute with oracle/frontend.py (the build's own
restatement of torchaudio 0.8.2), so every *network* fixture is cut at the features seam, where no
stand-in arithmetic is involved; wav-level fixtures are labelled `unpinned_frontend`.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [halfresnet34] [tdnn] [scoring] [asnorm] [examples] [config5]
"""
import importlib
import os
import sys
import types

import numpy
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import frontend as ofe  # noqa: E402
from oracle import xvector as oxv  # noqa: E402
from sidekit_amd.nnet.weights import seeded_state_dict, state_dict_spec  # noqa: E402


# ---- stand-ins for packages missing from the container -------------------------------------------
def _install_standins():
    ta = types.ModuleType("torchaudio")
    tr = types.ModuleType("torchaudio.transforms")

    class _Spectrogram(torch.nn.Module):
        def __init__(self, n_fft, win_length, hop_length, window_fn):
            super().__init__()
            self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
            self.register_buffer("window", window_fn(win_length))

    class _MelScale(torch.nn.Module):
        def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
            super().__init__()
            self.register_buffer("fb", ofe.mel_filterbank(n_stft, f_min, f_max, n_mels, sample_rate))

    class MelSpectrogram(torch.nn.Module):
        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0., f_max=None, pad=0,
                     n_mels=128, window_fn=torch.hann_window, power=2., **kw):
            super().__init__()
            self.spectrogram = _Spectrogram(n_fft, win_length, hop_length, window_fn)
            self.mel_scale = _MelScale(n_mels, sample_rate, f_min, f_max, n_fft // 2 + 1)

        def forward(self, x):
            s = self.spectrogram
            spec = ofe.stft_power(x, s.n_fft, s.hop_length, s.win_length, s.window)
            return torch.matmul(spec.transpose(1, 2), self.mel_scale.fb).transpose(1, 2)

    class MFCC(torch.nn.Module):
        def __init__(self, sample_rate=16000, n_mfcc=40, dct_type=2, norm='ortho', log_mels=False, melkwargs=None):
            super().__init__()
            self.MelSpectrogram = MelSpectrogram(sample_rate=sample_rate, **melkwargs)
            self.register_buffer("dct_mat", ofe.dct_matrix(n_mfcc, melkwargs["n_mels"]))

        def forward(self, x):
            mel = torch.log(self.MelSpectrogram(x) + 1e-6)
            return torch.matmul(mel.transpose(1, 2), self.dct_mat).transpose(1, 2)

    class _Mask(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tr.MelSpectrogram, tr.MFCC, tr.TimeMasking, tr.FrequencyMasking, tr.Resample = MelSpectrogram, MFCC, _Mask, _Mask, _Mask
    ta.transforms = tr
    ta.functional = types.ModuleType("torchaudio.functional")
    ta.sox_effects = types.ModuleType("torchaudio.sox_effects")
    for name, mod in (("torchaudio", ta), ("torchaudio.transforms", tr), ("torchaudio.functional", ta.functional),
                      ("torchaudio.sox_effects", ta.sox_effects)):
        sys.modules[name] = mod
    for name in ("h5py", "soundfile", "kaldiio", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def import_reference():
    _install_standins()
    pkg = types.ModuleType("sidekit")
    pkg.__path__ = [os.path.join(REF, "sidekit")]
    pkg.PARALLEL_MODULE = 'multiprocessing'
    pkg.PARAM_TYPE = numpy.float32
    pkg.STAT_TYPE = numpy.float64
    sys.modules["sidekit"] = pkg
    nn = types.ModuleType("sidekit.nnet")
    nn.__path__ = [os.path.join(REF, "sidekit", "nnet")]
    sys.modules["sidekit.nnet"] = nn
    mods = {}
    for m in ("sidekit.bosaris", "sidekit.statserver", "sidekit.iv_scoring", "sidekit.bosaris.detplot", "sidekit.nnet.xvector",
              "sidekit.nnet.pooling"):
        mods[m] = importlib.import_module(m)
    pkg.StatServer = mods["sidekit.statserver"].StatServer
    return mods


def digest(t):
    """Size-independent summary of a big activation: moments + a fixed strided sample."""
    a = t.detach().double().flatten()
    idx = torch.linspace(0, a.numel() - 1, 257).long()
    return numpy.concatenate([[a.mean().item(), a.abs().mean().item(), a.pow(2).mean().sqrt().item(), a.max().item(), a.min().item()],
                              a[idx].numpy()])


def halfresnet_fixtures(mods, out):
    xv, pooling = mods["sidekit.nnet.xvector"], mods["sidekit.nnet.pooling"]
    n_spk, seed = 16, 1234
    ref = xv.Xtractor(n_spk, model_archi="halfresnet34", loss="aam")
    ref.stat_pooling = pooling.AttentivePooling(256, 10, global_context=True)  # SURVEY F1'
    assert list(ref.state_dict().keys()) == list(state_dict_spec("halfresnet34", n_spk).keys()), "key order differs"
    sd = seeded_state_dict("halfresnet34", n_spk, seed=seed)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    fx = {"n_spk": n_spk, "seed": seed}
    with torch.no_grad():
        for tag, B, T, fseed in (("small", 2, 51, 11), ("len4s", 1, 401, 12), ("odd", 1, 77, 13)):
            g = torch.Generator().manual_seed(fseed)
            feats = torch.randn(B, 80, T, generator=g)
            taps = {}
            hooks = [ref.sequence_network.bn1.register_forward_hook(lambda m, i, o: taps.__setitem__("stem", torch.relu(o)))]
            for li in range(1, 5):
                hooks.append(getattr(ref.sequence_network, f"layer{li}").register_forward_hook(
                    lambda m, i, o, li=li: taps.__setitem__(f"layer{li}", o)))
            x = ref.sequence_network(feats)
            pooled = ref.stat_pooling(x)
            pre = ref.before_speaker_embedding(pooled)
            emb = xv.l2_norm(pre)
            logits = ref.after_speaker_embedding(emb, target=None)
            emb2 = torch.nn.functional.normalize(emb, dim=1)
            for h in hooks:
                h.remove()
            fx[f"{tag}_feat_seed"] = fseed
            fx[f"{tag}_shape"] = numpy.array([B, 80, T])
            for k in ("stem", "layer1", "layer2", "layer3"):
                fx[f"{tag}_{k}_digest"] = digest(taps[k])
            fx[f"{tag}_layer4"] = taps["layer4"].numpy().astype(numpy.float32)   # (B,256,T',10)
            fx[f"{tag}_pooled"] = pooled.numpy()
            fx[f"{tag}_pre_norm"] = pre.numpy()
            fx[f"{tag}_emb"] = emb2.numpy()
            fx[f"{tag}_logits"] = logits.numpy()
            # the oracle must agree with the reference it restates
            o_logits, o_emb = oxv.halfresnet34_from_feats(feats, sd)
            assert torch.allclose(o_emb, emb2, atol=2e-6), (tag, (o_emb - emb2).abs().max())
            assert torch.allclose(o_logits, logits, atol=2e-4)
        # wav-level (front-end = build's restatement on both sides -> unpinned_frontend)
        import scipy.io.wavfile
        sr, wav = scipy.io.wavfile.read(os.path.join(REF, "egs/examples_decode/1272-128104-0000.wav"))
        assert sr == 16000
        pcm = wav[16000:16000 + 32000].astype(numpy.int16)   # a 2 s excerpt, kept as int16
        x = torch.from_numpy(pcm.astype(numpy.float32) / 32768.0)
        _, emb = ref(x, is_eval=True)
        fx["wav_pcm16"] = pcm
        fx["wav_emb_unpinned_frontend"] = emb.numpy()
        fx["wav_feats_unpinned_frontend"] = ref.preprocessor(x, is_eval=True).numpy()
    numpy.savez_compressed(os.path.join(out, "halfresnet34.npz"), **fx)
    print("halfresnet34.npz", {k: getattr(v, "shape", v) for k, v in fx.items() if not k.endswith("digest")})


def tdnn_fixtures(mods, out):
    xv = mods["sidekit.nnet.xvector"]
    n_spk, seed = 16, 4321
    fx = {"n_spk": n_spk, "seed": seed}
    for loss in ("aam", "cce"):
        ref = xv.Xtractor(n_spk, model_archi="xvector", loss=loss)
        assert list(ref.state_dict().keys()) == list(state_dict_spec("xvector", n_spk, loss=loss).keys()), "key order differs"
        sd = seeded_state_dict("xvector", n_spk, loss=loss, seed=seed)
        ref.load_state_dict(sd, strict=True)
        ref.eval()
        with torch.no_grad():
            for tag, B, T, fseed in (("t63", 2, 63, 21), ("t126", 1, 126, 22)):
                g = torch.Generator().manual_seed(fseed)
                feats = torch.randn(B, 80, T, generator=g)
                # sub-modules called in forward order by hand (Xtractor.forward itself raises TypeError)
                x = ref.sequence_network(feats)
                pooled = ref.stat_pooling(x)
                pre = ref.before_speaker_embedding(pooled)
                emb = xv.l2_norm(pre)
                fx[f"{loss}_{tag}_feat_seed"] = fseed
                fx[f"{loss}_{tag}_shape"] = numpy.array([B, 80, T])
                if loss == "aam":
                    fx[f"{tag}_conv5_digest"] = digest(x)
                    fx[f"{tag}_pooled"] = pooled.numpy()
                    fx[f"{tag}_pre_norm"] = pre.numpy()
                    logits = ref.after_speaker_embedding(emb, target=None)
                    fx[f"{loss}_{tag}_logits"] = logits.numpy()
                    fx[f"{loss}_{tag}_emb"] = torch.nn.functional.normalize(emb, dim=1).numpy()
                    o_logits, o_emb = oxv.tdnn_from_feats(feats, sd, "aam")
                    assert torch.allclose(o_logits, logits, atol=3e-4)
                else:
                    fx[f"{loss}_{tag}_emb"] = emb.numpy()
                    o_emb = oxv.tdnn_from_feats(feats, sd, "cce")
                assert torch.allclose(o_emb, torch.from_numpy(fx[f"{loss}_{tag}_emb"]), atol=2e-6)
    numpy.savez_compressed(os.path.join(out, "tdnn.npz"), **fx)
    print("tdnn.npz", sorted(fx))


def scoring_fixtures(mods, out):
    from oracle import scoring as osc
    ivs, sts_mod, bos = mods["sidekit.iv_scoring"], mods["sidekit.statserver"], mods["sidekit.bosaris"]
    det = mods["sidekit.bosaris.detplot"]
    rs = numpy.random.RandomState(5)
    D, Ne, Nt, rank = 256, 32, 40, 24
    fx = {}

    def make_sts(models, segs, X):
        s = sts_mod.StatServer()
        s.modelset = numpy.array(models, dtype="|O")
        s.segset = numpy.array(segs, dtype="|O")
        s.start = numpy.empty(len(segs), dtype="|O")
        s.stop = numpy.empty(len(segs), dtype="|O")
        s.stat0 = numpy.ones((len(segs), 1))
        s.stat1 = numpy.array(X, dtype=numpy.float64)
        return s

    spk = rs.randn(8, D)
    E = spk[rs.randint(0, 8, Ne)] + 0.8 * rs.randn(Ne, D)
    T = spk[rs.randint(0, 8, Nt)] + 0.8 * rs.randn(Nt, D)
    enr_ids = [f"m{i:03d}" for i in rs.permutation(Ne)]          # unsorted on purpose
    tst_ids = [f"s{i:03d}" for i in rs.permutation(Nt)]
    enroll = make_sts(enr_ids, enr_ids, E)
    test = make_sts(tst_ids, tst_ids, T)
    # trial list: a random 60 % of the full matrix, plus two models / one segment that do not exist
    mm, ss = numpy.meshgrid(numpy.arange(Ne), numpy.arange(Nt), indexing="ij")
    keep = rs.rand(Ne, Nt) < 0.6
    models = numpy.array([enr_ids[i] for i in mm[keep]] + ["ghost_a", "ghost_b"], dtype="|O")
    segs = numpy.array([tst_ids[j] for j in ss[keep]] + [tst_ids[0], "ghost_seg"], dtype="|O")
    ndx = bos.Ndx(models=models, testsegs=segs)
    fx.update(E=E, T=T, enr_ids=numpy.array(enr_ids), tst_ids=numpy.array(tst_ids), trial_models=models.astype(str),
              trial_segs=segs.astype(str), ndx_modelset=ndx.modelset.astype(str), ndx_segset=ndx.segset.astype(str),
              ndx_trialmask=ndx.trialmask)
    sc = ivs.cosine_scoring(enroll, test, ndx, wccn=None, check_missing=True, device=torch.device("cpu"))
    fx.update(cos_modelset=sc.modelset.astype(str), cos_segset=sc.segset.astype(str), cos_scoremask=sc.scoremask,
              cos_scoremat=sc.scoremat)
    W = rs.randn(D, D) / numpy.sqrt(D)
    scw = ivs.cosine_scoring(enroll, test, ndx, wccn=W, check_missing=True, device=torch.device("cpu"))
    fx.update(wccn=W, cos_wccn_scoremat=scw.scoremat)
    # PLDA parameters: random but well conditioned
    mu = 0.1 * rs.randn(D)
    F = rs.randn(D, rank) / numpy.sqrt(D)
    G = rs.randn(D, 12) / numpy.sqrt(D)
    A = rs.randn(D, D) / numpy.sqrt(D)
    Sigma = A.dot(A.T) + 0.5 * numpy.eye(D)
    fx.update(mu=mu, F=F, G=G, Sigma=Sigma)
    for tag, kw in (("plda", {}), ("plda_scaled", {"scaling_factor": 0.7}), ("plda_open", {"p_known": 0.3})):
        p = ivs.fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, **kw)
        fx[f"{tag}_scoremat"] = p.scoremat
        fx[f"{tag}_scoremask"] = p.scoremask
    pf = ivs.PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, full_model=True)
    fx["plda_full_scoremat"] = pf.scoremat
    pf2 = ivs.full_PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, p_known=0.2, scaling_factor=0.9)
    fx["plda_full_open_scoremat"] = pf2.scoremat
    # duplicate enrolment models -> averaged with a warning (iv_scoring.py:409-411)
    dup_ids = [enr_ids[i // 2] for i in range(Ne)]
    enroll_dup = make_sts(dup_ids, [f"e{i:03d}" for i in range(Ne)], E)
    pd_ = ivs.fast_PLDA_scoring(enroll_dup, test, ndx, mu, F, Sigma)
    fx.update(dup_ids=numpy.array(dup_ids), plda_dup_modelset=pd_.modelset.astype(str), plda_dup_scoremat=pd_.scoremat,
              plda_dup_scoremask=pd_.scoremask)
    # oracle agreement on the aligned arrays
    em = {m: i for i, m in enumerate(enr_ids)}
    tm = {s: i for i, s in enumerate(tst_ids)}
    Ea = E[[em[m] for m in sc.modelset]]
    Ta = T[[tm[s] for s in sc.segset]]
    assert numpy.allclose(osc.cosine_scores(Ea, Ta), sc.scoremat, atol=1e-6)
    assert numpy.allclose(osc.fast_plda_scores(Ea, Ta, mu, F, Sigma), fx["plda_scoremat"], rtol=1e-9, atol=1e-9)
    assert numpy.allclose(osc.fast_plda_scores(Ea, Ta, mu, F, Sigma, p_known=0.3), fx["plda_open_scoremat"], rtol=1e-9, atol=1e-9)
    assert numpy.allclose(osc.full_plda_scores(Ea, Ta, mu, F, G, Sigma), fx["plda_full_scoremat"], rtol=1e-9, atol=1e-9)
    # StatServer algebra before/after
    s2 = make_sts(enr_ids, enr_ids, E)
    s2.norm_stat1()
    fx["norm_stat1"] = s2.stat1
    s3 = make_sts(enr_ids, enr_ids, E)
    s3.center_stat1(mu)
    fx["center_stat1"] = s3.stat1
    s4 = make_sts(dup_ids, [f"e{i:03d}" for i in range(Ne)], E).mean_stat_per_model()
    fx.update(mean_per_model_modelset=s4.modelset.astype(str), mean_per_model_stat1=s4.stat1)
    s5 = make_sts(enr_ids, enr_ids, E)
    s5.whiten_stat1(mu, Sigma)
    fx["whiten_stat1"] = s5.stat1
    # Key / Scores.get_tar_non / EER
    lab = numpy.where(rs.rand(len(models)) < 0.3, "target", "nontarget").astype("|O")
    key = bos.Key(models=models, testsegs=segs, trials=lab)
    fx.update(trial_labels=lab.astype(str), key_modelset=key.modelset.astype(str), key_segset=key.segset.astype(str), key_tar=key.tar,
              key_non=key.non)
    # Scores.get_tar_non's shape-mismatch branch (scores.py:166) raises under numpy >= 1.25 (elementwise == of
    # arrays of different length), so the mismatching key goes through align_with_ndx directly (what that branch does)
    aligned = sc.align_with_ndx(key)
    tar = aligned.scoremat[key.tar & aligned.scoremask]
    non = aligned.scoremat[key.non & aligned.scoremask]
    fx.update(cos_tar=tar, cos_non=non, aligned_scoremat=aligned.scoremat, aligned_scoremask=aligned.scoremask)
    # and the matching-shape branch with a key cut down to the scored sets
    ok = numpy.array([m in set(sc.modelset) and s_ in set(sc.segset) for m, s_ in zip(models, segs)])
    key2 = bos.Key(models=models[ok], testsegs=segs[ok], trials=lab[ok])
    tar2, non2 = sc.get_tar_non(key2)
    fx.update(cos_tar2=tar2, cos_non2=non2)
    # ROCCH on synthetic scores with ties
    tar_s = numpy.round(rs.randn(300) + 1.5, 1)
    non_s = numpy.round(rs.randn(700), 1)
    pmiss, pfa = det.rocch(tar_s, non_s)
    fx.update(rocch_tar=tar_s, rocch_non=non_s, rocch_pmiss=pmiss, rocch_pfa=pfa, rocch_eer=det.rocch2eer(pmiss, pfa))
    y = rs.rand(200)
    gh, wd, hg = det.pavx(y)
    fx.update(pav_y=y, pav_ghat=gh, pav_width=wd, pav_height=hg)
    o_pm, o_pf = osc.rocch(tar_s, non_s)
    assert numpy.array_equal(o_pm, pmiss) and numpy.array_equal(o_pf, pfa)
    assert osc.rocch2eer(o_pm, o_pf) == fx["rocch_eer"]
    og, ow, oh = osc.pavx(y)
    assert numpy.array_equal(og, gh) and numpy.array_equal(ow, wd) and numpy.array_equal(oh, hg)
    pm2, pf2_ = det.rocch(tar, non)
    fx["cos_eer"] = det.rocch2eer(pm2, pf2_)
    # beyond SURVEY 8's rows: Mahalanobis and two-covariance scoring (iv_scoring.py:116-213), same trial list (these two reorder
    # the caller's StatServers in place, hence the fresh copies); new random draws come last so the older keys regenerate unchanged
    Mm = rs.randn(D, D) / numpy.sqrt(D)
    Mm = Mm.dot(Mm.T) + 0.3 * numpy.eye(D)
    Ww = rs.randn(D, D) / numpy.sqrt(D)
    Ww = Ww.dot(Ww.T) + 0.5 * numpy.eye(D)
    Bb = rs.randn(D, D) / numpy.sqrt(D)
    Bb = Bb.dot(Bb.T) + 0.2 * numpy.eye(D)
    mh = ivs.mahalanobis_scoring(make_sts(enr_ids, enr_ids, E), make_sts(tst_ids, tst_ids, T), ndx, Mm)
    tc = ivs.two_covariance_scoring(make_sts(enr_ids, enr_ids, E), make_sts(tst_ids, tst_ids, T), ndx, Ww, Bb)
    assert list(mh.modelset) == list(sc.modelset) and list(tc.segset) == list(sc.segset)
    fx.update(maha_M=Mm, maha_scoremat=mh.scoremat, twocov_W=Ww, twocov_B=Bb, twocov_scoremat=tc.scoremat)
    numpy.savez_compressed(os.path.join(out, "scoring.npz"), **fx)
    print("scoring.npz", sorted(fx))


def asnorm_fixtures(mods, out):
    from oracle import scoring as osc
    sn = importlib.import_module("sidekit.score_normalization")
    g = torch.Generator().manual_seed(31)
    spk = torch.randn(12, 256, generator=g)
    enrol = torch.nn.functional.normalize(spk[torch.randint(0, 12, (64,), generator=g)] + 0.7 * torch.randn(64, 256, generator=g), dim=1)
    cohort = 3.0 * torch.randn(300, 256, generator=g)          # un-normalised on purpose
    s = sn.asnorm(enrol, cohort, None)
    assert numpy.allclose(osc.asnorm(enrol, cohort), s, atol=1e-6)
    numpy.savez_compressed(os.path.join(out, "asnorm.npz"), seed=31, enrol=enrol.numpy(), cohort=cohort.numpy(), snorm=s.astype(numpy.float32))
    print("asnorm.npz", s.shape, s.dtype)


def examples_fixtures(mods, out):
    """Fixture set (ii) of SURVEY 8(c) / BASELINE config 1: the reference HalfResNet34 (F1' pooling shape, seeded checkpoint) run on
    the three ``egs/examples_decode`` wavs exactly as ``extract_xvectors.py:143-147`` feeds them (whole file, batch 1) and on
    their first 64000 samples.  The wavs travel as PCM16 arrays (the GPU box has no reference tree).  The reference front-end
    is the torchaudio stand-in (= oracle/frontend.py), so ``emb_*`` is at once the features-seam target (oracle features ->
    reference network: pinned) and the wav-level target of the build's own front-end (labelled ``unpinned_frontend``)."""
    import scipy.io.wavfile
    xv, pooling = mods["sidekit.nnet.xvector"], mods["sidekit.nnet.pooling"]
    n_spk, seed = 16, 1234
    ref = xv.Xtractor(n_spk, model_archi="halfresnet34", loss="aam")
    ref.stat_pooling = pooling.AttentivePooling(256, 10, global_context=True)  # SURVEY F1'
    sd = seeded_state_dict("halfresnet34", n_spk, seed=seed)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    fx = {"n_spk": n_spk, "seed": seed, "sample_rate": 16000}
    names = []
    with open(os.path.join(REF, "egs/examples_decode/wav_example.scp")) as f:
        for line in f:
            key, wav = line.split()
            names.append(key)
            sr, pcm = scipy.io.wavfile.read(os.path.join(REF, "egs/examples_decode", wav))
            assert sr == 16000 and pcm.dtype == numpy.int16 and pcm.ndim == 1
            fx[f"pcm16_{key}"] = pcm
            x = torch.from_numpy(pcm.astype(numpy.float32) / 32768.0)
            with torch.no_grad():
                for tag, sig in (("full", x), ("first4s", x[:64000])):
                    _, emb = ref(sig, is_eval=True)
                    fx[f"emb_{tag}_{key}_unpinned_frontend"] = emb.numpy()
                    feats = ofe.melspec_frontend(sig.unsqueeze(0))
                    o_logits, o_emb = oxv.halfresnet34_from_feats(feats, sd)
                    assert torch.allclose(o_emb, emb, atol=2e-6), (key, tag, (o_emb - emb).abs().max())
    fx["keys"] = numpy.array(names)
    numpy.savez_compressed(os.path.join(out, "examples_decode.npz"), **fx)
    print("examples_decode.npz", {k: getattr(v, "shape", v) for k, v in fx.items()})


def config5_fixtures(mods, out):
    """BASELINE config 5 at its stated size, pinned by the reference itself: PLDA ``(mu, F, Sigma)`` trained by the
    reference's ``FactorAnalyser.plda`` (factor_analyser.py:830-932, rank 128, 10 EM iterations) on the disjoint synthetic training
    set, then the reference's ``cosine_scoring`` / ``fast_PLDA_scoring`` (iv_scoring.py:63-113,370-477) on the 1000 x 1000 full trial
    mask, ``Scores.get_tar_non`` and ``rocch`` / ``rocch2eer`` (detplot.py:354-436).  Stored: the PLDA parameters, digests of the
    regenerated inputs, a strided sample + row / column sums + moments of both score matrices, the ROCCH vertices and the two EERs."""
    from oracle import scoring as osc
    sys.path.insert(0, HERE)
    import config5_inputs as c5
    ivs, sts_mod, bos = mods["sidekit.iv_scoring"], mods["sidekit.statserver"], mods["sidekit.bosaris"]
    det = mods["sidekit.bosaris.detplot"]
    fa = importlib.import_module("sidekit.factor_analyser")

    def make_sts(models, segs, X):
        s = sts_mod.StatServer()
        s.modelset, s.segset = numpy.array(models, dtype="|O"), numpy.array(segs, dtype="|O")
        s.start, s.stop = numpy.empty(len(segs), dtype="|O"), numpy.empty(len(segs), dtype="|O")
        s.stat0 = numpy.ones((len(segs), 1))
        s.stat1 = numpy.array(X, dtype=numpy.float64)
        return s

    X, lab = c5.plda_training_set()
    train = make_sts([f"spk{l:04d}" for l in lab], c5.ids("tr", X.shape[0]), X)
    plda = fa.FactorAnalyser()
    plda.plda(train, rank_f=c5.PLDA_RANK, nb_iter=10, save_final=False)
    mu, F, Sigma = plda.mean, plda.F, plda.Sigma
    E, T, spk_e, spk_t = c5.trial_set()
    enr_ids, tst_ids = c5.ids("enr", c5.NE), c5.ids("tst", c5.NT)
    enroll, test = make_sts(enr_ids, enr_ids, E), make_sts(tst_ids, tst_ids, T)
    mm, ss = numpy.meshgrid(numpy.arange(c5.NE), numpy.arange(c5.NT), indexing="ij")
    models, segs = enr_ids[mm.ravel()], tst_ids[ss.ravel()]
    tar_mask = spk_e[:, None] == spk_t[None, :]
    ndx = bos.Ndx(models=models, testsegs=segs)
    key = bos.Key(models=models, testsegs=segs, trials=numpy.where(tar_mask.ravel(), "target", "nontarget").astype("|O"))
    assert ndx.trialmask.all() and numpy.array_equal(key.tar, tar_mask)
    fx = {"mu": mu, "F": F, "Sigma": Sigma, "E_digest": c5.digest(E), "T_digest": c5.digest(T), "X_digest": c5.digest(X),
          "n_target": int(tar_mask.sum()), "sample_rows": 7, "sample_cols": 11}
    for tag, sc in (("cos", ivs.cosine_scoring(enroll, test, ndx, wccn=None, check_missing=True, device=torch.device("cpu"))),
                    ("plda", ivs.fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma))):
        assert list(sc.modelset) == list(enr_ids) and list(sc.segset) == list(tst_ids) and sc.scoremask.all()
        m = sc.scoremat
        tar, non = sc.get_tar_non(key)
        pmiss, pfa = det.rocch(tar, non)
        m64 = m.astype(numpy.float64)
        fx.update({f"{tag}_dtype": str(m.dtype), f"{tag}_sample": m[::7, ::11].copy(), f"{tag}_row_sums": m64.sum(axis=1),
                   f"{tag}_col_sums": m64.sum(axis=0), f"{tag}_moments": numpy.array([m64.mean(), m64.std(), m64.min(), m64.max()]),
                   f"{tag}_pmiss": pmiss, f"{tag}_pfa": pfa, f"{tag}_eer": det.rocch2eer(pmiss, pfa)})
        print(tag, m.dtype, "EER", fx[f"{tag}_eer"])
        # the oracle must agree with the reference it restates, at this size too
        o = osc.cosine_scores(E, T) if tag == "cos" else osc.fast_plda_scores(E, T, mu, F, Sigma)
        assert numpy.allclose(o, m, rtol=1e-9, atol=2e-6 if tag == "cos" else 1e-9), numpy.abs(o - m).max()
        assert abs(osc.eer(o[tar_mask], o[~tar_mask]) - fx[f"{tag}_eer"]) < 1e-9
    numpy.savez_compressed(os.path.join(out, "config5.npz"), **fx)
    print("config5.npz", {k: getattr(v, "shape", v) for k, v in fx.items()})


def main():
    mods = import_reference()
    torch.set_num_threads(8)
    only = sys.argv[1:] or ["halfresnet34", "tdnn", "scoring", "asnorm", "examples", "config5"]
    if "halfresnet34" in only:
        halfresnet_fixtures(mods, HERE)
    if "tdnn" in only:
        tdnn_fixtures(mods, HERE)
    if "scoring" in only:
        scoring_fixtures(mods, HERE)
    if "asnorm" in only:
        asnorm_fixtures(mods, HERE)
    if "examples" in only:
        examples_fixtures(mods, HERE)
    if "config5" in only:
        config5_fixtures(mods, HERE)


if __name__ == "__main__":
    main()
