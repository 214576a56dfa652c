"""BASELINE config 1 on its stated inputs (the three egs/examples_decode wavs, batch 1, wav.scp -> ark/scp through
sidekit_amd.bin.extract_xvectors), very short clips as a model's first call, the `install_as_sidekit()` boundary with
reference-style caller code, and per-stage bf16 drift against the fp32 oracle."""
import os
import sys

import numpy
import pytest
import scipy.io.wavfile
import torch

from oracle import frontend as ofe
from oracle import xvector as oxv
from sidekit_amd.bin import extract_xvectors
from sidekit_amd.kaldi_io import read_scp
from sidekit_amd.nnet import Xtractor
from sidekit_amd.nnet.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def ex(golden_dir):
    return numpy.load(os.path.join(golden_dir, "examples_decode.npz"))


def _checkpoint(ex, path):
    n_spk = int(ex["n_spk"])
    sd = seeded_state_dict("halfresnet34", n_spk, seed=int(ex["seed"]))
    torch.save({"speaker_number": n_spk, "model_archi": {"model_type": "halfresnet34", "loss": {"type": "aam"}}, "model_state_dict": sd}, path)
    return sd


@pytest.mark.parametrize("tag", ["full", "first4s"])
def test_config1_wav_scp_batch1_through_the_cli(gpu, ex, tmp_path, tag):
    """`extract_xvectors.py --model --wav-scp --out-scp --device` (README.md:39-43) at batch 1: 93680 / 199760 / 158400-sample
    files and their first 64000 samples; targets = the imported reference on the same PCM (front-end: unpinned_frontend)."""
    _checkpoint(ex, tmp_path / "model.pt")
    keys = [str(k) for k in ex["keys"]]
    with open(tmp_path / "wav_example.scp", "w") as f:                 # same two-column form as egs/examples_decode/wav_example.scp
        for k in keys:
            pcm = ex[f"pcm16_{k}"]
            pcm = pcm if tag == "full" else pcm[:64000]
            scipy.io.wavfile.write(tmp_path / f"{k}.wav", int(ex["sample_rate"]), pcm)
            f.write(f"{k} {tmp_path / (k + '.wav')}\n")
    extract_xvectors.cli(["--model", str(tmp_path / "model.pt"), "--wav-scp", str(tmp_path / "wav_example.scp"), "--out-scp",
                          str(tmp_path / "xv.scp"), "--device", "cuda", "--batch-size", "1"])
    got = dict(read_scp(str(tmp_path / "xv.scp")))
    assert list(got) == keys
    for k in keys:
        want = ex[f"emb_{tag}_{k}_unpinned_frontend"]
        assert got[k].shape == (1, 256) and got[k].dtype == numpy.float32
        assert rel(got[k], want) < TOL, (k, tag)
        assert abs(float(numpy.linalg.norm(got[k])) - 1.0) < 1e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pcm16_entry_is_bit_identical_to_the_float_entry(gpu, ex, dtype):
    """`xt_forward_pcm16` (int16 rows widened inside the STFT kernel's load) against `xt_forward` on `pcm.float() / 32768`: the
    three egs/examples_decode wavs alone (batch 1, as extract_xvectors.py:146 calls the model) and as one ragged batch, plus the
    TDNN front-end; x-vectors and logits must be the SAME BITS, and match the reference embeddings."""
    n_spk = int(ex["n_spk"])
    model = Xtractor(n_spk, model_archi="halfresnet34", loss="aam", seed=int(ex["seed"])).to(gpu).eval()
    model.compute_dtype = dtype
    keys = [str(k) for k in ex["keys"]]
    pcms = [torch.from_numpy(ex[f"pcm16_{k}"].astype(numpy.int16)) for k in keys]
    for k, pcm in zip(keys, pcms):
        lg_i, e_i = model(pcm.to(gpu), is_eval=True)
        lg_f, e_f = model((pcm.float() / 32768.0).to(gpu), is_eval=True)
        assert torch.equal(e_i, e_f) and torch.equal(lg_i, lg_f), k
        if dtype == "fp32":
            assert rel(e_i, ex[f"emb_full_{k}_unpinned_frontend"]) < TOL
    lens = [p.shape[0] for p in pcms]
    batch = torch.zeros(3, max(lens), dtype=torch.int16)
    for r, p in enumerate(pcms):
        batch[r, :lens[r]] = p
    batch[0, lens[0]:] = 12345          # padding is never read: rows are cut at their own length
    _, e_i = model(batch.to(gpu), is_eval=True, lengths=lens)
    _, e_f = model((batch.float() / 32768.0).to(gpu), is_eval=True, lengths=lens)
    assert torch.equal(e_i, e_f)
    if dtype == "fp32":
        for r, k in enumerate(keys):
            assert rel(e_i[r], ex[f"emb_full_{k}_unpinned_frontend"]) < TOL
        tdnn = Xtractor(n_spk, model_archi="xvector", loss="aam", seed=4321).to(gpu).eval()
        _, t_i = tdnn(batch.to(gpu), is_eval=True, lengths=lens)
        _, t_f = tdnn((batch.float() / 32768.0).to(gpu), is_eval=True, lengths=lens)
        assert torch.equal(t_i, t_f)


def test_config1_features_seam_is_pinned(gpu, ex):
    """The same six targets entered after the front-end: oracle features (CPU) -> forward_features on the GPU.  No front-end
    of the build is involved on either side, so this is the pinned half of fixture set (ii)."""
    model = Xtractor(int(ex["n_spk"]), model_archi="halfresnet34", loss="aam", seed=0).to(gpu).eval()
    model.load_state_dict(seeded_state_dict("halfresnet34", int(ex["n_spk"]), seed=int(ex["seed"])), strict=True)
    model.compute_dtype = "fp32"
    for k in [str(k) for k in ex["keys"]]:
        x = torch.from_numpy(ex[f"pcm16_{k}"].astype(numpy.float32) / 32768.0)
        for tag, sig in (("full", x), ("first4s", x[:64000])):
            feats = ofe.melspec_frontend(sig.unsqueeze(0))
            _, emb = model.forward_features(feats.cuda())
            assert rel(emb, ex[f"emb_{tag}_{k}_unpinned_frontend"]) < TOL, (k, tag)
    # bf16 trunk on the real speech: same direction within bf16 noise
    model.compute_dtype = "bf16"
    k = str(ex["keys"][0])
    x = torch.from_numpy(ex[f"pcm16_{k}"].astype(numpy.float32) / 32768.0)
    _, e16 = model(x.cuda(), is_eval=True)
    cos = float(torch.nn.functional.cosine_similarity(e16.cpu(), torch.from_numpy(ex[f"emb_full_{k}_unpinned_frontend"])))
    assert cos > 0.999, cos


def test_short_clips_as_first_call(gpu):
    """A batch of 0.05-0.15 s clips as the very first call of a model (the first length-sorted batch of extract_xvectors):
    6..16 feature frames, one row tile per layer -- the SE statistics workspace must be sized from the per-layer maximum, not
    per frame (a 2-KB layer-4 tile per utterance used to overrun it).  Batched == each clip alone == the oracle."""
    sd = seeded_state_dict("halfresnet34", 16, seed=91)
    lens = [800, 1023, 1300, 1599, 1600, 1777, 2047, 2400, 960, 2399, 1120, 2240]
    torch.manual_seed(3)
    wav = 0.1 * torch.randn(len(lens), max(lens))
    with torch.no_grad():
        _, ref = oxv.forward_ragged([wav[i, :n] for i, n in enumerate(lens)], sd)
    for dtype in ("fp32", "bf16"):
        model = Xtractor(16, model_archi="halfresnet34", loss="aam", seed=0).to(gpu).eval()     # fresh handle: nothing reserved yet
        model.load_state_dict(sd, strict=True)
        model.compute_dtype = dtype
        _, emb = model(wav.cuda(), is_eval=True, lengths=lens)
        finite = torch.isfinite(ref).all(dim=1)
        assert bool(finite.sum() >= 8)                                   # T' = 1 clips give the reference's NaN x-vector (0/0 unbiased std)
        for i, n in enumerate(lens):
            if not bool(finite[i]):
                assert bool(torch.isnan(emb[i]).all()), (dtype, i, n)
                continue
            if dtype == "fp32":
                assert rel(emb[i], ref[i]) < TOL, (i, n)
            else:
                assert float(torch.nn.functional.cosine_similarity(emb[i].cpu(), ref[i], dim=0)) > 0.995, (i, n)
            _, one = model(wav[i, :n].cuda(), is_eval=True)
            assert torch.equal(one[0], emb[i]), (dtype, i, n)            # bit for bit: no neighbour's statistics leaked in
        # a long batch afterwards grows the workspace by its own B x L product only, and results stay right
        torch.manual_seed(4)
        long_wav = 0.1 * torch.randn(2, 48000)
        with torch.no_grad():
            _, lref = oxv.halfresnet34_forward(long_wav, sd)
        _, lemb = model(long_wav.cuda(), is_eval=True)
        assert (rel(lemb, lref) < TOL) if dtype == "fp32" else (float(torch.nn.functional.cosine_similarity(lemb.cpu(), lref).min()) > 0.995)


def test_workspace_grows_by_product_not_by_maxima(gpu):
    """a B=64 batch of 1 s clips followed by one 60 s utterance must not reserve 64 x 60 s."""
    model = Xtractor(16, model_archi="halfresnet34", loss="aam", seed=1).to(gpu).eval()
    model.compute_dtype = "bf16"
    torch.cuda.synchronize()
    model(0.1 * torch.randn(64, 16000, device=gpu), is_eval=True)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    _, e = model(0.1 * torch.randn(1, 960000, device=gpu), is_eval=True)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert bool(torch.isfinite(e).all())
    assert free_before - free_after < 6 * 2 ** 30, (free_before - free_after) / 2 ** 30   # the 64 x 60 s product would need > 40 GB
    assert sorted(model._reserved["bf16"]) == [(1, 960000), (64, 16000)]


def test_install_as_sidekit_runs_reference_style_callers(gpu, ex, tmp_path):
    """INTEGRATION.md's mechanism: after `install_as_sidekit()` the reference's own import lines and call shapes
    (sidekit/bin/extract_xvectors.py:74-89,146; sidekit/nnet/xvector.py:68,240-262) run on the MI355X path unchanged."""
    import sidekit_amd
    saved = {k: v for k, v in sys.modules.items() if k == "sidekit" or k.startswith("sidekit.")}
    try:
        pkg = sidekit_amd.install_as_sidekit()
        assert sys.modules["sidekit"] is pkg
        # ---- verbatim reference-style caller code ------------------------------------------------
        import sidekit
        from sidekit.nnet.xvector import Xtractor as RefXtractor
        from sidekit.iv_scoring import cosine_scoring, fast_PLDA_scoring
        from sidekit.bosaris.detplot import rocch, rocch2eer
        from sidekit.bosaris import IdMap, Key, Ndx, Scores
        from sidekit.statserver import StatServer
        from sidekit.score_normalization import asnorm
        from sidekit.sidekit_io import read_plda_hdf5, write_plda_hdf5
        from sidekit.nnet import extract_embeddings
        assert RefXtractor is Xtractor and sidekit.StatServer is StatServer and sidekit.Ndx is Ndx and sidekit.asnorm is asnorm
        assert sidekit.STAT_TYPE is numpy.float64 and sidekit.PARAM_TYPE is numpy.float32
        # load_model of extract_xvectors.py:74-89
        sd = _checkpoint(ex, tmp_path / "model.pt")
        model_config = torch.load(tmp_path / "model.pt", map_location="cpu", weights_only=False)
        model_opts = model_config["model_archi"]
        if "embedding_size" not in model_opts:
            model_opts["embedding_size"] = 256
        xtractor = RefXtractor(model_config["speaker_number"], model_archi=model_opts["model_type"], loss=model_opts["loss"]["type"],
                               embedding_size=model_opts["embedding_size"])
        xtractor.load_state_dict(model_config["model_state_dict"], strict=True)
        xtractor = xtractor.to(torch.device("cuda"))
        xtractor.eval()
        # the extraction loop of extract_xvectors.py:143-147 and the mean / cosine of compute_spk_cosine
        keys = [str(k) for k in ex["keys"]]
        vecs = {}
        for k in keys:
            signal = torch.tensor(ex[f"pcm16_{k}"][:64000].astype(numpy.float32) / 32768.0, dtype=torch.float32).to("cuda")
            _, vec = xtractor(signal, is_eval=True)
            vecs[k] = vec.detach().cpu().numpy()
            assert rel(vecs[k], ex[f"emb_first4s_{k}_unpinned_frontend"]) < TOL
        # scoring callers: StatServer by attribute assignment (xvector.py:1905-1914), cosine_scoring, EER
        enroll = StatServer()
        enroll.modelset = numpy.array(keys).astype('>U')
        enroll.segset = numpy.array(keys).astype('>U')
        enroll.start = numpy.empty(3, dtype="|O")
        enroll.stop = numpy.empty(3, dtype="|O")
        enroll.stat0 = numpy.ones((3, 1))
        enroll.stat1 = numpy.concatenate([vecs[k] for k in keys]).astype(numpy.float64)
        ndx = Ndx(models=numpy.repeat(numpy.array(keys), 3), testsegs=numpy.tile(numpy.array(keys), 3))
        scores = cosine_scoring(enroll, enroll, ndx, wccn=None, check_missing=True)
        assert isinstance(scores, Scores) and scores.scoremat.dtype == numpy.float32 and scores.scoremat.shape == (3, 3)
        assert numpy.allclose(numpy.diag(scores.scoremat), 1.0, atol=1e-5)
        want = enroll.stat1 @ enroll.stat1.T
        assert numpy.allclose(scores.scoremat, want, atol=2e-6)
        pm, pf = rocch(numpy.diag(scores.scoremat).astype(float), scores.scoremat[~numpy.eye(3, dtype=bool)].astype(float))
        assert rocch2eer(pm, pf) == 0
        snorm = asnorm(torch.from_numpy(enroll.stat1).float(), torch.randn(250, 256), ndx)
        assert snorm.shape == (3, 3)
        # HDF5 round trip through the aliased modules
        enroll.write(str(tmp_path / "enroll.h5"))
        assert numpy.allclose(StatServer(str(tmp_path / "enroll.h5")).stat1, enroll.stat1, atol=1e-7)
        with pytest.raises(AssertionError):
            cosine_scoring(enroll, "not a StatServer", ndx)
    finally:
        for k in [k for k in sys.modules if k == "sidekit" or k.startswith("sidekit.")]:
            del sys.modules[k]
        sys.modules.update(saved)


# per-stage relative-error budgets of the bf16 trunk against the fp32 oracle: bf16 rounding (2^-9 per stored value) grows with
# depth; a stage that breaks its budget localises a defect that the final cosine would wash out
# measured (tests/tools/bf16_stage_errors.py, T = 52 .. 801): stem 1.66e-3, layer1 4.4e-3, layer2 5.9e-3, layer3 8.5e-3 .. 9.3e-3, layer4 1.1e-2 .. 1.5e-2
BF16_BUDGET = {"stem": 2.5e-3, "layer1": 7e-3, "layer2": 9e-3, "layer3": 1.4e-2, "layer4": 2.2e-2}


def _bf16(buf):
    return torch.from_numpy((buf.view(numpy.uint16).astype(numpy.uint32) << 16).view(numpy.float32).copy())


@pytest.mark.parametrize("case", ["t401", "ragged"])
def test_bf16_stage_taps_against_the_fp32_oracle(gpu, case):
    sd = seeded_state_dict("halfresnet34", 16, seed=1234)
    model = Xtractor(16, model_archi="halfresnet34", loss="aam", seed=0).to(gpu).eval()
    model.load_state_dict(sd, strict=True)
    model.compute_dtype = "bf16"
    g = torch.Generator().manual_seed(55)
    frames = [401, 401] if case == "t401" else [401, 137, 260, 52]
    T = max(frames)
    feats = torch.randn(len(frames), 80, T, generator=g)
    names = ["stem", "layer1", "layer2", "layer3", "layer4"]
    model.set_debug(True)
    _, emb = model.forward_features(feats.cuda(), frames=frames if case == "ragged" else None)
    raw = model.debug_taps(names)
    model.set_debug(False)
    for b, t in enumerate(frames):
        taps = {}
        with torch.no_grad():
            _, o_emb = oxv.halfresnet34_from_feats(feats[b:b + 1, :, :t], sd, taps=taps)
        for li, name in enumerate(names):
            ref = taps[name][0]                                            # (C, H, W) of this utterance
            C, H, W = ref.shape
            Hmax = T
            for _ in range(max(li - 1, 0)):
                Hmax = (Hmax + 1) // 2
            got = _bf16(raw[name]).reshape(len(frames), Hmax, W, C)[b, :H].permute(2, 0, 1)
            err = rel(got, ref)
            assert err < BF16_BUDGET[name], (case, b, name, err)
            assert err > 1e-4 or name == "stem", (case, b, name, err)      # really the bf16 path
        cos = float(torch.nn.functional.cosine_similarity(emb[b:b + 1].cpu(), o_emb))
        assert cos > 0.9995, (case, b, cos)                              # measured 1 - cos = 5e-5 .. 7e-5
