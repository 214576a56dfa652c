"""End-to-end drivers on the GPU: extract_xvectors (wav.scp -> ark/scp) -> compute_spk_cosine -> EER,
the chain of egs/vpc2020_decode/local/compute_metrics.sh:59-77, checked against the oracle."""
import os

import numpy
import pytest
import scipy.io.wavfile
import torch

from oracle import scoring as osc
from oracle import xvector as oxv
from sidekit_amd.bin import compute_metrics, compute_spk_cosine, extract_xvectors
from sidekit_amd.kaldi_io import read_scp
from sidekit_amd.nnet.weights import seeded_state_dict

pytestmark = pytest.mark.gpu


def test_extract_score_eer_chain(gpu, tmp_path):
    rs = numpy.random.RandomState(4)
    n_spk = 16
    sd = seeded_state_dict("halfresnet34", n_spk, seed=77)
    ckpt = {"speaker_number": n_spk, "model_archi": {"model_type": "halfresnet34", "loss": {"type": "aam"}}, "model_state_dict": sd}
    torch.save(ckpt, tmp_path / "model.pt")
    # six utterances of three "speakers", different lengths, PCM16 wav files; one entry goes through a shell pipe
    utts, waves = [], {}
    for s in range(3):
        for u in range(2):
            key = f"spk{s}-utt{u}"
            x = (0.1 * rs.randn(rs.randint(12000, 40000)) * 32768).clip(-32768, 32767).astype(numpy.int16)
            scipy.io.wavfile.write(tmp_path / f"{key}.wav", 16000, x)
            waves[key] = torch.from_numpy(x.astype(numpy.float32) / 32768.0)
            utts.append(key)
    with open(tmp_path / "wav.scp", "w") as f:
        for i, key in enumerate(utts):
            p = tmp_path / f"{key}.wav"
            f.write(f"{key} cat {p} |\n" if i == 1 else f"{key} {p}\n")
    with open(tmp_path / "spk2utt", "w") as f:
        for s in range(3):
            f.write(f"spk{s} spk{s}-utt0 spk{s}-utt1\n")
    extract_xvectors.cli(["--model", str(tmp_path / "model.pt"), "--wav-scp", str(tmp_path / "wav.scp"), "--out-scp",
                          str(tmp_path / "xv.scp"), "--out-spk-scp", str(tmp_path / "spk_xv.scp"), "--spk2utt-file",
                          str(tmp_path / "spk2utt"), "--device", "cuda", "--batch-size", "4"])
    got = dict(read_scp(str(tmp_path / "xv.scp")))
    assert list(got) == utts                                        # wav.scp order, one (1, 256) float matrix each
    with torch.no_grad():
        _, ref = oxv.forward_ragged([waves[k] for k in utts], sd)
    for i, k in enumerate(utts):
        assert got[k].shape == (1, 256) and got[k].dtype == numpy.float32
        assert numpy.linalg.norm(got[k] - ref[i].numpy()) / numpy.linalg.norm(ref[i].numpy()) < 1e-4
    spk = dict(read_scp(str(tmp_path / "spk_xv.scp")))
    for s in range(3):
        m = numpy.mean([got[f"spk{s}-utt0"], got[f"spk{s}-utt1"]], axis=0)
        numpy.testing.assert_allclose(spk[f"spk{s}"], m / numpy.linalg.norm(m), atol=1e-6)
    # listed trials: enrol speaker vs test utterance
    with open(tmp_path / "utt2spk", "w") as f:
        for k in utts:
            f.write(f"{k} {k.split('-')[0]}\n")
    trials = [(f"spk{s}", k, "target" if k.startswith(f"spk{s}-") else "nontarget") for s in range(3) for k in utts]
    with open(tmp_path / "trials", "w") as f:
        for e, t, lab in trials:
            f.write(f"{e} {t} {lab}\n")
    compute_spk_cosine.cli([str(tmp_path / "trials"), str(tmp_path / "utt2spk"), str(tmp_path / "xv.scp"), str(tmp_path / "xv.scp"),
                            str(tmp_path / "scores")])
    lines = [l.split() for l in open(tmp_path / "scores")]
    assert [(l[0], l[1]) for l in lines] == [(e, t) for e, t, _ in trials]
    for (e, t, _), l in zip(trials, lines):
        mean = numpy.mean([got[f"{e}-utt0"].reshape(-1), got[f"{e}-utt1"].reshape(-1)], axis=0)
        mean /= numpy.linalg.norm(mean)
        v = got[t].reshape(-1).astype(numpy.float64)
        want = float(mean.astype(numpy.float64) @ v / (numpy.linalg.norm(mean.astype(numpy.float64)) * numpy.linalg.norm(v)))
        assert abs(float(l[2]) - want) < 1e-6
    eer = compute_metrics.eer_from_files(str(tmp_path / "scores"), str(tmp_path / "trials"))
    tar = numpy.array([float(l[2]) for l, (_, _, lab) in zip(lines, trials) if lab == "target"])
    non = numpy.array([float(l[2]) for l, (_, _, lab) in zip(lines, trials) if lab == "nontarget"])
    assert abs(eer - osc.eer(tar, non)) < 1e-12
    with pytest.raises(NotImplementedError):
        extract_xvectors.cli(["--model", str(tmp_path / "model.pt"), "--wav-scp", str(tmp_path / "wav.scp"), "--out-scp",
                              str(tmp_path / "xv2.scp"), "--vad"])
