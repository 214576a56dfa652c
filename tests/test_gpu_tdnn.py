"""TDNN x-vector (model_archi='xvector') parity on the GPU through the C ABI."""
import os

import numpy
import pytest
import torch

from oracle import frontend as ofe
from oracle import xvector as oxv
from sidekit_amd.nnet import Xtractor
from sidekit_amd.nnet.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def fx(golden_dir):
    return numpy.load(os.path.join(golden_dir, "tdnn.npz"))


def _model(gpu, fx, loss):
    m = Xtractor(int(fx["n_spk"]), model_archi="xvector", loss=loss, seed=0).to(gpu).eval()
    m.load_state_dict(seeded_state_dict("xvector", int(fx["n_spk"]), loss=loss, seed=int(fx["seed"])), strict=True)
    return m


def _feats(fx, key):
    g = torch.Generator().manual_seed(int(fx[f"{key}_feat_seed"]))
    return torch.randn(*[int(s) for s in fx[f"{key}_shape"]], generator=g)


@pytest.mark.parametrize("tag", ["t63", "t126"])
def test_golden_aam(gpu, fx, tag):
    m = _model(gpu, fx, "aam")
    feats = _feats(fx, f"aam_{tag}")
    m.set_debug(True)
    logits, emb = m.forward_features(feats.cuda())
    raw = m.debug_taps(["pooled", "pre_norm"])
    assert rel(raw["pooled"].view(numpy.float32), fx[f"{tag}_pooled"]) < TOL
    assert rel(raw["pre_norm"].view(numpy.float32), fx[f"{tag}_pre_norm"]) < TOL
    assert rel(emb, fx[f"aam_{tag}_emb"]) < TOL and rel(logits, fx[f"aam_{tag}_logits"]) < TOL


def test_golden_cce_returns_embedding_only(gpu, fx):
    m = _model(gpu, fx, "cce")
    out = m.forward_features(_feats(fx, "cce_t63").cuda())
    assert torch.is_tensor(out) and rel(out, fx["cce_t63_emb"]) < TOL          # xvector.py:896-898
    raw = m.forward_features(_feats(fx, "cce_t63").cuda(), norm_embedding=False)
    assert rel(torch.nn.functional.normalize(raw, dim=1), fx["cce_t63_emb"]) < TOL
    assert float(raw.norm(dim=1).min()) > 1.5                                  # really un-normalised


def test_wav_variable_length_batch(gpu, fx):
    """BASELINE config 4 in miniature: ragged 2-10 s batch, parity per utterance against the oracle run alone."""
    m = _model(gpu, fx, "aam")
    sd = m.state_dict()
    rs = numpy.random.RandomState(0)
    lens = rs.randint(32000, 160001, (6,)).tolist()
    torch.manual_seed(8)
    wav = 0.1 * torch.randn(len(lens), max(lens))
    with torch.no_grad():
        _, ref = oxv.forward_ragged([wav[i, :n] for i, n in enumerate(lens)], sd, arch="xvector")
    _, emb = m(wav.cuda(), is_eval=True, lengths=lens)
    for i in range(len(lens)):
        assert rel(emb[i], ref[i]) < TOL, (i, lens[i])
    f = m.features(wav[:2, :64000].cuda())
    assert f.shape == (2, 80, 126) and rel(f, ofe.mfcc_frontend(wav[:2, :64000])) < TOL
    with pytest.raises(ValueError, match="context"):
        m.forward_features(torch.randn(1, 80, 14).cuda())
    with pytest.raises(NotImplementedError):
        m.compute_dtype = "bf16"
        try:
            m(wav[:1].cuda(), is_eval=True)
        finally:
            m.compute_dtype = None


@pytest.mark.ab_variant
def test_mfcc_fft_matches_the_dft_contraction(gpu, fx, monkeypatch):
    """The MFCC spectrum as a 2048-point real FFT (frontend_fft.hip) against the round-1 form, a (frames x 1024) x (1024 x 2050)
    DFT contraction on the exact-f32 matrix cores (SIDEKIT_AMD_MFCC_DFT_GEMM=1), and both against the oracle: ragged lengths
    with reflect-padded edges, features after CMVN."""
    lens = [16000 + 17, 40000, 14 * 512 + 5, 64000]       # incl. the shortest utterance the TDNN context admits (15 frames)
    torch.manual_seed(9)
    wav = 0.1 * torch.randn(len(lens), max(lens))
    m_fft = _model(gpu, fx, "aam")
    monkeypatch.setenv("SIDEKIT_AMD_MFCC_DFT_GEMM", "1")
    m_dft = _model(gpu, fx, "aam")
    monkeypatch.delenv("SIDEKIT_AMD_MFCC_DFT_GEMM")
    for i, n in enumerate(lens):
        ref = ofe.mfcc_frontend(wav[i:i + 1, :n])
        a = m_fft.features(wav[i:i + 1, :n].cuda())
        b = m_dft.features(wav[i:i + 1, :n].cuda())
        assert a.shape == ref.shape and rel(a, ref) < TOL and rel(b, ref) < TOL and rel(a, b) < TOL, (i, n)
    fa = m_fft.features(wav.cuda(), lengths=lens)         # the same utterances as one ragged batch
    for i, n in enumerate(lens):
        t = 1 + n // 512
        assert rel(fa[i, :, :t], ofe.mfcc_frontend(wav[i:i + 1, :n])[0]) < TOL, (i, n)


@pytest.mark.ab_variant
def test_large_gemm_tiling_is_bit_identical(gpu, fx, monkeypatch):
    """128 x 128 tiles (problems of >= 2048 rows) against the 64 x 64 kernel (SIDEKIT_AMD_GEMM64=1): the same k-ordered FMA
    chain per output element, so the TDNN forward of a 40-utterance ragged batch (7k rows) must not move by one bit."""
    m = _model(gpu, fx, "aam")
    lens = numpy.random.RandomState(5).randint(32000, 160001, (40,)).tolist()
    g = torch.Generator(device="cuda").manual_seed(11)
    wav = 0.1 * torch.randn(len(lens), max(lens), device="cuda", generator=g)
    logits_a, emb_a = m(wav, is_eval=True, lengths=lens)
    monkeypatch.setenv("SIDEKIT_AMD_GEMM64", "1")
    logits_b, emb_b = m(wav, is_eval=True, lengths=lens)
    monkeypatch.delenv("SIDEKIT_AMD_GEMM64")
    assert torch.equal(emb_a, emb_b) and torch.equal(logits_a, logits_b)
