"""HalfResNet34 parity on the GPU, through the C ABI: golden fixtures (reference outputs), oracle on
fresh seeded inputs, per-utterance semantics of ragged batches, bf16 drift, error behaviour."""
import os

import numpy
import pytest
import torch

from oracle import frontend as ofe
from oracle import xvector as oxv
from sidekit_amd.nnet import Xtractor
from sidekit_amd.nnet.weights import seeded_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north-star: embeddings within 1e-4 relative of the reference in fp32


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return ((a - b).norm() / b.norm()).item()


def bf16_to_f32(buf):
    return torch.from_numpy((buf.view(numpy.uint16).astype(numpy.uint32) << 16).view(numpy.float32).copy())


@pytest.fixture(scope="module")
def fx(golden_dir):
    return numpy.load(os.path.join(golden_dir, "halfresnet34.npz"))


@pytest.fixture(scope="module")
def model(gpu, fx):
    m = Xtractor(int(fx["n_spk"]), model_archi="halfresnet34", loss="aam", seed=0).to(gpu).eval()
    m.load_state_dict(seeded_state_dict("halfresnet34", int(fx["n_spk"]), seed=int(fx["seed"])), strict=True)
    m.compute_dtype = "fp32"
    return m


def _feats(fx, tag):
    g = torch.Generator().manual_seed(int(fx[f"{tag}_feat_seed"]))
    return torch.randn(*[int(s) for s in fx[f"{tag}_shape"]], generator=g)


@pytest.mark.parametrize("tag", ["small", "odd", "len4s"])
def test_golden_features_to_embedding(model, fx, tag):
    feats = _feats(fx, tag)
    B, _, T = feats.shape
    model.set_debug(True)
    logits, emb = model.forward_features(feats.cuda())
    raw = model.debug_taps(["layer4", "pooled", "pre_norm"])
    model.set_debug(False)
    ref4 = torch.from_numpy(fx[f"{tag}_layer4"])                    # (B, 256, T', 10)
    l4 = torch.from_numpy(raw["layer4"].view(numpy.float32).copy()).reshape(B, ref4.shape[2], 10, 256).permute(0, 3, 1, 2)
    assert rel(l4, ref4) < TOL
    pooled = torch.from_numpy(raw["pooled"].view(numpy.float32).copy()).reshape(B, 2, 10, 256).permute(0, 1, 3, 2).reshape(B, 5120)
    assert rel(pooled, fx[f"{tag}_pooled"]) < TOL
    assert rel(raw["pre_norm"].view(numpy.float32), fx[f"{tag}_pre_norm"]) < TOL
    assert rel(emb, fx[f"{tag}_emb"]) < TOL
    assert rel(logits, fx[f"{tag}_logits"]) < TOL
    assert emb.shape == (B, 256) and logits.shape == (B, int(fx["n_spk"])) and emb.is_cuda
    assert torch.allclose(emb.norm(dim=1), torch.ones(B, device=emb.device), atol=1e-5)


def _coloured(seed, n_utt, n):
    """Speech-like dynamic range: white noise through a one-pole low-pass (high bands ~40 dB under the low ones) plus two tones."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(n_utt, n, generator=g, dtype=torch.float64)
    x = torch.zeros_like(w)
    acc = torch.zeros(n_utt, dtype=torch.float64)
    for i in range(n):                                               # y[i] = 0.95 y[i-1] + w[i]
        acc = 0.95 * acc + w[:, i]
        x[:, i] = acc
    t = torch.arange(n, dtype=torch.float64) / 16000.0
    x = 0.02 * x + 0.3 * torch.sin(2 * numpy.pi * 440.0 * t) + 0.05 * torch.sin(2 * numpy.pi * 3000.0 * t)
    return x


def test_front_end_accuracy_against_the_float64_oracle(model, capsys):
    """The f32 front-end kernel (real FFT, fused mel projection, log, CMVN) against the oracle evaluated in float64 on signals with a
    large dynamic range, ragged lengths included (reflect-padded first / last frames, frames beyond the end): the error budget of
    the f32 arithmetic itself, independent of the oracle's own f32 rounding."""
    lens = [16000, 15999, 8000 + 77, 600]
    x64 = _coloured(3, len(lens), max(lens))
    got = model.features(x64.float().cuda(), lengths=lens).double().cpu()
    worst = 0.0
    for i, n in enumerate(lens):
        ref = ofe.melspec_frontend(x64[i, :n].float().double())[0]      # the same f32 samples, float64 arithmetic from there on
        t = 1 + n // 160
        err = (got[i, :, :t] - ref).abs().max().item()
        worst = max(worst, err)
        assert err < 1e-4, (i, n, err)                                  # CMVN'ed log-mel values are O(1)
    with capsys.disabled():
        print(f"  [front-end vs float64 oracle: max abs error {worst:.2e}]", end="")


def test_golden_wav_to_embedding(model, fx):
    x = torch.from_numpy(fx["wav_pcm16"].astype(numpy.float32) / 32768.0)
    feats = model.preprocessor(x.cuda(), is_eval=True)                # MelSpecFrontEnd.forward
    assert feats.shape == (1, 80, 201)
    assert rel(feats, fx["wav_feats_unpinned_frontend"]) < TOL
    _, emb = model(x.cuda(), is_eval=True)                            # 1-D input like extract_xvectors.py:146
    assert rel(emb, fx["wav_emb_unpinned_frontend"]) < TOL
    # N4: int16-scaled input gives (nearly) the same x-vector; only the +1e-6 inside the log breaks exact invariance
    _, emb2 = model((x * 32768.0).cuda(), is_eval=True)
    assert float(torch.nn.functional.cosine_similarity(emb2, emb)) > 0.9999


def test_stagewise_against_oracle(model):
    sd = model.state_dict()
    g = torch.Generator().manual_seed(101)
    feats = torch.randn(3, 80, 96, generator=g)
    taps = {}
    with torch.no_grad():
        o_logits, o_emb = oxv.halfresnet34_from_feats(feats, sd, taps=taps)
    model.set_debug(True)
    logits, emb = model.forward_features(feats.cuda())
    raw = model.debug_taps(["stem", "layer1", "layer2", "layer3", "layer4"])
    model.set_debug(False)
    for name in ("stem", "layer1", "layer2", "layer3", "layer4"):
        ref = taps[name]
        B, C, H, W = ref.shape
        x = torch.from_numpy(raw[name].view(numpy.float32).copy()).reshape(B, H, W, C).permute(0, 3, 1, 2)
        assert rel(x, ref) < TOL, name
    assert rel(emb, o_emb) < TOL and rel(logits, o_logits) < TOL


def test_ragged_batch_is_per_utterance(model):
    """a zero-padded batch must reproduce each utterance run alone (CMVN, conv padding, pooling
    over its own length) -- including lengths that are not multiples of the row tiles."""
    sd = model.state_dict()
    torch.manual_seed(5)
    lens = [40000, 16123, 9000, 31999, 700]
    wav = 0.1 * torch.randn(len(lens), max(lens))
    with torch.no_grad():
        _, ref = oxv.forward_ragged([wav[i, :n] for i, n in enumerate(lens)], sd)
    padded = wav.clone()
    for i, n in enumerate(lens):
        padded[i, n:] = 7.0                                       # garbage in the padding must not leak
    _, emb = model(padded.cuda(), is_eval=True, lengths=lens)
    for i in range(len(lens) - 1):
        assert rel(emb[i], ref[i]) < TOL, (i, lens[i])
    # 700 samples -> 5 frames -> one pooled frame: the unbiased std of the global context is 0/0 and the reference
    # returns a NaN x-vector; so does this build (NaN must survive the ReLU like torch's)
    assert bool(torch.isnan(ref[4]).all()) and bool(torch.isnan(emb[4]).all())
    # and the batched result equals the single-utterance call bit for bit (same kernels, same order)
    _, one = model(wav[1, :lens[1]].cuda(), is_eval=True)
    assert torch.equal(one[0], emb[1])
    # features entry with per-utterance frame counts
    f = ofe.melspec_frontend(wav[:2, :16000])
    with torch.no_grad():
        _, r0 = oxv.halfresnet34_from_feats(f[:1, :, :60], sd)
    _, e = model.forward_features(f.cuda(), frames=[60, 101])
    assert rel(e[0], r0[0]) < TOL


def test_batch_invariance_and_determinism(model):
    torch.manual_seed(6)
    wav = 0.1 * torch.randn(1, 32000).cuda()
    _, a = model(wav.repeat(5, 1), is_eval=True)
    _, b = model(wav, is_eval=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[4], b[0])
    _, c = model(wav.repeat(5, 1), is_eval=True)
    assert torch.equal(a, c)                                          # SE partial sums are reduced in a fixed order


def test_bf16_trunk_tracks_fp32(model):
    torch.manual_seed(7)
    wav = 0.1 * torch.randn(4, 48000).cuda()
    _, e32 = model(wav, is_eval=True)
    model.compute_dtype = "bf16"
    try:
        _, e16 = model(wav, is_eval=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):           # reference-style mixed precision switch (xvector.py:1890)
            model.compute_dtype = None
            _, e16b = model(wav, is_eval=True)
    finally:
        model.compute_dtype = "fp32"
    assert torch.equal(e16, e16b)
    cos = torch.nn.functional.cosine_similarity(e16, e32)
    assert float(cos.min()) > 0.999, cos
    assert 1e-4 < rel(e16, e32) < 5e-2                                # really a different precision, but close


def test_error_behaviour(model):
    with pytest.raises(RuntimeError, match="input is on cpu"):
        model(torch.zeros(16000), is_eval=True)
    with pytest.raises(NotImplementedError):
        model(torch.zeros(16000).cuda(), is_eval=False)
    with pytest.raises(ValueError):                                   # torch.stft reflect padding needs > n_fft/2 samples
        model(torch.zeros(300).cuda(), is_eval=True)
    with pytest.raises(ValueError):
        model(torch.zeros(2, 16000).cuda(), is_eval=True, lengths=[16000, 20000])
    with pytest.raises(RuntimeError, match="expected"):
        model(torch.zeros(1, 2, 16000).cuda(), is_eval=True)
    nan = torch.zeros(16000).cuda()                                   # a silent utterance: constant features -> CMVN 0/eps, finite
    _, e = model(nan, is_eval=True)
    assert bool(torch.isfinite(e).all())


def test_row_tile_boundaries(model):
    """Frame counts around the conv row tiles (8 / 16 rows) and the stride-2 halvings: every length alone vs the oracle."""
    sd = model.state_dict()
    frames = [9, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129]
    lens = [(t - 1) * 160 + 7 for t in frames]          # T = 1 + L // 160
    lens = [max(n, 513) for n in lens]                  # reflect padding needs > 512 samples
    torch.manual_seed(12)
    wav = 0.1 * torch.randn(len(lens), max(lens))
    with torch.no_grad():
        _, ref = oxv.forward_ragged([wav[i, :n] for i, n in enumerate(lens)], sd)
    _, emb = model(wav.cuda(), is_eval=True, lengths=lens)
    for i, n in enumerate(lens):
        assert rel(emb[i], ref[i]) < TOL, (i, n, 1 + n // 160)


def test_bf16_row_tile_boundaries_and_per_utterance(model):
    """The bf16 path has its own tilings (17-row layer4 tiles, three workgroups per CU, persistent layer1 workgroups, the
    in-place first-block shortcut): frame counts around every row-tile edge, batched (ragged, garbage in the padding) vs
    each utterance alone -- bit for bit -- and against the fp32 path."""
    frames = [17 * 8, 17 * 8 + 1, 34 * 8, 34 * 8 + 1, 51 * 8 - 7, 51 * 8 + 1, 65, 129, 401, 64]
    lens = [(t - 1) * 160 + 11 for t in frames]
    torch.manual_seed(13)
    wav = 0.1 * torch.randn(len(lens), max(lens))
    padded = wav.clone()
    for i, n in enumerate(lens):
        padded[i, n:] = -3.0
    _, e32 = model(padded.cuda(), is_eval=True, lengths=lens)
    model.compute_dtype = "bf16"
    try:
        _, e16 = model(padded.cuda(), is_eval=True, lengths=lens)
        assert bool(torch.isfinite(e16).all())
        cos = torch.nn.functional.cosine_similarity(e16, e32)
        assert float(cos.min()) > 0.999, cos
        for i in (0, 1, 3, 5, 8):
            _, one = model(wav[i, :lens[i]].cuda(), is_eval=True)
            assert torch.equal(one[0], e16[i]), (i, frames[i])
    finally:
        model.compute_dtype = "fp32"


@pytest.mark.ab_variant
def test_ab_paths_agree(gpu, monkeypatch):
    """The A/B switches of the -DSK_AB build (shortcut as a stored tensor / mel projection as a GEMM / attention unfused) are the older
    formulations of the same arithmetic: they must agree with the default path (bf16 bit for bit on the shortcut side)."""
    torch.manual_seed(21)
    wav = 0.1 * torch.randn(3, 30000).cuda()
    lens = [30000, 17000, 22222]
    base = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=5).to(gpu).eval()
    long_wav, long_lens = 0.1 * torch.randn(2, 200000).cuda(), [200000, 150001]      # T' = 157 / 118: several 64-row chunks in the fused pooling
    out, out_long = {}, {}
    for dtype in ("fp32", "bf16"):
        base.compute_dtype = dtype
        out[dtype] = base(wav, is_eval=True, lengths=lens)[1]
        out_long[dtype] = base(long_wav, is_eval=True, lengths=long_lens)[1]
    for var in ("SIDEKIT_AMD_SHORTCUT_TENSOR", "SIDEKIT_AMD_MEL_GEMM", "SIDEKIT_AMD_ATT_SEPARATE"):
        monkeypatch.setenv(var, "1")
        alt = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=5).to(gpu).eval()
        for dtype in ("fp32", "bf16"):
            alt.compute_dtype = dtype
            e = alt(wav, is_eval=True, lengths=lens)[1]                 # the switch is read when the native handle is made (first forward)
            assert rel(e, out[dtype]) < (2e-5 if dtype == "fp32" else 2e-2), (var, dtype)
            if var == "SIDEKIT_AMD_ATT_SEPARATE":   # the fused attention / statistics kernel computes the same scores: only the order of the time sums differs
                assert rel(e, out[dtype]) < 2e-6, (var, dtype, rel(e, out[dtype]))
                e_long = alt(long_wav, is_eval=True, lengths=long_lens)[1]
                assert rel(e_long, out_long[dtype]) < 2e-6, (var, dtype, rel(e_long, out_long[dtype]))
        monkeypatch.delenv(var)


def test_profile_slots(model):
    """xt_set_profile: every kernel class, or only the named ones (bench.py brackets the dominant class in its timed region)."""
    wav = 0.1 * torch.randn(16, 16000).cuda()
    model.set_profile(True)
    model.get_profile(reset=True)
    model(wav, is_eval=True)
    full = model.get_profile(reset=True)
    assert {"conv_L1", "conv_L4", "frontend", "stem", "se_residual", "pool_tail"} <= set(full) and full["conv_L1"][1] == 6 and full["se_residual"][1] == 16
    model(wav[:2], is_eval=True)                      # at most 12 utterances: conv2 of layers 3-4 runs its small-grid tiling, booked under the same class
    small = model.get_profile(reset=True)
    assert small["se_residual"][1] == 16 and small["conv_L1"][1] == 6 and small["conv_L3"][1] == 11 and small["conv_L4"][1] == 5
    model.set_profile(True, slots=["stem", "conv_L3"])
    model(wav, is_eval=True)
    part = model.get_profile(reset=True)
    assert set(part) == {"stem", "conv_L3"} and part["conv_L3"][1] == 11 and part["stem"][0] > 0
    model.set_profile(False)
    model(wav, is_eval=True)
    assert model.get_profile(reset=True) == {}


def test_embedding_does_not_depend_on_the_batch_size(gpu):
    """An utterance's embedding is bit-identical whether 7, 256 or 600 utterances share its batch: the embedding GEMMs cut K
    into the same slices for every M (above 512 rows the slices run inside the workgroup instead of as blockIdx.z)."""
    m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=77).to(gpu).eval()
    g = torch.Generator(device="cuda").manual_seed(5)
    wav = 0.1 * torch.randn(600, 16000, device="cuda", generator=g)
    for dt in ("fp32", "bf16"):
        m.compute_dtype = dt
        big = m(wav, is_eval=True)[1]
        parts = torch.cat([m(wav[:7], is_eval=True)[1], m(wav[7:263], is_eval=True)[1], m(wav[263:], is_eval=True)[1]])
        assert torch.equal(big, parts), (dt, float((big - parts).abs().max()))


SMALL_GRID_CASES = [(1, 64000, False), (1, 45 * 16000, False), (3, 48000, True), (8, 64000, False), (8, 160000, True), (12, 64000, False), (13, 64000, False),
                    (12, 48000, True), (40, 32000, True)]     # 12 / 13: either side of xt_handle::SMALL_GRID_MAX_B (csrc/xt_api.hip)


def _small_grid_check(gpu, monkeypatch, settings, cases):
    """models with the same weights under each (SIDEKIT_AMD_SMALL_GRID, SIDEKIT_AMD_GATE_PROLOGUE) setting -> same bits as the first one, every case, both precisions"""
    models = {}
    for key, (grid, gate) in settings.items():
        monkeypatch.setenv("SIDEKIT_AMD_GATE_PROLOGUE", gate)
        monkeypatch.setenv("SIDEKIT_AMD_SMALL_GRID", grid)
        m = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=41).to(gpu).eval()
        m.compute_dtype = "fp32"; m(torch.zeros(1, 4000, device="cuda") + 0.01, is_eval=True)      # the handles are created under this setting
        m.compute_dtype = "bf16"; m(torch.zeros(1, 4000, device="cuda") + 0.01, is_eval=True)
        models[key] = m
    ref = next(iter(models))
    g = torch.Generator(device="cuda").manual_seed(17)
    for B, L, ragged in cases:
        wav = 0.1 * torch.randn(B, L, device="cuda", generator=g)
        lens = torch.randint(L // 4, L + 1, (B,), generator=torch.Generator().manual_seed(B * 7 + 1)).tolist() if ragged else None
        for dt in ("bf16", "fp32"):
            outs = {}
            for key, m in models.items():
                m.compute_dtype = dt
                outs[key] = m(wav, is_eval=True, lengths=lens)
            for key in models:
                assert torch.equal(outs[key][1], outs[ref][1]) and torch.equal(outs[key][0], outs[ref][0]), (B, L, ragged, dt, key)
    torch.cuda.synchronize()


def test_small_grid_forms_give_the_bits_of_the_batch_forms(gpu, monkeypatch):
    """Small batches (at most 12 utterances, xt_handle::SMALL_GRID_MAX_B; 1 is the reference driver's call shape, sidekit/bin/extract_xvectors.py:146) run
    conv2 of layers 3-4 in 3- / 2-row tiles (csrc/conv3x3.hip, "Small-grid forms").  Forced off / on and chosen automatically (SIDEKIT_AMD_SMALL_GRID = 0 /
    2 / 1) on models with the same weights, x-vectors and logits are the same bits at every batch size -- batch 1 at 4 s and 45 s (563 row tiles), ragged
    batches, batches of 12 and 13 (either side of the threshold), a batch of 40 -- in both precisions."""
    _small_grid_check(gpu, monkeypatch, {"never": ("0", "0"), "always": ("2", "0"), "auto": ("1", "0")}, SMALL_GRID_CASES)


@pytest.mark.ab_variant
def test_in_convolution_gate_forms_give_the_bits_of_the_launch(gpu, monkeypatch):
    """A/B build only (csrc/se_gate_inl.h; both forms measured slower than the launch in round 5): two ways of computing the SE gate
    INSIDE conv2 instead of by the launch of ``se_pre_kernel`` between conv1 and conv2 (sidekit/nnet/res_net.py:272-281,316-319) -- a fifth wave of
    conv2 (layers 1-2) and a prologue in which every workgroup of an utterance walks the 1024 virtual threads of ``se_pre_kernel`` on its 256 real ones
    (every layer; selected for at most 8 utterances, xt_handle::GATE_AB_MAX_B, or always).  Same bits as the launch at every batch size."""
    _small_grid_check(gpu, monkeypatch, {"launch": ("0", "0"), "wave always": ("2", "2"), "wave small": ("1", "1"), "prologue always": ("1", "4"), "prologue small": ("1", "3")},
                      [c for c in SMALL_GRID_CASES if c[0] != 13])


@pytest.mark.ab_variant
def test_layer1_pair_kernel_gives_the_bits_of_the_two_launches(gpu, monkeypatch):
    """A/B build only (round 6; measured SLOWER than the two launches, profiles/r06_conv_pair_L1.txt, so the product does not carry it): conv2 of
    block k and conv1 of block k + 1 of layer 1 as ONE kernel (csrc/conv_pair.hip: the block output reaches the next conv1 through LDS;
    sidekit/nnet/res_net.py:309-320, two consecutive blocks; SIDEKIT_AMD_PAIR=1) against the two stand-alone launches: the stage taps after every
    layer, the x-vectors and the logits are the same BITS -- the pair kernel's tiles, MFMA order, epilogue arithmetic and SE-sum order are the
    stand-alone kernels'.  Batches of 1 .. 256; uniform, ragged, clips shorter than one 8-row tile and lengths that leave one valid row in the last
    tile; a 45-s utterance (563 row tiles); pipelined submits."""
    def make(env):
        if env:
            monkeypatch.setenv("SIDEKIT_AMD_PAIR", "1")
        m = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=43).to(gpu).eval()
        m.compute_dtype = "bf16"
        m(torch.zeros(1, 4000, device="cuda") + 0.01, is_eval=True)
        if env:
            monkeypatch.delenv("SIDEKIT_AMD_PAIR")
        return m
    plain, paired = make(False), make(True)          # the switch is read per forward: set around every call of `paired`
    g = torch.Generator(device="cuda").manual_seed(19)
    names = ["stem", "layer1", "layer2", "layer3", "layer4"]

    def run(m, env, wav, lens):
        if env:
            monkeypatch.setenv("SIDEKIT_AMD_PAIR", "1")
        m.set_debug(True)
        logits, emb = m(wav, is_eval=True, lengths=lens)
        taps = m.debug_taps(names)
        m.set_debug(False)
        if env:
            monkeypatch.delenv("SIDEKIT_AMD_PAIR")
        return logits, emb, taps

    cases = [(1, 64000, None), (3, 48000, [48000, 1290, 31999]), (2, 2400, [2400, 1130]),            # 16 / 8 frames: two tiles / exactly one tile
             (5, 20000, [20000, 1280 + 159, 1280 * 2, 1280 * 2 + 160, 19999]),                      # 9, 17, 18 rows: one / two valid rows in the last tile
             (1, 45 * 16000, None), (40, 32000, "ragged"), (256, 16000, None), (130, 24000, "ragged")]
    for B, L, lens in cases:
        wav = 0.1 * torch.randn(B, L, device="cuda", generator=g)
        if lens == "ragged":
            lens = torch.randint(L // 4, L + 1, (B,), generator=torch.Generator().manual_seed(B * 5 + 3)).tolist()
        (la, ea, ta), (lb, eb, tb) = run(paired, True, wav, lens), run(plain, False, wav, lens)
        for n in names:
            assert numpy.array_equal(ta[n], tb[n]), (B, L, n, int((ta[n] != tb[n]).sum()))
        same = lambda x, y: torch.equal(x, y) or bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())      # T' = 1 clips are NaN in both (the reference's unbiased std)
        assert same(ea, eb) and same(la, lb), (B, L)
    # two batches in flight (one workgroup per CU for the persistent grids: persist_cap)
    w = [0.1 * torch.randn(256, 32000, device="cuda", generator=g) for _ in range(3)]
    want = [plain(x, is_eval=True)[1].clone() for x in w]
    monkeypatch.setenv("SIDEKIT_AMD_PAIR", "1")
    t = [paired.submit(w[0]), paired.submit(w[1])]
    got = [paired.collect(t[0])[1]]
    t.append(paired.submit(w[2]))
    got += [paired.collect(t[1])[1], paired.collect(t[2])[1]]
    torch.cuda.synchronize()
    monkeypatch.delenv("SIDEKIT_AMD_PAIR")
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_very_short_clips_as_the_first_call(gpu):
    """A batch of 0.05-0.16 s clips (6-16 frames: ONE row tile per layer, T' = 1-2) as a fresh model's first call -- the
    workspace is then sized by that shape alone (the SE-statistics buffers were once under-reserved for it) -- against each clip
    run alone, bit for bit, in both precisions, and against the oracle in fp32.  Clips that pool over a single frame (T' = 1)
    are NaN in the reference too: the global-context std is unbiased (pooling.py:155-158)."""
    m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=99).to(gpu).eval()
    lens = [1290, 800, 1600, 1930, 2100, 2400, 2399, 815, 1500, 2559]
    single_frame = [i for i, n in enumerate(lens) if 1 + n // 160 <= 8]           # 6-8 frames -> T' = 1
    torch.manual_seed(21)
    wav = 0.1 * torch.randn(len(lens), max(lens))
    for dt in ("bf16", "fp32"):
        m.compute_dtype = dt
        _, emb = m(wav.cuda(), is_eval=True, lengths=lens)          # first call of this precision's handle
        for i, n in enumerate(lens):
            _, one = m(wav[i, :n].cuda(), is_eval=True)
            if i in single_frame:
                assert bool(torch.isnan(emb[i]).all()) and bool(torch.isnan(one[0]).all()), (dt, i, n)
            else:
                assert bool(torch.isfinite(emb[i]).all()) and torch.equal(one[0], emb[i]), (dt, i, n)
    with torch.no_grad():
        _, ref = oxv.forward_ragged([wav[i, :n] for i, n in enumerate(lens)], m.state_dict(), arch="halfresnet34")
    for i, n in enumerate(lens):
        if i in single_frame:
            assert bool(torch.isnan(ref[i]).all()), (i, n)
        else:
            assert rel(emb[i], ref[i]) < TOL, (i, n)
