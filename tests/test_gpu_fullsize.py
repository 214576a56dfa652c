"""BASELINE.json's full sizes, checked through size-independent properties (the oracle would take minutes)."""
import numpy
import pytest
import torch

from oracle import xvector as oxv
from sidekit_amd.bosaris import rocch, rocch2eer
from sidekit_amd.nnet import Xtractor

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def model(gpu):
    return Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(gpu).eval()


def test_config2_batch256_4s(model):
    """HalfResNet34, batch 256, 4 s: unit norm, batch permutation equivariance, gain invariance, spot-check vs oracle."""
    g = torch.Generator(device="cuda").manual_seed(0)
    wav = 0.1 * torch.randn(256, 64000, device="cuda", generator=g)
    for dtype in ("fp32", "bf16"):
        model.compute_dtype = dtype
        logits, emb = model(wav, is_eval=True)
        assert emb.shape == (256, 256) and logits.shape == (256, 7205)
        assert torch.allclose(emb.norm(dim=1), torch.ones(256, device="cuda"), atol=1e-5)
        assert float(logits.abs().max()) <= 30.0 + 1e-3                        # s * cosine
        perm = torch.randperm(256, device="cuda", generator=g)
        _, emb_p = model(wav[perm], is_eval=True)
        assert torch.equal(emb_p, emb[perm])                                   # utterances are independent
        _, emb_g = model(0.25 * wav[:8], is_eval=True)                         # N4: gain invariance through CMVN
        assert float(torch.nn.functional.cosine_similarity(emb_g, emb[:8]).min()) > (0.9999 if dtype == "fp32" else 0.995)
    model.compute_dtype = "fp32"
    _, emb = model(wav, is_eval=True)
    with torch.no_grad():
        _, ref = oxv.halfresnet34_forward(wav[[3, 200]].cpu(), model.state_dict())
    assert rel(emb[[3, 200]], ref) < 1e-4


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_two_lane_forward_is_bit_identical_to_the_serial_one(model, dtype):
    """xt_set_lanes: a batch of >= 128 utterances forwarded as two halves on two HIP streams (the product default) against the
    same batch on one stream -- x-vectors and logits are the same bits, for uniform and for ragged (odd-sized) batches, and a
    batch below the threshold is not split at all."""
    g = torch.Generator(device="cuda").manual_seed(5)
    model.compute_dtype = dtype
    try:
        for B, L, ragged in ((256, 64000, False), (131, 40000, True), (64, 30000, False)):
            wav = 0.1 * torch.randn(B, L, device="cuda", generator=g)
            lens = torch.randint(L // 3, L + 1, (B,), generator=torch.Generator().manual_seed(B)).tolist() if ragged else None
            model.set_lanes(2)
            assert model.get_lanes() == 2
            two = [model(wav, is_eval=True, lengths=lens) for _ in range(8 if dtype == "bf16" else 3)]   # steady state: both lanes reuse their workspaces
            model.set_lanes(4)                                              # four parts of >= 64 utterances where the batch allows
            two += [model(wav, is_eval=True, lengths=lens) for _ in range(4 if dtype == "bf16" else 2)]
            model.set_lanes(1)
            lg1, e1 = model(wav, is_eval=True, lengths=lens)
            torch.cuda.synchronize()
            # every repetition: a kernel that misbehaves beside another lane's kernels does so intermittently (the STFT kernel did,
            # for a few frames per batch, until both front-ends ran ahead of the fork: DESIGN.md section 6)
            for lg2, e2 in two:
                assert torch.equal(e2, e1) and torch.equal(lg2, lg1), (dtype, B, torch.nonzero((e2 != e1).any(dim=1)).flatten().tolist()[:8])
            assert bool(torch.isfinite(e1).all())
    finally:
        model.set_lanes(2)
        model.compute_dtype = None


def test_front_ends_are_stable_beside_another_streams_bf16_trunk(gpu):
    """Regression for the round-3 hazard: with SLP-formed packed-f32 instructions (v_pk_add_f32 / v_pk_mul_f32 with op_sel / neg
    modifiers) the STFT kernels gave a few wrong spectrum bins per batch whenever ANOTHER stream was issuing dense bf16 MFMAs
    (5-15 % of the utterances of a 128-batch per run: profiles/r03_two_lane_frontend_hazard.txt).  The library is built without
    them (csrc/Makefile); here both front-ends (log-mel, MFCC) run on a side stream beside a second model's bf16 forward and
    must reproduce their solo features bit for bit, every time."""
    m1 = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=1).to(gpu).eval()
    m2 = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=2).to(gpu).eval()
    m3 = Xtractor(64, model_archi="xvector", loss="aam", seed=3).to(gpu).eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    a = 0.1 * torch.randn(128, 64000, device="cuda", generator=g)
    b = 0.1 * torch.randn(128, 64000, device="cuda", generator=g)
    ref2, ref3 = m2.features(b), m3.features(b)
    m1.compute_dtype = "bf16"
    m1.set_lanes(1)
    m1(a, is_eval=True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for trial in range(12):
        m1(a, is_eval=True)                      # main stream: layer-1 .. layer-4 bf16 MFMA kernels
        with torch.cuda.stream(side):
            f2, f3 = m2.features(b), m3.features(b)
        torch.cuda.synchronize()
        assert torch.equal(f2, ref2), (trial, int((f2 != ref2).any(dim=(1, 2)).sum()))
        assert torch.equal(f3, ref3), (trial, int((f3 != ref3).any(dim=(1, 2)).sum()))


def test_small_kernels_are_stable_beside_another_streams_bf16_trunk(gpu):
    """The same regression for the two kernels that still carry packed-f32 FMAs (tests/test_isa_guard.py lists the forms): the stem
    (explicit two-element FMAs) and the SE gate (SLP-formed `v_pk_fma_f32`, built with SLP for speed).  A second model's whole forward
    -- stem, sixteen SE gates, pooling, tail -- runs on a side stream beside the first model's bf16 trunk, several times per trial so
    that it overlaps every layer, in both compute dtypes, and must reproduce its solo x-vectors and logits bit for bit."""
    m1 = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=1).to(gpu).eval()
    m2 = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=2).to(gpu).eval()
    g = torch.Generator(device="cuda").manual_seed(2)
    a = 0.1 * torch.randn(256, 64000, device="cuda", generator=g)
    b = 0.1 * torch.randn(48, 24000, device="cuda", generator=g)
    m1.compute_dtype = "bf16"
    m1.set_lanes(1)
    m1(a, is_eval=True)
    side = torch.cuda.Stream()
    for dtype in ("bf16", "fp32"):
        m2.compute_dtype = dtype
        ref_logits, ref_emb = m2(b, is_eval=True)
        torch.cuda.synchronize()
        for trial in range(10):
            m1(a, is_eval=True)                  # main stream: 5.8 ms of bf16 MFMA kernels
            with torch.cuda.stream(side):
                outs = [m2(b, is_eval=True) for _ in range(4 if dtype == "bf16" else 2)]
            torch.cuda.synchronize()
            for logits, emb in outs:
                assert torch.equal(emb, ref_emb), (dtype, trial, int((emb != ref_emb).any(dim=1).sum()))
                assert torch.equal(logits, ref_logits), (dtype, trial)


def test_partial_batches_fit_the_side_lanes(gpu):
    """with three or four lanes a batch SMALLER than the reserved one is split into fewer, larger parts (lanes = 4, reserve 256:
    B = 255 -> three parts of 85; lanes = 3, B = 191 -> two parts of 96 / 95) and the side lanes used to be sized for ceil(256 / n) only:
    the STFT kernel wrote past the lane's feature workspace before the call failed.  On a FRESH handle per lane count (no earlier
    set_lanes(2) to grow lane 1) such batches must run and give the serial forward's bits."""
    g = torch.Generator(device="cuda").manual_seed(9)
    wav = 0.1 * torch.randn(256, 24000, device="cuda", generator=g)
    serial = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=6).to(gpu).eval()
    serial.compute_dtype = "bf16"
    serial.set_lanes(1)
    refs = {B: serial(wav[:B], is_eval=True)[1].clone() for B in (256, 255, 191, 130, 127)}
    for lanes in (4, 3):
        m = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=6).to(gpu).eval()
        m.compute_dtype = "bf16"
        m.set_lanes(lanes)                          # before the first forward: the handle is created with this lane count
        assert m.get_lanes() == lanes
        for B in (256, 255, 191, 130, 127):
            _, emb = m(wav[:B], is_eval=True)
            assert torch.equal(emb, refs[B]), (lanes, B, int((emb != refs[B]).any(dim=1).sum()))
        torch.cuda.synchronize()


def test_config4_tdnn_512_variable_length(gpu):
    """TDNN fp32, 512 utterances of 2-10 s (RandomState(0) lengths): finite unit-norm output, spot parity vs the oracle."""
    m = Xtractor(7205, model_archi="xvector", loss="aam", seed=4321).to(gpu).eval()
    lens = numpy.random.RandomState(0).randint(32000, 160001, (512,))
    g = torch.Generator(device="cuda").manual_seed(0)
    wav = 0.1 * torch.randn(512, int(lens.max()), device="cuda", generator=g)
    logits, emb = m(wav, is_eval=True, lengths=lens.tolist())
    assert emb.shape == (512, 256) and bool(torch.isfinite(emb).all())
    assert torch.allclose(emb.norm(dim=1), torch.ones(512, device="cuda"), atol=1e-5)
    idx = [0, 17, 511, int(lens.argmin()), int(lens.argmax())]
    with torch.no_grad():
        _, ref = oxv.forward_ragged([wav[i, :lens[i]].cpu() for i in idx], m.state_dict(), arch="xvector")
    assert rel(emb[idx], ref) < 1e-4


def test_long_utterance_45s(model):
    """Per-speaker concatenations (xvector.py:1919-1999) and sliding-window sources are long single utterances: 45 s =
    4501 frames (563 layer1 row tiles, T' = 563 pooled frames) against the oracle in fp32, and bf16 next to it."""
    g = torch.Generator(device="cuda").manual_seed(3)
    wav = 0.1 * torch.randn(1, 45 * 16000, device="cuda", generator=g)
    model.compute_dtype = "fp32"
    _, e32 = model(wav, is_eval=True)
    with torch.no_grad():
        _, ref = oxv.halfresnet34_forward(wav.cpu(), model.state_dict())
    assert rel(e32, ref) < 1e-4
    model.compute_dtype = "bf16"
    try:
        _, e16 = model(wav, is_eval=True)
    finally:
        model.compute_dtype = "fp32"
    assert float(torch.nn.functional.cosine_similarity(e16, e32).min()) > 0.999


def test_config3_100k_utterances_at_the_stated_size(model, gpu, capsys):
    """BASELINE configs[2] at its stated size on the one GPU a box offers: 100 000 utterances x 4 s through the sharded driver
    (`bin/shard_extract_score`, the loop that replaces sidekit/bin/extract_xvectors.py:130-150), 391 batches of 256 (the last one 160: the
    two-lane split of a partial batch), the 1000 x 1000 trial set of config 5 scored with the reference-trained PLDA parameters of
    tests/golden/config5.npz, and every pair of the corpus scored matrix-free.  The oracle would need half an hour for this, so the checks
    are the size-independent properties: count, finiteness, unit norm, the gathered block sitting where the index range says, three
    EERs in range, all-pairs count = N (N - 1)."""
    import json
    import os
    from sidekit_amd.bin import shard_extract_score
    plda = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5.npz")
    model.compute_dtype = "bf16"
    try:
        out = shard_extract_score.main(["--utterances", "100000", "--batch", "256", "--seconds", "4", "--all-pairs", "--plda", plda], model=model)
    finally:
        model.compute_dtype = "fp32"
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d == json.loads(json.dumps(out))
    N = 100000
    assert d["ranks"] == 1 and d["utterances"] == N and d["trials"] == 1000 * 1000 and d["dtype"] == "bf16"
    assert d["xv_finite"] and d["xv_norm_max_dev"] < 1e-5 and d["gathered_own_block_ok"]
    assert d["all_pairs"] == N * (N - 1)
    for k in ("cosine_eer", "plda_eer", "all_pairs_eer"):
        assert numpy.isfinite(d[k]) and 0.0 <= d[k] < 0.5, (k, d[k])
    assert abs(d["cosine_eer"] - d["all_pairs_eer"]) < 0.05          # the same score distribution over different trial subsets
    assert d["x_vectors_per_s"] > 10000                                # includes the on-device synthesis of the waveforms
    scratch = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(scratch):                                         # the r04 config-3 line under profiles/ is a copy of this file
        with open(os.path.join(scratch, "config3_100k_one_gpu.json"), "w") as f:
            f.write(line + "\n")


def test_error_in_a_side_lane_joins_every_lane(gpu):
    """an error after the fork (here: an utterance of the SECOND part too short for the reflect padding, caught by that part's
    length check once part 0 is already queued) must not leave a forked lane running unjoined -- the call raises, the caller's stream still
    orders behind every lane, and the next forwards on the same handle give the serial forward's bits."""
    m = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=8).to(gpu).eval()
    m.compute_dtype = "bf16"
    g = torch.Generator(device="cuda").manual_seed(4)
    wav = 0.1 * torch.randn(256, 32000, device="cuda", generator=g)
    m.set_lanes(1)
    ref = m(wav, is_eval=True)[1].clone()
    m.set_lanes(2)
    assert torch.equal(m(wav, is_eval=True)[1], ref)
    lens = [32000] * 256
    lens[200] = 300                                   # n_fft / 2 = 512 samples are the minimum: part 1 (rows 128..255) fails, part 0 is queued
    for _ in range(3):
        with pytest.raises(ValueError):
            m(wav, is_eval=True, lengths=lens)
        assert torch.equal(m(wav, is_eval=True)[1], ref)
    torch.cuda.synchronize()
    lens[200], lens[10] = 32000, 300                  # and in part 0: nothing has been forked yet
    with pytest.raises(ValueError):
        m(wav, is_eval=True, lengths=lens)
    assert torch.equal(m(wav, is_eval=True)[1], ref)


def test_normalize_rows_matches_torch(gpu):
    """sc_normalize_rows = torch.nn.functional.normalize(x, dim=1) (sidekit/score_normalization.py:128, nnet/xvector.py:243): unit rows, zero rows
    stay zero (eps), float32 rounding apart."""
    from sidekit_amd.iv_scoring import normalize_rows_device
    x = torch.randn(1000, 256, generator=torch.Generator().manual_seed(0))
    x[17] = 0.0
    got = normalize_rows_device(x).cpu()
    want = torch.nn.functional.normalize(x, dim=1)
    assert got.shape == want.shape and torch.equal(got[17], torch.zeros(256))
    assert float((got - want).abs().max()) < 2e-7
    x2 = torch.randn(33, 100)                        # a dimension that is not a multiple of 4 or 64
    assert float((normalize_rows_device(x2).cpu() - torch.nn.functional.normalize(x2, dim=1)).abs().max()) < 2e-7


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_pipelined_forwards_are_bit_identical(gpu, dtype):
    """Xtractor.submit / collect (xt_forward_begin / xt_forward_end): two whole batches in flight on the handle's two slot streams give the
    x-vectors and logits of the plain forward, bit for bit -- uniform, ragged, PCM16, a partial batch after a full one, interleaved with plain
    forwards on the same handle -- tickets come back in order, and a third submit without a collect is refused."""
    m = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=12).to(gpu).eval()
    m.compute_dtype = dtype
    g = torch.Generator(device="cuda").manual_seed(21)
    batches = []
    for B, L, ragged, pcm in ((256, 32000, False, False), (200, 24000, True, False), (256, 32000, False, True), (37, 48000, True, False), (256, 32000, False, False)):
        wav = 0.1 * torch.randn(B, L, device="cuda", generator=g)
        if pcm:
            wav = (wav * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
        lens = torch.randint(L // 3, L + 1, (B,), generator=torch.Generator().manual_seed(B + L)).tolist() if ragged else None
        batches.append((wav, lens))
    refs = [tuple(t.clone() for t in m(w, is_eval=True, lengths=l)) for w, l in batches]
    for rep in range(3 if dtype == "bf16" else 1):
        tickets, outs = [], []
        for i, (w, l) in enumerate(batches):
            tickets.append(m.submit(w, lengths=l))
            if len(tickets) == 2:
                outs.append(m.collect(tickets.pop(0)))
            if i == 2:                                           # a plain forward between two submits: it shares slot 0's stream order
                lg, e = m(batches[0][0], is_eval=True)
                assert torch.equal(e, refs[0][1])
        while tickets:
            outs.append(m.collect(tickets.pop(0)))
        torch.cuda.synchronize()
        for (lg, e), (rlg, re_) in zip(outs, refs):
            assert torch.equal(e, re_) and torch.equal(lg, rlg)
    t1, t2 = m.submit(batches[0][0]), m.submit(batches[0][0])
    with pytest.raises(RuntimeError):
        m.submit(batches[0][0])                                   # two are in flight
    with pytest.raises(RuntimeError):
        m.collect(t2)                                             # out of order
    assert torch.equal(m.collect(t1)[1], refs[0][1]) and torch.equal(m.collect(t2)[1], refs[0][1])
    bad = [32000] * 256
    bad[5] = 100
    with pytest.raises(ValueError):
        m.submit(batches[0][0], lengths=bad)                      # refused before anything is queued
    assert torch.equal(m.collect(m.submit(batches[0][0]))[1], refs[0][1])


def test_pipelined_forwards_tdnn(gpu):
    """The same for the TDNN x-vector (ragged rows, fp32): submit / collect against the plain forward."""
    m = Xtractor(64, model_archi="xvector", loss="aam", seed=13).to(gpu).eval()
    g = torch.Generator(device="cuda").manual_seed(22)
    wav = 0.1 * torch.randn(96, 64000, device="cuda", generator=g)
    lens = torch.randint(32000, 64001, (96,), generator=torch.Generator().manual_seed(3)).tolist()
    ref_a, ref_b = m(wav, is_eval=True, lengths=lens)[1].clone(), m(wav[:40], is_eval=True)[1].clone()
    for _ in range(2):
        t1 = m.submit(wav, lengths=lens)
        t2 = m.submit(wav[:40])
        assert torch.equal(m.collect(t1)[1], ref_a)
        t3 = m.submit(wav, lengths=lens)
        assert torch.equal(m.collect(t2)[1], ref_b) and torch.equal(m.collect(t3)[1], ref_a)
    torch.cuda.synchronize()


def test_pipelined_and_plain_forwards_interleaved_at_random(gpu):
    """scripts/soak_pipelined.py as a test: random batch shapes (1..300 utterances, 0.3..6 s, ragged / uniform, float32 / PCM16) through submit / collect,
    interleaved at random with plain forwards -- split ones and small unsplit ones, which run on the CALLER's stream in slot 0's workspace (the soak found
    such a forward racing with a pipelined batch still running there; it now orders behind lane 0's stream) -- against a one-at-a-time reference model."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("soak_pipelined", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak_pipelined.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(250, verbose=False) == 0
