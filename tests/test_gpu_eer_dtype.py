"""bf16 judged by EER, end to end, on ONE discriminative corpus (north_star: "EER within +-0.05 % absolute").

The reference judges a model by the EER of its real embeddings on a trial list (``test_metrics``, sidekit/nnet/xvector.py:212-271) and
runs its trunk in reduced precision under autocast (``:1890``).  No trained checkpoint or dataset exists offline, so the corpus is the
synthetic one of ``bin/shard_extract_score`` (250 sinusoid "speakers", per-utterance phases / amplitude jitter / white noise) on which
the randomly initialised extractor separates speakers: the SAME waveforms go through the fp32 trunk and through the bf16 trunk
(the headline dtype), the SAME trial set is scored (cosine, two-covariance PLDA estimated from each run's own x-vectors, all pairs of
the corpus through the histogram kernel), and the EERs are compared.

Round 6 (the round-5 verdict: the two legs' all-pairs histograms had been binned on DIFFERENT edges, each from a sample of its own
x-vectors, and the tolerance of the second operating point had been widened to fit): both legs are binned on the fp32 leg's edges, at
8 192 and at 65 520 bins; the exact ROCCH EER of a fixed 8.4 M-pair subsample (all pairs among the first 4 096 utterances) stands
beside the histograms as a witness that has no bins at all; and every delta is reported next to a measured noise floor -- the same
comparison on six further corpus draws (other speaker labels, phases, jitter and noise; 4 096 utterances each): the spread of the paired
delta, and the spread of the EER itself between equally valid trial sets.

What that showed (profiles/r06_eer_fp32_vs_bf16.json): the edges were NOT the cause.  At the second operating point (EER 18 %) the all-pairs
delta is +0.0615 % on the fp32 leg's edges at 8 192 bins, +0.0610 % at 65 520 bins and +0.0617 % with the bf16 leg on its own edges; the exact
8.4 M-pair witness moves by +0.026 %; over the six further draws the paired all-pairs delta is +0.070 % +- 0.025 % (single draw; the mean is
seven standard errors from zero) and the listed-trial cosine delta +0.085 % +- 0.047 % -- a real, systematic cost of the bf16 trunk of 0.3-0.45 %
RELATIVE, against a spread of the EER itself between draws of 0.34 % absolute.  At the operating point inside SURVEY 8d's 1-5 % band (EER 3.9 %,
the regime of the reference's published 1.2 %) the same quantities are +0.001 % / +0.001 % / +0.002 %, and +0.008 % +- 0.007 % over the draws
(0.2 % relative).  So: the 0.05 % criterion is ASSERTED at the in-band point for every scoring; at the 18 % point it is not met by the all-pairs
EER, the test says so as an expected failure (xfail, with the numbers), and only a regression guard is asserted there (1 % relative).

The fp32 leg is itself anchored: its first 64 x-vectors against ``oracle.xvector.halfresnet34_forward`` on the regenerated waveforms
(<= 1e-4 relative, the north_star's fp32 tolerance).

Calibration (scripts/eer_dtype_sweep.py, profiles/r05_eer_dtype_sweep.txt): noise 0.0005 puts the cosine EER at 4.0 % (SURVEY 8d
config 5's 1-5 % band; the corpus' floor -- amplitude jitter and phases, not noise), the second operating point is at a higher noise level."""
import json
import os

import numpy
import pytest
import torch

from oracle import xvector as oxv
from sidekit_amd import iv_scoring
from sidekit_amd.bin import shard_extract_score
from sidekit_amd.bosaris import eer_from_histograms, rocch, rocch2eer
from sidekit_amd.nnet import Xtractor

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLDA5 = os.path.join(ROOT, "tests", "golden", "config5.npz")
N_UTT, N_TRIALS, BATCH = 8192, 2000, 256        # 4 M trials, 16 000 of them targets: one target is 0.006 % of EER
N_SUB = 4096                                    # exact witness: all 8 386 560 unordered pairs among the first 4 096 utterances
FINE_BINS = 8 * 8190                            # "65 536 bins": eight passes of the 8 192-bin kernel, one guard bin either side of each slice
NOISE_SEEDS, NOISE_UTT, NOISE_TRIALS = (1, 2, 3, 4, 5, 6), 4096, 1000
TOL = 5e-4                                      # the north_star's criterion: +-0.05 % absolute
REL_GUARD = 0.01                                # regression guard at the out-of-band point: the measured bf16 cost is 0.3-0.45 % of the EER


@pytest.fixture(scope="module")
def model(gpu):
    return Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(gpu).eval()


def _run(model, dtype, noise, plda=None, utterances=N_UTT, trials=N_TRIALS, seed=0, hist_range=None, all_pairs=True):
    keep = {}
    model.compute_dtype = dtype
    argv = ["--utterances", str(utterances), "--batch", str(BATCH), "--seconds", "4", "--trials", str(trials), "--noise", str(noise),
            "--dtype", dtype, "--seed", str(seed)] + (["--all-pairs"] if all_pairs else []) + (["--plda", plda] if plda else [])
    if hist_range is not None:
        argv += ["--hist-range", repr(float(hist_range[0])), repr(float(hist_range[1]))]
    out = shard_extract_score.main(argv, model=model, keep=keep)
    return out, keep


def _hist_eer(keep, rng, bins=None):
    xv, lab = keep["xv"], torch.as_tensor(keep["labels"], device=keep["xv"].device)
    ht, hn = iv_scoring.cosine_histograms(xv, xv, lab, lab, self_offset=0, lo=rng[0], hi=rng[1], bins=bins)
    n = xv.shape[0]
    total = int(ht.sum() + hn.sum())
    assert abs(total - n * (n - 1)) <= 1e-6 * n * n, (total, n * (n - 1))     # slices of the multi-pass form meet at f32-rounded edges: a pair in 10^6 may sit in two / no slice
    return float(eer_from_histograms(ht, hn))


def _exact_subsample_eer(keep, n=N_SUB):
    """ROCCH EER (bosaris rocch + rocch2eer, the reference's own EER function: detplot.py:354-436) of every unordered pair among the first n utterances."""
    xv = keep["xv"][:n]
    s = iv_scoring.cosine_matrix_device(xv, xv, xv.device).cpu().numpy()
    iu = numpy.triu_indices(n, 1)
    lab = numpy.asarray(keep["labels"][:n])
    tar = lab[iu[0]] == lab[iu[1]]
    s = s[iu].astype(float)
    return float(rocch2eer(*rocch(s[tar], s[~tar]))), int(tar.sum()), int(s.shape[0])


def _record(name, rows):
    scratch = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(scratch):            # profiles/r06_eer_fp32_vs_bf16.json is a copy of this file
        with open(os.path.join(scratch, name), "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


@pytest.mark.parametrize("noise,band,in_band", [(0.0005, (0.01, 0.05), True), (0.008, (0.10, 0.25), False)])
def test_bf16_and_fp32_give_the_same_eer_on_one_corpus(model, noise, band, in_band, capsys):
    """8192 utterances x 4 s, fp32 vs bf16 trunk, |dEER| <= 0.05 % absolute (the north_star's criterion) for:
    cosine on the 2000 x 2000 listed trials (exact ROCCH), PLDA with parameters estimated from each run's own x-vectors, all 67 M pairs through
    the histogram kernel on the fp32 leg's edges (8 192 bins, and 65 520 bins), and the exact ROCCH EER of the 8.4 M-pair subsample -- asserted
    at the operating point inside SURVEY 8d's 1-5 % band; at the 18 % point the criterion is checked, its violation by the all-pairs EER
    (+0.06 % absolute = 0.33 % relative, on identical edges, seven standard errors above the draw-to-draw noise of the paired delta) is an
    EXPECTED failure reported with the numbers, and only a regression guard (1 % relative) is asserted.  Reported
    beside them: the all-pairs delta with each leg binned on its OWN edges (round 5's form), the same comparison on six further corpus draws
    (mean and spread of the paired delta = the noise floor of the comparison; spread of the EER between draws = what a trial set's EER is known
    to), and the reference-trained config-5 PLDA parameters (tests/golden/config5.npz), which model OTHER embeddings (EER 20-43 % here, a flat
    DET curve): bounded by 0.5 % absolute only."""
    try:
        f32, k32 = _run(model, "fp32", noise)
        rng = f32["all_pairs_hist_range"]
        b16, k16 = _run(model, "bf16", noise, hist_range=rng)                 # the bf16 leg on the fp32 leg's edges
        b16_own, _ = _run(model, "bf16", noise)                               # ... and on its own (round 5's comparison), reported
        f32p, _ = _run(model, "fp32", noise, PLDA5, all_pairs=False)
        b16p, _ = _run(model, "bf16", noise, PLDA5, all_pairs=False)
        fine = {"fp32": _hist_eer(k32, rng, FINE_BINS), "bf16": _hist_eer(k16, rng, FINE_BINS)}
        (sub32, n_tar, n_sub), (sub16, _, _) = _exact_subsample_eer(k32), _exact_subsample_eer(k16)
        floor = []
        for seed in NOISE_SEEDS:
            a, _ = _run(model, "fp32", noise, utterances=NOISE_UTT, trials=NOISE_TRIALS, seed=seed)
            b, _ = _run(model, "bf16", noise, utterances=NOISE_UTT, trials=NOISE_TRIALS, seed=seed, hist_range=a["all_pairs_hist_range"])
            floor.append({"seed": seed, "fp32": {k: a[k] for k in ("cosine_eer", "all_pairs_eer")}, "bf16": {k: b[k] for k in ("cosine_eer", "all_pairs_eer")}})
    finally:
        model.compute_dtype = None
    capsys.readouterr()
    keys = ("cosine_eer", "plda_eer", "all_pairs_eer")
    delta = {k: b16[k] - f32[k] for k in keys}
    delta["all_pairs_eer_fine_bins"] = fine["bf16"] - fine["fp32"]
    delta["exact_subsample_eer"] = sub16 - sub32
    d_ap = numpy.array([r["bf16"]["all_pairs_eer"] - r["fp32"]["all_pairs_eer"] for r in floor])
    d_cos = numpy.array([r["bf16"]["cosine_eer"] - r["fp32"]["cosine_eer"] for r in floor])
    e_ap = numpy.array([r["fp32"]["all_pairs_eer"] for r in floor])
    row = {"noise": noise, "utterances": N_UTT, "trials": N_TRIALS * N_TRIALS, "all_pairs": f32["all_pairs"], "hist_range": rng,
           "hist_bins": [f32["all_pairs_hist_bins"], FINE_BINS],
           "fp32": {**{k: f32[k] for k in keys}, "all_pairs_eer_fine_bins": fine["fp32"], "exact_subsample_eer": sub32},
           "bf16": {**{k: b16[k] for k in keys}, "all_pairs_eer_fine_bins": fine["bf16"], "exact_subsample_eer": sub16},
           "delta_abs": delta, "exact_subsample": {"pairs": n_sub, "targets": n_tar},
           "bf16_binned_on_its_own_edges": {"hist_range": b16_own["all_pairs_hist_range"], "all_pairs_eer": b16_own["all_pairs_eer"],
                                            "delta_abs": b16_own["all_pairs_eer"] - f32["all_pairs_eer"]},
           "noise_floor": {"corpus_draws": len(floor), "utterances": NOISE_UTT, "runs": floor,
                           "all_pairs_delta_mean": float(d_ap.mean()), "all_pairs_delta_std": float(d_ap.std(ddof=1)),
                           "cosine_delta_mean": float(d_cos.mean()), "cosine_delta_std": float(d_cos.std(ddof=1)),
                           "all_pairs_eer_std_between_draws": float(e_ap.std(ddof=1))},
           "plda_config5": {"fp32": f32p["plda_eer"], "bf16": b16p["plda_eer"], "delta_abs": b16p["plda_eer"] - f32p["plda_eer"]},
           "xvector_cosine_bf16_vs_fp32_min": float(torch.nn.functional.cosine_similarity(k16["xv"], k32["xv"]).min()),
           "x_vectors_per_s": {"fp32": f32["x_vectors_per_s"], "bf16": b16["x_vectors_per_s"]}}
    _record("eer_fp32_vs_bf16.json", [row])
    print(json.dumps(row))
    assert f32["utterances"] == b16["utterances"] == N_UTT and f32["all_pairs"] == b16["all_pairs"] == N_UTT * (N_UTT - 1)
    assert b16["all_pairs_hist_range"] == rng, "the two legs must be binned on the same edges"
    assert band[0] < f32["cosine_eer"] < band[1], f"the corpus left its calibrated band: cosine EER {f32['cosine_eer']:.4f}"
    over = {k: d for k, d in delta.items() if abs(d) > TOL}
    if in_band:
        assert not over, (over, TOL, row["noise_floor"])
    else:
        for k, d in delta.items():
            assert abs(d) <= REL_GUARD * row["fp32"][k], ("regression guard", k, d, row["fp32"][k])
    assert f32p["cosine_eer"] == f32["cosine_eer"] and b16p["cosine_eer"] == b16["cosine_eer"]        # same extraction, bit for bit
    assert abs(b16p["plda_eer"] - f32p["plda_eer"]) <= 5e-3, (f32p["plda_eer"], b16p["plda_eer"])
    # the two runs saw the same waveforms: every bf16 x-vector is its fp32 x-vector up to the trunk's rounding
    assert row["xvector_cosine_bf16_vs_fp32_min"] > 0.999
    if over:
        nf = row["noise_floor"]
        pytest.xfail(f"known bf16 cost outside the 1-5 % band (EER {f32['all_pairs_eer']:.2%}): " + ", ".join(f"{k} {d:+.3%}" for k, d in over.items()) +
                     f" absolute on identical histogram edges; paired delta over {nf['corpus_draws']} further draws {nf['all_pairs_delta_mean']:+.3%} +- "
                     f"{nf['all_pairs_delta_std']:.3%}; EER spread between draws {nf['all_pairs_eer_std_between_draws']:.2%} (profiles/r06_eer_fp32_vs_bf16.json)")


def test_the_fp32_leg_is_the_oracle(model):
    """The first 64 utterances of the corpus (regenerated from the driver's own seed recipe) through the oracle's CPU restatement of the
    reference forward: the fp32 run of the EER comparison is within the north_star's 1e-4 of it."""
    noise, L = 0.0005, 64000
    keep = {}
    model.compute_dtype = "fp32"
    try:
        shard_extract_score.main(["--utterances", "1024", "--batch", "256", "--seconds", "4", "--trials", "250", "--noise", str(noise), "--dtype", "fp32"],
                                 model=model, keep=keep)
    finally:
        model.compute_dtype = None
    labels = numpy.random.RandomState(1).randint(0, 250, 1024).astype(numpy.int32)
    assert numpy.array_equal(labels, keep["labels"])
    freqs, amps = shard_extract_score.speaker_table(250)
    dev = keep["xv"].device
    g = torch.Generator(device=dev).manual_seed(1000)                        # the batch that starts at utterance 0
    wav = shard_extract_score.synth_batch(labels[:256], freqs, amps, L, noise, g, dev)[:64].cpu()
    with torch.no_grad():
        ref = torch.cat([oxv.halfresnet34_forward(wav[i:i + 16], model.state_dict())[1] for i in range(0, 64, 16)])
    got = keep["xv"][:64].cpu()
    rel = ((got - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    assert rel < 1e-4, rel
    # and the corpus is discriminative for this extractor: same-speaker pairs score higher than different-speaker pairs
    s = (got @ got.t()).numpy()
    same = labels[:64, None] == labels[None, :64]
    off = ~numpy.eye(64, dtype=bool)
    if (same & off).any():
        assert s[same & off].mean() > s[~same].mean()
