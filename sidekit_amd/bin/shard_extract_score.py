"""Sharded extraction + scoring on the GPUs of one node (BASELINE.json configs 3 and 5).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m sidekit_amd.bin.shard_extract_score \\
        --utterances 100000 --batch 256 --seconds 4

Every rank (one process per GPU) extracts the contiguous utterance range ``shard_range(n, rank, world)`` of a synthetic
corpus, the ``(N_r, 256)`` x-vector blocks are all-gathered over RCCL (the only exchange step of the path), and the trial
set OF THOSE x-vectors is scored: the first ``--trials`` utterances are the enrolment side, the next ``--trials`` the test side
(full trial mask, target = same synthetic speaker), the remainder trains the PLDA parameters.  Scoring is sharded by enrolment
rows and stays on the device end to end: cosine (f32 MFMA) and fast PLDA (f64 MFMA) row blocks are gathered on rank 0, which
prints EERs and timings as one JSON line.  ``--all-pairs`` adds the matrix-free path: every pair of the corpus scored into
target / non-target histograms, the per-rank counts summed with one all-reduce.  Without ``torch.distributed.run`` it runs
as a single rank.

The corpus: speaker s is a fixed set of sinusoids (``RandomState(0)``), an utterance adds per-utterance phases, amplitude
jitter and white noise (a batch is generated on the device with seed ``1000 + index of its first utterance``) -- enough
structure for the randomly initialised extractor to separate speakers with a non-trivial EER.  The PLDA parameters are inputs
to the scoring path: ``--plda FILE`` reads ``(mu, F, Sigma)`` from a SIDEKIT PLDA HDF5 file (``sidekit_io.read_plda_hdf5``) or from
an ``.npz`` with those three arrays (``tests/golden/config5.npz`` holds the ones the reference's ``FactorAnalyser.plda`` trained);
without it a moment estimate (between / within speaker covariance of the training x-vectors) stands in, since PLDA *training*
proper (``sidekit/factor_analyser.py:830-932``) is out of scope.

``main(argv, model=None, scoring=None, keep=None)``: ``keep`` (a dict) receives the gathered x-vectors (``"xv"``, device tensor), the labels and
rank 0's two score matrices -- for tests that compare two runs; the model and the module that scores (``cosine_matrix_device``, ``plda_matrix_device``,
``cosine_histograms``) default to the GPU ones; ``--backend gloo --device cpu`` with injected stand-ins runs this driver's real
control flow (ragged gather, row shards, ``self_offset``, the counter all-reduce) on CPU ranks (``tests/test_sharding_cpu.py``).
"""
import argparse
import json
import os
import time

import numpy
import torch
import torch.distributed as dist

from .. import iv_scoring
from ..bosaris import eer_from_histograms, rocch, rocch2eer
from ..nnet import Xtractor
from ..sharding import gather_xvectors, score_sharded, shard_range

N_TONES = 6


def speaker_table(n_spk):
    rs = numpy.random.RandomState(0)
    return rs.uniform(120.0, 7000.0, (n_spk, N_TONES)), rs.uniform(0.3, 1.0, (n_spk, N_TONES))


def synth_batch(spk, freqs, amps, L, noise, gen, dev):
    """(B,) speaker ids -> (B, L) float32 waveforms on `dev`."""
    B = spk.shape[0]
    t = torch.arange(L, device=dev, dtype=torch.float32) / 16000.0
    f = torch.as_tensor(freqs[spk], dtype=torch.float32, device=dev)               # (B, tones)
    a = torch.as_tensor(amps[spk], dtype=torch.float32, device=dev) * (0.7 + 0.6 * torch.rand(B, N_TONES, device=dev, generator=gen))
    ph = 6.2831853 * torch.rand(B, N_TONES, device=dev, generator=gen)
    x = torch.zeros(B, L, device=dev)
    for k in range(N_TONES):
        x += a[:, k:k + 1] * torch.sin(6.2831853 * f[:, k:k + 1] * t[None, :] + ph[:, k:k + 1])
    return 0.05 * x + noise * torch.randn(B, L, device=dev, generator=gen)


def plda_moments(X, labels, rank):
    """Two-covariance moment estimate -> (mu, F, Sigma) as ``fast_PLDA_scoring`` takes them (float64 numpy)."""
    X = numpy.asarray(X, dtype=numpy.float64)
    mu = X.mean(axis=0)
    ids, inv = numpy.unique(labels, return_inverse=True)
    means = numpy.stack([X[inv == i].mean(axis=0) for i in range(ids.shape[0])])
    W = numpy.cov((X - means[inv]).T, bias=True)
    Bc = numpy.cov((means - mu).T, bias=True)
    w, V = numpy.linalg.eigh(Bc)
    top = numpy.argsort(w)[::-1][:rank]
    F = V[:, top] * numpy.sqrt(numpy.maximum(w[top], 1e-12))
    Sigma = W + 1e-4 * numpy.trace(W) / W.shape[0] * numpy.eye(W.shape[0])
    return mu, F, Sigma


def hist_range_from_sample(scoring, xv, dev):
    """[lo, hi) of the all-pairs histograms from a strided sample of 2048 x-vectors: the smallest sampled score, widened by a quarter of the
    sampled range (never below -1), up to 1."""
    N = xv.shape[0]
    sample = xv[:: max(1, N // 2048)][:2048]
    smin = float(scoring.cosine_matrix_device(sample, sample, dev).min())
    return max(-1.0, smin - 0.25 * (1.0 - smin)), 1.0 + 1e-6


def load_plda(path):
    """(mu, F, Sigma) float64 from a SIDEKIT PLDA HDF5 file (sidekit_io.py:282-324) or an .npz holding those three arrays."""
    if path.endswith(".npz"):
        z = numpy.load(path)
        return tuple(numpy.asarray(z[k], dtype=numpy.float64) for k in ("mu", "F", "Sigma"))
    from ..sidekit_io import read_plda_hdf5
    mu, F, _G, Sigma = read_plda_hdf5(path)
    return numpy.asarray(mu, dtype=numpy.float64), numpy.asarray(F, dtype=numpy.float64), numpy.asarray(Sigma, dtype=numpy.float64)


def main(argv=None, model=None, scoring=None, keep=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--trials", type=int, default=1000, help="enrolment models = test segments = this many (full trial mask)")
    ap.add_argument("--speakers", type=int, default=250)
    ap.add_argument("--noise", type=float, default=0.004, help="white-noise level of the synthetic utterances (sets the EER: 0.004 -> cosine ~15 %, PLDA ~10 %; 0.03 -> 46 %)")
    ap.add_argument("--plda-rank", type=int, default=128)
    ap.add_argument("--all-pairs", action="store_true", help="also score every pair of the corpus into histograms (no N x N matrix)")
    ap.add_argument("--hist-range", type=float, nargs=2, default=None, metavar=("LO", "HI"),
                    help="score range of the all-pairs histograms; default: derived from a sample of THIS run's x-vectors.  Two runs that are to be "
                         "compared (fp32 against bf16, tests/test_gpu_eer_dtype.py) must be binned on the same edges: pass the first run's range to the second")
    ap.add_argument("--hist-bins", type=int, default=None, help="bins of the all-pairs histograms: 8192 (default, one pass) or a multiple of 8190 (that many passes / 8190)")
    ap.add_argument("--seed", type=int, default=0, help="corpus seed: another draw of speaker labels, phases, amplitude jitter and noise for the same speaker table")
    ap.add_argument("--plda", default=None, help="PLDA (mu, F, Sigma): SIDEKIT HDF5 or .npz; default: moment estimate from the corpus")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI; gloo for CPU rehearsals")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu only with an injected model / scoring module")
    args = ap.parse_args(argv)
    scoring = iv_scoring if scoring is None else scoring
    assert args.utterances >= 2 * args.trials + 2 * args.speakers, "need utterances for enrolment, test and PLDA training"
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if args.device == "cuda":
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        if model is None or scoring is iv_scoring:
            raise RuntimeError("--device cpu needs an injected model and scoring module: the product path has no CPU fallback")
        dev = torch.device("cpu")
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
    own_group = "RANK" in os.environ and not dist.is_initialized()
    if own_group:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    if dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    if model is None:
        model = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
        model.compute_dtype = args.dtype
    L = int(args.seconds * 16000)
    N = args.utterances
    labels = numpy.random.RandomState(1 + args.seed).randint(0, args.speakers, N).astype(numpy.int32)     # speaker of every utterance
    freqs, amps = speaker_table(args.speakers)
    start, stop = shard_range(N, rank, world)
    n_batches = (stop - start + args.batch - 1) // args.batch
    sync()
    t0 = time.perf_counter()
    blocks, tickets = [], []
    pipelined = dev.type == "cuda" and hasattr(model, "submit")     # whole batches in flight (Xtractor.submit / collect; model.pipeline_depth)
    for k in range(n_batches):
        lo = start + k * args.batch
        hi = min(lo + args.batch, stop)
        g = torch.Generator(device=dev).manual_seed(1000 + lo + 1000003 * args.seed)   # a batch's seed = its first utterance: independent of the rank count when shards are batch aligned
        wav = synth_batch(labels[lo:hi], freqs, amps, L, args.noise, g, dev)
        if pipelined:
            tickets.append(model.submit(wav))
            if len(tickets) == model.pipeline_depth:
                blocks.append(model.collect(tickets.pop(0))[1])
        else:
            blocks.append(model(wav, is_eval=True)[1])
    while tickets:
        blocks.append(model.collect(tickets.pop(0))[1])
    local_xv = torch.cat(blocks) if blocks else torch.empty(0, model.embedding_size, device=dev)
    sync()
    t_extract = time.perf_counter() - t0
    t0 = time.perf_counter()
    xv = gather_xvectors(local_xv)                      # (utterances, 256) on every rank, still on the device
    sync()
    t_gather = time.perf_counter() - t0
    assert xv.shape == (N, model.embedding_size) and xv.device.type == dev.type
    norms = xv.double().norm(dim=1)
    a0, b0 = shard_range(N, rank, world)
    gathered_own_block_ok = bool(torch.equal(xv[a0:b0], local_xv))     # the gather put this rank's block where the index range says
    # ---- the gathered x-vectors' own trial set: enrol [0, n), test [n, 2n), PLDA training on the rest
    n = args.trials
    E, T, train = xv[:n], xv[n:2 * n], xv[2 * n:]
    tar = labels[:n, None] == labels[None, n:2 * n]
    t0 = time.perf_counter()
    cos_rows = score_sharded(lambda a, b: scoring.cosine_matrix_device(E[a:b], T, dev), n)   # rank 0 gets (n, n), device resident
    sync()
    t_cos = time.perf_counter() - t0
    if args.plda:
        mu, F, Sigma = load_plda(args.plda)
        assert mu.shape[0] == xv.shape[1], "PLDA dimension differs from the x-vectors'"
    else:
        mu, F, Sigma = plda_moments(train.cpu().numpy(), labels[2 * n:], args.plda_rank)  # identical on every rank (same gathered data)
    Phi, Psi, cst = iv_scoring.plda_parameters(mu, F, Sigma)
    mu_d = torch.as_tensor(mu, device=dev)
    Ec, Tc = E.double() - mu_d, T.double() - mu_d                                        # center_stat1 (statserver.py:810-817)
    t0 = time.perf_counter()
    plda_rows = score_sharded(lambda a, b: scoring.plda_matrix_device(Ec[a:b], Tc, Phi, Psi, cst, 1.0, dev), n)
    sync()
    t_plda = time.perf_counter() - t0
    out = {"ranks": world, "utterances": N, "x_vectors_per_s": N / t_extract, "extract_s": t_extract, "all_gather_s": t_gather,
           "trials": n * n, "cosine_score_s": t_cos, "plda_score_s": t_plda, "dtype": args.dtype,
           "plda": args.plda or "moment estimate", "backend": dist.get_backend() if dist.is_initialized() else None,
           "xv_finite": bool(torch.isfinite(xv).all()), "xv_norm_max_dev": float((norms - 1.0).abs().max()),
           "gathered_own_block_ok": gathered_own_block_ok}
    if args.all_pairs:
        # matrix-free: rank r counts the pairs (i, j), i in its enrolment-row shard, j over the whole corpus, i != j
        a, b = shard_range(N, rank, world)
        lab_d = torch.as_tensor(labels, device=dev)
        t0 = time.perf_counter()
        # histogram range from a strided sample of the gathered x-vectors (the same on every rank, no collective): 8192 bins over
        # [-1, 1) are 2.4e-4 wide, and an extractor whose x-vectors share a common direction (every score in [0.98, 1]) would land in a few
        # dozen of them -- the binned EER then carries the bin width, not the scores.  The bins go where the scores are (widened by a quarter
        # of the sampled range; whatever falls outside is counted in the end bins by the kernel).  The range depends on the run's own
        # x-vectors: two runs whose EERs are to be COMPARED must share it (--hist-range; round 5 compared fp32 and bf16 on different edges).
        lo, hi = args.hist_range if args.hist_range else hist_range_from_sample(scoring, xv, dev)
        ht, hn = scoring.cosine_histograms(xv[a:b], xv, lab_d[a:b], lab_d, self_offset=a, lo=lo, hi=hi, device=dev,
                                           **({"bins": args.hist_bins} if args.hist_bins else {}))
        out["all_pairs_hist_range"] = [lo, hi]
        out["all_pairs_hist_bins"] = int(ht.shape[0])
        sync()
        counts = torch.as_tensor(numpy.stack([ht, hn]).astype(numpy.int64), device=dev)
        if dist.is_initialized():
            dist.all_reduce(counts)
        t_hist = time.perf_counter() - t0
        counts = counts.cpu().numpy()
        out.update(all_pairs=int(counts.sum()), all_pairs_s=t_hist, all_pairs_eer=float(eer_from_histograms(counts[0], counts[1])))
    if keep is not None:
        keep.update(xv=xv, labels=labels, tar=tar)
    if rank == 0:
        for name, rows in (("cosine", cos_rows), ("plda", plda_rows)):
            s = rows.cpu().numpy().astype(float)
            out[f"{name}_eer"] = float(rocch2eer(*rocch(s[tar], s[~tar])))
            if keep is not None:
                keep[f"{name}_scores"] = s
        print(json.dumps(out), flush=True)
    if own_group:
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
