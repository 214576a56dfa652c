"""Sharded extraction + scoring on the GPUs of one node (BASELINE.json configs 3 and 5).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m sidekit_amd.bin.shard_extract_score \\
        --utterances 100000 --batch 256 --seconds 4

Every rank (one process per GPU) extracts the contiguous utterance range ``shard_range(n, rank, world)`` from
synthetic waveforms (batch k of rank r is generated on the device with seed ``1000 + r * n_batches + k``), the
``(N_r, 256)`` x-vector blocks are all-gathered over RCCL (the only exchange step of the path), then a synthetic
trial matrix is scored with cosine / fast PLDA sharded by enrolment rows and rank 0 prints the EER and timings as
one JSON line.  Without ``torch.distributed.run`` it runs as a single rank.
"""
import argparse
import json
import os
import time

import numpy
import torch
import torch.distributed as dist

from .. import iv_scoring
from ..bosaris import rocch, rocch2eer
from ..nnet import Xtractor
from ..sharding import gather_xvectors, score_sharded, shard_range


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--trials", type=int, default=1000, help="enrolment models = test segments = this many (full trial mask)")
    args = ap.parse_args(argv)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    model = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
    model.compute_dtype = args.dtype
    L = int(args.seconds * 16000)
    start, stop = shard_range(args.utterances, rank, world)
    n_batches = (stop - start + args.batch - 1) // args.batch
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    blocks = []
    for k in range(n_batches):
        b = min(args.batch, stop - start - k * args.batch)
        g = torch.Generator(device=dev).manual_seed(1000 + rank * n_batches + k)
        wav = 0.1 * torch.randn(b, L, device=dev, generator=g)
        blocks.append(model(wav, is_eval=True)[1])
    local_xv = torch.cat(blocks) if blocks else torch.empty(0, 256, device=dev)
    torch.cuda.synchronize(dev)
    t_extract = time.perf_counter() - t0
    t0 = time.perf_counter()
    xv = gather_xvectors(local_xv)                      # (utterances, 256) on every rank
    torch.cuda.synchronize(dev)
    t_gather = time.perf_counter() - t0
    assert xv.shape == (args.utterances, 256)
    # scoring leg: synthetic speakers (config 5 recipe), enrolment rows sharded over the ranks
    rs = numpy.random.RandomState(0)
    n_spk, D, N = 250, 256, args.trials
    c = rs.randn(n_spk, D)
    spk_e, spk_t = rs.randint(0, n_spk, N), rs.randint(0, n_spk, N)
    norm = lambda x: x / numpy.linalg.norm(x, axis=1, keepdims=True)
    E, T = norm(c[spk_e] + 1.8 * rs.randn(N, D)), norm(c[spk_t] + 1.8 * rs.randn(N, D))
    t0 = time.perf_counter()
    rows = score_sharded(lambda a, b: torch.from_numpy(iv_scoring.cosine_matrix(E[a:b], T, dev)).to(dev), N)
    t_score = time.perf_counter() - t0
    if rank == 0:
        s = rows.cpu().numpy()
        tar = spk_e[:, None] == spk_t[None, :]
        eer = rocch2eer(*rocch(s[tar].astype(float), s[~tar].astype(float)))
        print(json.dumps({"ranks": world, "utterances": args.utterances, "x_vectors_per_s": args.utterances / t_extract,
                          "extract_s": t_extract, "all_gather_s": t_gather, "cosine_trials": N * N, "score_s": t_score, "eer": eer,
                          "dtype": args.dtype}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
