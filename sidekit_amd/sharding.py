"""Utterance sharding across the GPUs of one node.

The reference has no multi-GPU inference (``extract_xvectors.py`` is one process, one device).  Here
utterances are independent, so each rank (one process per GPU, ``torch.distributed`` over RCCL)
extracts a contiguous index range with replicated weights and the only exchange step of the path is
ONE ``all_gather`` of the ``(N_r, 256)`` float32 x-vector blocks, after which trial scoring shards by
enrolment rows.  The functions take the process group explicitly so that they run unchanged on the
``gloo`` backend in the CPU tests.
"""
import numpy
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous ``[start, stop)`` of ``n_items`` for ``rank``: sizes differ by at most one."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_by_length(lengths, world_size):
    """Length-balanced assignment for variable-length sets: longest-first round-robin ("snake") so every
    rank gets about the same number of frames.  Returns one index array per rank."""
    order = numpy.argsort(-numpy.asarray(lengths), kind="stable")
    shards = [[] for _ in range(world_size)]
    for pos, idx in enumerate(order):
        r = pos % (2 * world_size)
        shards[r if r < world_size else 2 * world_size - 1 - r].append(int(idx))
    return [numpy.array(sorted(s), dtype=numpy.int64) for s in shards]


def gather_xvectors(local, group=None):
    """All-gather per-rank ``(N_r, D)`` blocks (N_r may differ) into the full ``(sum N_r, D)`` matrix, rank order.

    One collective on the padded blocks (``all_gather_into_tensor``) plus a tiny count exchange.  With a process group of ONE rank the
    collectives still run (microseconds): a single-GPU launch under ``torch.distributed.run`` then exercises the same RCCL calls the
    N-rank job makes (``tests/test_gpu_00_multirank.py``).
    """
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, n_local, group=group)
    counts = counts.cpu().tolist()
    n_max = max(counts)
    padded = local
    if local.shape[0] < n_max:
        padded = torch.zeros((n_max, local.shape[1]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((world * n_max, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == n_max for c in counts):
        return out
    return torch.cat([out[r * n_max: r * n_max + counts[r]] for r in range(world)], dim=0)


def extract_sharded(extract_fn, n_utterances, group=None):
    """Run ``extract_fn(start, stop) -> (stop-start, D) tensor`` on this rank's range and gather everything."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    start, stop = shard_range(n_utterances, rank, world)
    return gather_xvectors(extract_fn(start, stop), group)


def score_sharded(score_rows_fn, n_enroll, group=None, dst=0):
    """Trial scoring sharded by enrolment rows: rank r computes ``score_rows_fn(start, stop) -> (stop-start, Nt)``;
    the row blocks are gathered on ``dst`` (returns the full matrix there, None elsewhere)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    start, stop = shard_range(n_enroll, rank, world)
    block = score_rows_fn(start, stop)
    if not dist.is_initialized():
        return block
    full = gather_xvectors(block, group)  # same ragged row gather
    return full if rank == dst else None
