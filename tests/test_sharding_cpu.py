"""Utterance sharding + x-vector gather + row-sharded scoring on 2 gloo ranks (CPU)."""
import os
import socket

import numpy
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sidekit_amd.sharding import extract_sharded, gather_xvectors, score_sharded, shard_by_length, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_utt, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = torch.arange(n_utt * 8, dtype=torch.float32).reshape(n_utt, 8)   # "x-vector" of utterance i = row i

    def extract(start, stop):
        return table[start:stop].clone()

    full = extract_sharded(extract, n_utt)
    assert torch.equal(full, table)
    # ragged blocks straight through the gather
    mine = table[:3 + 2 * rank]
    g = gather_xvectors(mine)
    assert torch.equal(g, torch.cat([table[:3 + 2 * r] for r in range(world)]))
    # scoring sharded by enrolment rows, gathered on rank 0
    test = torch.nn.functional.normalize(torch.randn(5, 8, generator=torch.Generator().manual_seed(0)), dim=1)
    enr = torch.nn.functional.normalize(table[:n_utt] + 1.0, dim=1)
    scores = score_sharded(lambda a, b: enr[a:b] @ test.t(), n_utt)
    if rank == 0:
        assert torch.allclose(scores, enr @ test.t())
        numpy.save(os.path.join(out_dir, "ok.npy"), numpy.ones(1))
    else:
        assert scores is None
    dist.destroy_process_group()


def test_two_rank_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 11, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def test_shard_ranges_partition():
    for n in (0, 1, 7, 8, 100000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(100000, 3, 8) == (37500, 50000)              # BASELINE config 3: 12500 per rank
    lens = numpy.random.RandomState(0).randint(32000, 160001, 512)  # BASELINE config 4 lengths
    parts = shard_by_length(lens, 8)
    assert sorted(numpy.concatenate(parts).tolist()) == list(range(512))
    loads = [lens[p].sum() for p in parts]
    assert (max(loads) - min(loads)) / numpy.mean(loads) < 0.02


class _StubXtractor:
    """Stands in for the GPU model in the 2-rank CLI test: embedding = (sum, length, first, last sample)."""
    device = "cpu"
    embedding_size = 4
    compute_dtype = "fp32"

    def __call__(self, x, is_eval=False, norm_embedding=True, lengths=None):
        x = x.float() / 32768.0 if x.dtype == torch.int16 else x
        rows = [torch.stack([x[r, :n].sum(), torch.tensor(float(n)), x[r, 0], x[r, n - 1]]) for r, n in enumerate(lengths)]
        return None, torch.stack(rows).float()


def _cli_worker(rank, world, port, wav_scp, out_scp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world))   # what torch.distributed.run exports
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sidekit_amd.bin import extract_xvectors
    seen = []

    class Recording(extract_xvectors.StreamingExtractor):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            seen.append(self.workers)

    extract_xvectors.StreamingExtractor = Recording
    extract_xvectors.main(_StubXtractor(), wav_scp, out_scp, "cpu", batch_size=3, window=2)      # workers: this rank's share of the host
    with open(f"{out_scp}.workers{rank}", "w") as f:
        f.write(str(seen[0]))
    dist.destroy_process_group()


def test_extract_xvectors_cli_shards_the_wav_scp(tmp_path):
    """`extract_xvectors.main` under an initialised 2-rank group: contiguous wav.scp shards (7 = 4 + 3 utterances), one ragged
    gather, rank 0 alone writes the ark / scp in wav.scp order."""
    import scipy.io.wavfile
    from sidekit_amd.kaldi_io import read_scp
    rs = numpy.random.RandomState(0)
    expect = {}
    with open(tmp_path / "wav.scp", "w") as f:
        for i in range(7):
            x = rs.randint(-20000, 20000, rs.randint(700, 3000)).astype(numpy.int16)
            scipy.io.wavfile.write(tmp_path / f"u{i}.wav", 16000, x)
            f.write(f"utt{i} {tmp_path / f'u{i}.wav'}\n")
            v = x.astype(numpy.float32) / 32768.0
            expect[f"utt{i}"] = numpy.array([v.astype(numpy.float64).sum(), len(v), v[0], v[-1]])
    port = _free_port()
    mp.spawn(_cli_worker, args=(2, port, str(tmp_path / "wav.scp"), str(tmp_path / "xv.scp")), nprocs=2, join=True)
    got = dict(read_scp(str(tmp_path / "xv.scp")))
    assert list(got) == [f"utt{i}" for i in range(7)]
    for k, v in expect.items():
        assert got[k].shape == (1, 4) and numpy.allclose(got[k][0], v, rtol=1e-5, atol=1e-3), k
    # SURVEY 8e, host side: two ranks on one host take half of the cores each for decoding / staging (at most 8), not 8 each
    share = max(1, min(8, len(os.sched_getaffinity(0)) // 2))
    for r in range(2):
        assert int(open(f"{tmp_path / 'xv.scp'}.workers{r}").read()) == share
    # one process, no process group: the same ark BYTES although its batches come back length-sorted (the reference writes the ark in
    # wav.scp order, extract_xvectors.py:120,147; a Kaldi consumer reading `ark:` sequentially must not see the launch mode)
    from sidekit_amd.bin import extract_xvectors
    (tmp_path / "one").mkdir()
    extract_xvectors.main(_StubXtractor(), str(tmp_path / "wav.scp"), str(tmp_path / "one" / "xv.scp"), "cpu", batch_size=3, workers=2, window=2)
    assert open(tmp_path / "one" / "xv.ark", "rb").read() == open(tmp_path / "xv.ark", "rb").read()
    one = [l.split() for l in open(tmp_path / "one" / "xv.scp")]
    two = [l.split() for l in open(tmp_path / "xv.scp")]
    assert [k for k, _ in one] == [k for k, _ in two] and [rx.rpartition(":")[2] for _, rx in one] == [rx.rpartition(":")[2] for _, rx in two]


# ---- the sharded extraction + scoring driver itself (bin/shard_extract_score.py) on 2 gloo ranks -------------------------------
class _BandEnergyXtractor:
    """CPU stand-in for the GPU model inside ``shard_extract_score.main``: 16 log band energies of the first 2048 samples,
    L2-normalised -- enough for the synthetic sinusoid speakers to separate with a non-trivial EER."""
    embedding_size = 16
    compute_dtype = "fp32"

    def __call__(self, x, is_eval=False):
        p = torch.fft.rfft(x[:, :2048].double() * torch.hann_window(2048, dtype=torch.float64), dim=1).abs().pow(2)[:, :1024]
        e = torch.log(p.reshape(x.shape[0], 16, 64).sum(dim=2) + 1e-9)
        e = e - e.mean(dim=1, keepdim=True)
        return None, torch.nn.functional.normalize(e, dim=1).float()


class _CpuScoring:
    """CPU stand-ins with the signatures and return types of ``iv_scoring.cosine_matrix_device / plda_matrix_device /
    cosine_histograms`` (the GPU entry points the driver calls); row independent on purpose (one dot product per score), so that
    a row shard of the matrix equals those rows of the whole matrix bit for bit."""

    @staticmethod
    def cosine_matrix_device(e, t, device=None):
        return (e.double()[:, None, :] * t.double()[None, :, :]).sum(dim=2).float()

    @staticmethod
    def plda_matrix_device(e, t, Phi, Psi, cst, scaling_factor=1., device=None):
        Phi, Psi = torch.as_tensor(Phi), torch.as_tensor(Psi)
        qe, qt = 0.5 * ((e @ Phi) * e).sum(dim=1), 0.5 * ((t @ Phi) * t).sum(dim=1)
        cross = ((e @ Psi)[:, None, :] * t[None, :, :]).sum(dim=2)
        return scaling_factor * (qe[:, None] + qt[None, :] + cst + cross)

    @staticmethod
    def cosine_histograms(e, t, le, lt, self_offset=None, lo=-1.0, hi=1.0, device=None):
        from sidekit_amd.iv_scoring import HIST_BINS
        s = _CpuScoring.cosine_matrix_device(e, t).numpy()
        keep = numpy.ones(s.shape, dtype=bool)
        if self_offset is not None:
            i = numpy.arange(s.shape[0])
            keep[i, i + self_offset] = False
        tar = (le.cpu().numpy()[:, None] == lt.cpu().numpy()[None, :])
        b = numpy.clip(numpy.floor((s - lo) / (hi - lo) * HIST_BINS).astype(numpy.int64), 0, HIST_BINS - 1)
        return (numpy.bincount(b[keep & tar], minlength=HIST_BINS).astype(numpy.uint64),
                numpy.bincount(b[keep & ~tar], minlength=HIST_BINS).astype(numpy.uint64))


_DRIVER_ARGS = ["--utterances", "192", "--batch", "16", "--seconds", "0.2", "--trials", "48", "--speakers", "12", "--plda-rank", "6",
                "--noise", "0.1", "--all-pairs", "--backend", "gloo", "--device", "cpu"]


def _driver_worker(rank, world, port, out_dir):
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sidekit_amd.bin import shard_extract_score
    out = shard_extract_score.main(_DRIVER_ARGS, model=_BandEnergyXtractor(), scoring=_CpuScoring)   # opens the gloo group itself
    assert (out is not None) == (rank == 0)
    if rank == 0:
        with open(os.path.join(out_dir, "two_ranks.json"), "w") as f:
            json.dump(out, f)


def test_shard_extract_score_driver_on_two_ranks(tmp_path, monkeypatch):
    """The driver's real control flow on 2 gloo ranks against its own 1-rank run: contiguous shards (96 + 96 utterances, batch
    aligned so both runs synthesise the same waveforms), ONE ragged gather, cosine + PLDA row shards gathered on rank 0,
    ``cosine_histograms(self_offset=a)`` per rank + the counter all-reduce; every EER and the pair count equal the 1-rank run's."""
    import json
    from sidekit_amd.bin import shard_extract_score
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    one = shard_extract_score.main(_DRIVER_ARGS, model=_BandEnergyXtractor(), scoring=_CpuScoring)
    assert one["ranks"] == 1 and one["all_pairs"] == 192 * 191 and 0.0 < one["cosine_eer"] < 0.5
    port = _free_port()
    mp.spawn(_driver_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    with open(tmp_path / "two_ranks.json") as f:
        two = json.load(f)
    assert two["ranks"] == 2 and two["utterances"] == 192 and two["trials"] == 48 * 48
    assert two["all_pairs"] == one["all_pairs"]
    for k in ("cosine_eer", "plda_eer", "all_pairs_eer"):
        assert two[k] == one[k], (k, one[k], two[k])
    # round 6: two runs that are to be compared share their histogram edges (--hist-range), and --seed draws another corpus
    assert two["all_pairs_hist_range"] == one["all_pairs_hist_range"] and one["all_pairs_hist_bins"] == 8192
    lo, hi = one["all_pairs_hist_range"]
    same = shard_extract_score.main(_DRIVER_ARGS + ["--hist-range", repr(lo), repr(hi)], model=_BandEnergyXtractor(), scoring=_CpuScoring)
    assert same["all_pairs_hist_range"] == [lo, hi] and same["all_pairs_eer"] == one["all_pairs_eer"]
    wide = shard_extract_score.main(_DRIVER_ARGS + ["--hist-range", "-1.0", "1.0"], model=_BandEnergyXtractor(), scoring=_CpuScoring)
    assert wide["all_pairs_hist_range"] == [-1.0, 1.0] and wide["all_pairs"] == one["all_pairs"] and abs(wide["all_pairs_eer"] - one["all_pairs_eer"]) < 0.02
    other = shard_extract_score.main(_DRIVER_ARGS + ["--seed", "3"], model=_BandEnergyXtractor(), scoring=_CpuScoring)
    assert other["all_pairs"] == one["all_pairs"] and other["cosine_eer"] != one["cosine_eer"]
    # the injected-module switch is not a fallback: the product entry point refuses a CPU device
    import pytest
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        shard_extract_score.main(["--device", "cpu", "--utterances", "192", "--trials", "48", "--speakers", "12"])
