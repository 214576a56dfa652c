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
        rows = [torch.stack([x[r, :n].sum(), torch.tensor(float(n)), x[r, 0], x[r, n - 1]]) for r, n in enumerate(lengths)]
        return None, torch.stack(rows).float()


def _cli_worker(rank, world, port, wav_scp, out_scp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sidekit_amd.bin import extract_xvectors
    extract_xvectors.main(_StubXtractor(), wav_scp, out_scp, "cpu", batch_size=3, workers=2, window=2)
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
