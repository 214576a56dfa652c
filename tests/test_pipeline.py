"""Host side of the streaming extractor (sidekit_amd/pipeline.py) on the CPU: the wav parser against scipy on every sample
format the CLI accepts, the batch planner, and the decode -> batch -> stage -> forward -> collect loop with a stand-in model
(order, per-row lengths, padding that must not leak, int16 kept to the device and converted exactly)."""
import io
import struct

import numpy
import pytest
import scipy.io.wavfile
import torch

from sidekit_amd import pipeline


def _wav_bytes(sample, rate=16000, extra_chunk=False):
    buf = io.BytesIO()
    scipy.io.wavfile.write(buf, rate, sample)
    data = buf.getvalue()
    if extra_chunk:   # a LIST chunk of odd size (padded to even) between fmt and data, as ffmpeg / sox write
        i = data.index(b"data")
        chunk = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"
        data = data[:i] + chunk + data[i:]
        data = data[:4] + struct.pack("<I", len(data) - 8) + data[8:]
    return data


@pytest.mark.parametrize("dtype", ["int16", "int32", "uint8", "float32"])
@pytest.mark.parametrize("extra", [False, True])
def test_parse_wav_matches_scipy(dtype, extra):
    rs = numpy.random.RandomState(0)
    n = 1237
    if dtype == "float32":
        x = rs.uniform(-1, 1, n).astype(numpy.float32)
    elif dtype == "uint8":
        x = rs.randint(0, 256, n).astype(numpy.uint8)
    else:
        info = numpy.iinfo(dtype)
        x = rs.randint(info.min, info.max, n).astype(dtype)
    got, rate = pipeline.parse_wav(_wav_bytes(x, 8000, extra))
    assert rate == 8000 and got.shape == (n,)
    if dtype == "int16":
        assert got.dtype == numpy.int16 and numpy.array_equal(got, x)          # stays integer: converted on the device
    elif dtype == "int32":
        assert numpy.array_equal(got, x.astype(numpy.float32) / 2147483648.0)
    elif dtype == "uint8":
        assert numpy.array_equal(got, (x.astype(numpy.float32) - 128.0) / 128.0)
    else:
        assert numpy.array_equal(got, x)


def test_parse_wav_rejects_stereo_and_falls_back():
    with pytest.raises(IOError, match="mono"):
        pipeline.parse_wav(_wav_bytes(numpy.zeros((100, 2), dtype=numpy.int16)))
    with pytest.raises(IOError):
        pipeline.parse_wav(b"not a wav file at all")


def test_plan_batches():
    assert pipeline.plan_batches([5, 1, 3, 2, 9], 2) == [[1, 3], [2, 0], [4]]
    assert pipeline.plan_batches([], 4) == []
    assert pipeline.plan_batches([7, 7, 7], 8) == [[0, 1, 2]]
    # padded-sample budget: rows x longest row <= 20; a single over-long utterance still gets its own batch
    assert pipeline.plan_batches([4, 5, 6, 9, 30], 8, max_samples=20) == [[0, 1, 2], [3], [4]]


class _StubModel:
    """Embedding = (sum, sum of squares, length, first sample) of the row's own samples: anything read from the padding or
    from another row shows."""
    device = "cpu"

    def __init__(self):
        self.calls = []

    def __call__(self, x, is_eval=False, norm_embedding=True, lengths=None):
        assert is_eval and x.dtype in (torch.float32, torch.int16) and x.dim() == 2 and len(lengths) == x.shape[0]
        self.calls.append((x.shape[0], x.shape[1]))
        self.int16_batches = getattr(self, "int16_batches", 0) + (x.dtype == torch.int16)
        if x.dtype == torch.int16:          # 16-bit batches reach the model as the files hold them (xt_forward_pcm16 on the GPU)
            x = x.float() / 32768.0
        rows = []
        for r, n in enumerate(lengths):
            v = x[r, :n].double()
            rows.append(torch.stack([v.sum(), (v * v).sum(), torch.tensor(float(n), dtype=torch.float64), v[0]]))
        return None, torch.stack(rows).float()


def test_streaming_extractor_plumbing(tmp_path):
    rs = numpy.random.RandomState(1)
    entries, expect = [], {}
    for i in range(37):
        n = int(rs.randint(600, 5000))
        if i % 5 == 4:      # a float file among the int16 ones: its batch is staged as float32
            x = rs.uniform(-1, 1, n).astype(numpy.float32)
            f = x
        else:
            x = rs.randint(-32768, 32767, n).astype(numpy.int16)
            f = x.astype(numpy.float32) / 32768.0
        path = tmp_path / f"u{i}.wav"
        scipy.io.wavfile.write(path, 16000, x)
        src = f"cat {path} |" if i % 7 == 0 else str(path)          # wav.scp pipes run through the shell
        entries.append((f"utt{i}", src))
        d = f.astype(numpy.float64)
        expect[f"utt{i}"] = numpy.array([d.sum(), (d * d).sum(), n, d[0]])
    entries.append(("tensor", torch.linspace(-1, 1, 777)))          # an already decoded signal
    d = torch.linspace(-1, 1, 777).double().numpy()
    expect["tensor"] = numpy.array([d.sum(), (d * d).sum(), 777, d[0]])
    model = _StubModel()
    ex = pipeline.StreamingExtractor(model, batch_size=4, window=3, workers=3, pending=2)
    got = dict(ex.run(iter(entries)))
    assert set(got) == set(expect) and ex.stats["utterances"] == 38 and ex.stats["batches"] == len(model.calls)
    assert ex.stats["native_reads"] >= 5           # int16 files in all-int16 batches went file -> pinned row natively
    for k, v in expect.items():
        assert got[k].shape == (1, 4) and numpy.allclose(got[k][0], v, rtol=2e-6, atol=1e-4), k
    assert all(b <= 4 for b, _ in model.calls)
    # length-sorted inside each window of 12: padding stays small
    assert ex.stats["padded_samples"] < 1.6 * ex.stats["samples"]
    with pytest.raises(ValueError, match="sample rate"):
        scipy.io.wavfile.write(tmp_path / "r8.wav", 8000, numpy.zeros(900, dtype=numpy.int16))
        dict(pipeline.StreamingExtractor(model, batch_size=2).run([("bad", str(tmp_path / "r8.wav"))]))


def test_native_probe_and_read(tmp_path):
    """csrc/wav_io.cpp through the C-ABI: header walk (canonical, extra chunk, float, stereo, 8-bit, missing file) and payloads
    read into chosen rows of a strided int16 buffer."""
    rs = numpy.random.RandomState(2)
    a = rs.randint(-32768, 32767, 3001).astype(numpy.int16)
    b = rs.randint(-32768, 32767, 517).astype(numpy.int16)
    files = {"a.wav": _wav_bytes(a), "b.wav": _wav_bytes(b, 8000, extra_chunk=True), "f.wav": _wav_bytes(rs.rand(100).astype(numpy.float32)),
             "s.wav": _wav_bytes(numpy.zeros((50, 2), dtype=numpy.int16)), "u8.wav": _wav_bytes(rs.randint(0, 255, 64).astype(numpy.uint8)),
             "junk.wav": b"RIFFxxxxJUNK" + bytes(64)}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    names = list(files) + ["missing.wav"]
    kind, ns, rate, off = pipeline.probe_wavs([str(tmp_path / n) for n in names], threads=3)
    assert kind.tolist() == [1, 1, 0, 0, 0, 0, -1]
    assert ns[:2].tolist() == [3001, 517] and rate[:2].tolist() == [16000, 8000]
    dst = numpy.full((4, 3100), 7, dtype=numpy.int16)
    pipeline.read_pcm16([str(tmp_path / "a.wav"), str(tmp_path / "b.wav")], off[:2], ns[:2], [2, 0], dst[:, :3050], threads=2)
    assert numpy.array_equal(dst[2, :3001], a) and numpy.array_equal(dst[0, :517], b)
    assert (dst[1] == 7).all() and (dst[3] == 7).all() and (dst[2, 3001:] == 7).all() and (dst[0, 517:] == 7).all()
    with pytest.raises(IOError, match="short read"):
        pipeline.read_pcm16([str(tmp_path / "b.wav")], off[1:2], [5000], [0], numpy.zeros((1, 6000), dtype=numpy.int16))
    with pytest.raises(IOError, match="outside the staging buffer"):       # a row index past the buffer is refused, not written
        pipeline.read_pcm16([str(tmp_path / "b.wav")], off[1:2], ns[1:2], [4], dst)


def test_extract_xvectors_main_prechecks_and_writes_incrementally(tmp_path):
    """`bin.extract_xvectors.main`, one process: the whole shard is probed before the model is touched (an unreadable or too-short
    file raises up front and nothing has been extracted), x-vectors reach the ark as their batches come back and the scp ends up
    in wav.scp order."""
    from types import SimpleNamespace
    from sidekit_amd.bin import extract_xvectors
    from sidekit_amd.kaldi_io import read_scp
    rs = numpy.random.RandomState(5)
    model = _StubModel()
    model.embedding_size = 4
    model.preprocessor = SimpleNamespace(n_fft=1024)
    expect = {}
    with open(tmp_path / "wav.scp", "w") as f:
        for i, n in enumerate((3000, 900, 2100, 1500, 4000, 700, 2600)):
            x = rs.randint(-20000, 20000, n).astype(numpy.int16)
            scipy.io.wavfile.write(tmp_path / f"u{i}.wav", 16000, x)
            f.write(f"utt{i} {tmp_path / f'u{i}.wav'}\n")
            d = x.astype(numpy.float64) / 32768.0
            expect[f"utt{i}"] = numpy.array([d.sum(), (d * d).sum(), n, d[0]])
    extract_xvectors.main(model, str(tmp_path / "wav.scp"), str(tmp_path / "xv.scp"), "cpu", batch_size=2, workers=2, window=2)
    got = list(read_scp(str(tmp_path / "xv.scp")))
    assert [k for k, _ in got] == [f"utt{i}" for i in range(7)]                       # wav.scp order although batches are length sorted
    for k, v in got:
        assert v.shape == (1, 4) and numpy.allclose(v[0], expect[k], rtol=2e-6, atol=1e-4)
    # bad corpora are rejected before any forward
    for bad, exc, what in ((str(tmp_path / "missing.wav"), IOError, "cannot be opened"), (None, ValueError, "too short")):
        if bad is None:
            scipy.io.wavfile.write(tmp_path / "short.wav", 16000, numpy.zeros(400, dtype=numpy.int16))
            bad = str(tmp_path / "short.wav")
        with open(tmp_path / "bad.scp", "w") as f:
            f.write(f"utt0 {tmp_path / 'u0.wav'}\nbad {bad}\n")
        n_calls = len(model.calls)
        with pytest.raises(exc, match=what):
            extract_xvectors.main(model, str(tmp_path / "bad.scp"), str(tmp_path / "bad_xv.scp"), "cpu", batch_size=2, workers=2)
        assert len(model.calls) == n_calls
