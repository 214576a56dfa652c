"""Kaldi ark/scp restatement (CPU)."""
import struct

import numpy
import pytest

from sidekit_amd.kaldi_io import ArkScpWriter, read_ark, read_scp


def test_roundtrip_and_binary_layout(tmp_path):
    ark, scp = str(tmp_path / "x.ark"), str(tmp_path / "x.scp")
    rs = numpy.random.RandomState(0)
    items = {"utt-a": rs.randn(1, 256).astype(numpy.float32), "utt_b": rs.randn(256).astype(numpy.float32),
             "c": rs.randn(3, 5)}
    with ArkScpWriter(ark, scp) as w:
        for k, v in items.items():
            w(k, v)
    raw = open(ark, "rb").read()
    # first record, byte for byte: key, space, \0B, "FM ", \4 rows, \4 cols, payload
    assert raw.startswith(b"utt-a \0BFM \4" + struct.pack("<i", 1) + b"\4" + struct.pack("<i", 256))
    assert raw[6 + 15:6 + 15 + 1024] == items["utt-a"].tobytes()
    lines = open(scp).read().split("\n")
    assert lines[0] == f"utt-a {ark}:6"          # offset of the \0B marker
    back = dict(read_scp(scp))
    assert list(back) == list(items)
    for k in items:
        assert back[k].dtype == items[k].dtype and numpy.array_equal(back[k], items[k])
    assert [k for k, _ in read_ark(ark)] == list(items)
    with pytest.raises(ValueError):
        with ArkScpWriter(str(tmp_path / "y.ark")) as w:
            w("bad", numpy.zeros((2, 2, 2), dtype=numpy.float32))


def test_ordered_writer_matches_the_sequential_one_byte_for_byte(tmp_path):
    """Records written at pre-computed offsets in ANY arrival order give the file a sequential writer produces in key order (the
    reference's ark order, sidekit/bin/extract_xvectors.py:120,147); an interrupted run keeps what arrived."""
    from sidekit_amd.kaldi_io import OrderedArkWriter
    rs = numpy.random.RandomState(0)
    keys = [f"spk{i % 3}-utt{'x' * (i % 5)}{i}" for i in range(23)]
    vecs = {k: rs.randn(1, 16).astype(numpy.float32) for k in keys}
    with ArkScpWriter(str(tmp_path / "seq.ark"), str(tmp_path / "seq.scp")) as w:
        for k in keys:
            w(k, vecs[k])
    with OrderedArkWriter(str(tmp_path / "ord.ark"), str(tmp_path / "ord.scp"), keys, 16) as w:
        for i in rs.permutation(len(keys)):
            w(keys[i], vecs[keys[i]])
    assert open(tmp_path / "seq.ark", "rb").read() == open(tmp_path / "ord.ark", "rb").read()
    seq = [l.split()[0] + l.rpartition(":")[2] for l in open(tmp_path / "seq.scp")]
    assert seq == [l.split()[0] + l.rpartition(":")[2] for l in open(tmp_path / "ord.scp")]
    assert [k for k, _ in read_ark(str(tmp_path / "ord.ark"))] == keys
    # interrupted half way: the scp lists what arrived (arrival order) and every line resolves
    try:
        with OrderedArkWriter(str(tmp_path / "part.ark"), str(tmp_path / "part.scp"), keys, 16) as w:
            for k in keys[5:9]:
                w(k, vecs[k])
            raise KeyboardInterrupt
    except KeyboardInterrupt:
        pass
    got = list(read_scp(str(tmp_path / "part.scp")))
    assert [k for k, _ in got] == keys[5:9] and all(numpy.array_equal(v, vecs[k]) for k, v in got)
    with pytest.raises(KeyError):
        with OrderedArkWriter(str(tmp_path / "bad.ark"), None, keys, 16) as w:
            w("unknown", vecs[keys[0]])
