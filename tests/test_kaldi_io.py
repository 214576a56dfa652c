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
