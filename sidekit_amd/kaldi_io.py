"""Kaldi ark/scp float-matrix I/O -- just what the extraction drivers need.

The reference writes x-vectors with ``kaldiio.WriteHelper('ark,scp:...')`` and reads them back with
``kaldiio.ReadHelper('scp:...')`` (``sidekit/bin/extract_xvectors.py:120,147,164-173``,
``sidekit/bin/compute_spk_cosine.py:34-39``); ``kaldiio`` is not installed here, so the binary
layout is restated: every record is ``<key> <space> \\0 B F M <space> \\4 <int32 rows> \\4 <int32
cols> <rows*cols float32, row-major>`` (``DM`` + float64 for double matrices, ``FV`` / ``DV`` for
vectors), and an scp line is ``<key> <ark path>:<byte offset of the \\0B marker>``.
"""
import os
import struct

import numpy


class ArkScpWriter:
    """``with ArkScpWriter(ark_path, scp_path) as w: w(key, array)`` (float32 / float64, 1-D or 2-D)."""

    def __init__(self, ark_path, scp_path=None):
        self.ark_path = os.path.realpath(ark_path)
        self._ark = open(self.ark_path, "wb")
        self._scp = open(scp_path, "w") if scp_path else None

    def __call__(self, key, array):
        a = numpy.ascontiguousarray(array)
        if a.dtype not in (numpy.float32, numpy.float64):
            a = a.astype(numpy.float32)
        if a.ndim not in (1, 2):
            raise ValueError(f"kaldi matrices are 1-D or 2-D, got shape {a.shape}")
        self._ark.write(key.encode() + b" ")
        offset = self._ark.tell()
        kind = (b"F" if a.dtype == numpy.float32 else b"D") + (b"M " if a.ndim == 2 else b"V ")
        self._ark.write(b"\0B" + kind)
        for dim in a.shape:
            self._ark.write(b"\4" + struct.pack("<i", dim))
        self._ark.write(a.tobytes())
        line = f"{key} {self.ark_path}:{offset}\n"
        if self._scp:
            self._scp.write(line)
        return line

    def flush(self):
        self._ark.flush()
        if self._scp:
            self._scp.flush()

    def close(self):
        self._ark.close()
        if self._scp:
            self._scp.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _read_matrix(f):
    if f.read(2) != b"\0B":
        raise IOError("not a binary kaldi matrix (text-mode ark is not supported)")
    kind = f.read(3)
    if kind not in (b"FM ", b"DM ", b"FV ", b"DV "):
        raise IOError(f"unsupported kaldi object {kind!r} (compressed matrices are not supported)")
    dtype = numpy.float32 if kind[:1] == b"F" else numpy.float64
    dims = []
    for _ in range(2 if kind[1:2] == b"M" else 1):
        if f.read(1) != b"\4":
            raise IOError("corrupt kaldi header")
        dims.append(struct.unpack("<i", f.read(4))[0])
    n = int(numpy.prod(dims))
    data = numpy.frombuffer(f.read(n * dtype().itemsize), dtype=dtype)
    if data.size != n:
        raise IOError("truncated kaldi matrix")
    return data.reshape(dims).copy()


def read_scp(scp_path):
    """Generator of ``(key, array)`` in scp order (``kaldiio.ReadHelper('scp:...')``)."""
    handles = {}
    try:
        with open(scp_path) as scp:
            for line in scp:
                if not line.strip():
                    continue
                key, rx = line.split(None, 1)
                path, _, off = rx.strip().rpartition(":")
                if path not in handles:
                    handles[path] = open(path, "rb")
                f = handles[path]
                f.seek(int(off))
                yield key, _read_matrix(f)
    finally:
        for f in handles.values():
            f.close()


def read_ark(ark_path):
    """Generator of ``(key, array)`` over a binary ark file."""
    with open(ark_path, "rb") as f:
        while True:
            key = bytearray()
            while True:
                c = f.read(1)
                if not c:
                    return
                if c == b" ":
                    break
                key += c
            yield key.decode(), _read_matrix(f)
