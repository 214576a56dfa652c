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


class OrderedArkWriter:
    """An ark whose records sit in a GIVEN key order whatever order they arrive in.

    The reference driver writes one ``(1, E)`` float-matrix record per utterance in wav.scp order
    (``sidekit/bin/extract_xvectors.py:120,147``); the streaming extractor returns utterances in length-sorted batches.
    Every record of this kind has a size its key alone decides (``len(key) + 16 + 4 E`` bytes), so all offsets are known
    before the first x-vector exists: a record is written at its own offset the moment it arrives (what was extracted
    survives an interruption; the unwritten records are holes of zeros) and the finished file is byte for byte the one a
    sequential writer produces.  The scp grows in arrival order beside it and is rewritten in key order by ``close``."""

    def __init__(self, ark_path, scp_path, keys, cols, dtype=numpy.float32):
        self.ark_path = os.path.realpath(ark_path)
        self.scp_path = scp_path
        self.cols, self.dtype = int(cols), numpy.dtype(dtype)
        if self.dtype not in (numpy.float32, numpy.float64):
            raise ValueError("kaldi matrices are float32 or float64")
        self._head = b"\0B" + (b"F" if self.dtype == numpy.float32 else b"D") + b"M " + b"\4" + struct.pack("<i", 1) + b"\4" + struct.pack("<i", self.cols)
        self._keys = list(keys)
        if len(set(self._keys)) != len(self._keys):
            raise ValueError("duplicate keys")
        self._offset, pos = {}, 0
        for k in self._keys:
            kb = k.encode() + b" "
            self._offset[k] = pos + len(kb)              # what the scp points at: the \0B marker
            pos += len(kb) + len(self._head) + self.cols * self.dtype.itemsize
        self._ark = open(self.ark_path, "wb")
        self._ark.truncate(pos)
        self._scp = open(scp_path, "w") if scp_path else None
        self._written = set()

    def __call__(self, key, array):
        a = numpy.ascontiguousarray(array, dtype=self.dtype).reshape(-1)
        if a.shape[0] != self.cols:
            raise ValueError(f"{key}: expected {self.cols} values, got shape {numpy.shape(array)}")
        off = self._offset[key]                          # KeyError: a key that was not announced
        kb = key.encode() + b" "
        self._ark.seek(off - len(kb))
        self._ark.write(kb + self._head + a.tobytes())
        self._written.add(key)
        line = f"{key} {self.ark_path}:{off}\n"
        if self._scp:
            self._scp.write(line)
        return line

    def flush(self):
        self._ark.flush()
        if self._scp:
            self._scp.flush()

    def close(self, complete=True):
        self._ark.close()
        if self._scp:
            self._scp.close()
            if complete and len(self._written) == len(self._keys):     # every record is there: the scp in key order, as the reference writes it
                with open(self.scp_path, "w") as f:
                    for k in self._keys:
                        f.write(f"{k} {self.ark_path}:{self._offset[k]}\n")

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *exc):
        self.close(complete=exc_type is None)


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
