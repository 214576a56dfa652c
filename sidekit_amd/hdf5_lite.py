"""A small HDF5 reader / writer for the files SIDEKIT exchanges (h5py is not installed on the GPU box).

Scope: exactly what ``sidekit/statserver.py:392-489``, ``sidekit/bosaris/{idmap,ndx,key,scores}.py`` and
``sidekit/sidekit_io.py`` read and write through h5py -- groups (symbol-table form: version-0 superblock, version-1 object
headers, v1 group B-trees + local heaps, the layout HDF5 1.8/1.10 and h5py produce by default) holding N-d datasets of
fixed-point, IEEE floating-point and fixed-length string types, stored compact, contiguous or chunked (v1 chunk B-tree)
behind the ``deflate`` / ``shuffle`` / ``fletcher32`` filters.  The writer emits the same family: every dataset is one
deflate + fletcher32 chunk with unlimited maximum dimensions (what ``create_dataset(..., maxshape=(None, ...),
compression="gzip", fletcher32=True)`` asks for), so files written here open in h5py / the reference and vice versa
(``tests/test_hdf5_io.py``; cross-checked against h5py 3.3.0 / HDF5 1.10.6 in the build container).

Not supported (raises ``NotImplementedError``): version-2+ superblocks with link-message groups, variable-length and
compound types, external / virtual storage, filters other than the three above.

Format reference: "HDF5 File Format Specification Version 2.0" (public HDF Group document), sections II (superblock),
III.A (B-trees), III.D (local heaps), IV.A (object headers and their messages).
"""
import os
import struct
import zlib

import numpy

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


# ---- Fletcher-32 as HDF5 computes it (H5_checksum_fletcher32: big-endian 16-bit words, end-around carry every 360 words) ----
def fletcher32(data):
    data = bytes(data)
    n = len(data) // 2
    words = numpy.frombuffer(data, dtype=">u2", count=n).astype(numpy.uint64)
    sum1 = sum2 = 0
    for i in range(0, n, 360):
        blk = words[i:i + 360]
        t = int(blk.shape[0])
        csum = numpy.cumsum(blk)
        sum2 += t * sum1 + int(csum.sum())
        sum1 += int(csum[-1])
        sum1 = (sum1 & 0xFFFF) + (sum1 >> 16)
        sum2 = (sum2 & 0xFFFF) + (sum2 >> 16)
    if len(data) % 2:
        sum1 += data[-1] << 8
        sum2 += sum1
        sum1 = (sum1 & 0xFFFF) + (sum1 >> 16)
        sum2 = (sum2 & 0xFFFF) + (sum2 >> 16)
    sum1 = (sum1 & 0xFFFF) + (sum1 >> 16)
    sum2 = (sum2 & 0xFFFF) + (sum2 >> 16)
    return ((sum2 << 16) | sum1) & 0xFFFFFFFF


# =================================================================================================
# reader
# =================================================================================================
class _Dataset:
    def __init__(self, f, shape, dtype, layout, filters):
        self._f, self.shape, self.dtype, self._layout, self._filters = f, tuple(shape), dtype, layout, filters

    def __getitem__(self, key):
        arr = self._read()
        return arr if key == () else arr[key]

    def read_direct(self, dest):
        dest[...] = self._read().astype(dest.dtype, copy=False)

    def _read(self):
        f, kind = self._f, self._layout[0]
        n = int(numpy.prod(self.shape)) if self.shape else 1
        isz = self.dtype.itemsize
        if kind == "compact":
            raw = self._layout[1]
        elif kind == "contiguous":
            addr, size = self._layout[1:]
            raw = b"\0" * (n * isz) if addr == UNDEF else f._buf[addr:addr + n * isz]
        else:
            return self._read_chunked()
        return numpy.frombuffer(raw, dtype=self.dtype, count=n).reshape(self.shape).copy()

    def _read_chunked(self):
        _, btree, cdims = self._layout
        out = numpy.zeros(self.shape, dtype=self.dtype)
        if btree == UNDEF:
            return out
        rank = len(self.shape)
        for offset, fmask, raw in self._f._chunks(btree, rank):
            for idx in range(len(self._filters) - 1, -1, -1):     # undo the pipeline in reverse order
                fid, cd = self._filters[idx]
                if fmask & (1 << idx):
                    continue
                if fid == 3:
                    stored = struct.unpack("<I", raw[-4:])[0]
                    raw = raw[:-4]
                    if fletcher32(raw) != stored:
                        raise IOError("HDF5 chunk failed its Fletcher-32 checksum")
                elif fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    isz = cd[0] if cd else self.dtype.itemsize
                    a = numpy.frombuffer(raw, dtype=numpy.uint8)
                    m = a.shape[0] // isz
                    raw = a[:m * isz].reshape(isz, m).T.tobytes() + a[m * isz:].tobytes()
                else:
                    raise NotImplementedError(f"HDF5 filter id {fid}")
            chunk = numpy.frombuffer(raw, dtype=self.dtype, count=int(numpy.prod(cdims))).reshape(cdims)
            sel_out, sel_in = [], []
            for d in range(rank):
                lo = offset[d]
                hi = min(lo + cdims[d], self.shape[d])
                if hi <= lo:
                    break
                sel_out.append(slice(lo, hi))
                sel_in.append(slice(0, hi - lo))
            else:
                out[tuple(sel_out)] = chunk[tuple(sel_in)]
        return out


class _Group:
    def __init__(self, f, btree, heap):
        self._f = f
        self._links = f._group_links(btree, heap)       # name -> object header address

    def keys(self):
        return list(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, _Group) or part not in node._links:
                raise KeyError(path)
            node = node._f._object(node._links[part])
        return node

    def get(self, path, default=None):
        try:
            return self[path]
        except KeyError:
            return default


class File(_Group):
    """Read-only view of an HDF5 file: ``File(name)["group/dataset"][()]`` / ``.get(...)`` / ``.keys()`` as in h5py."""

    def __init__(self, name):
        with open(name, "rb") as fh:
            self._buf = fh.read()
        b = self._buf
        if b[:8] != SIGNATURE:
            raise IOError(f"{name}: not an HDF5 file")
        version = b[8]
        if version not in (0, 1):
            raise NotImplementedError(f"{name}: HDF5 superblock version {version} (only the symbol-table layout of versions 0/1 is read)")
        if b[13] != 8 or b[14] != 8:
            raise NotImplementedError("HDF5 files with offsets / lengths other than 8 bytes")
        pos = 24 + (4 if version == 1 else 0)
        base, _free, _eof, _drv = struct.unpack_from("<QQQQ", b, pos)
        if base != 0:
            raise NotImplementedError("HDF5 user block / non-zero base address")
        ste = pos + 32
        _name_off, ohdr, cache_type = struct.unpack_from("<QQI", b, ste)
        if cache_type == 1:
            btree, heap = struct.unpack_from("<QQ", b, ste + 24)
        else:
            btree, heap = self._symbol_table_message(ohdr)
        _Group.__init__(self, self, btree, heap)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    # ---- object headers (version 1) -----------------------------------------------------------
    def _messages(self, addr):
        b = self._buf
        if b[addr:addr + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers")
        version, _, nmsg, _refs, hsize = struct.unpack_from("<BBHII", b, addr)
        if version != 1:
            raise NotImplementedError(f"object header version {version}")
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                                    # continuation block
                    blocks.append(struct.unpack_from("<QQ", body, 0))
                out.append((mtype, body))
        return out

    def _symbol_table_message(self, addr):
        for mtype, body in self._messages(addr):
            if mtype == 0x0011:
                return struct.unpack_from("<QQ", body, 0)
        raise NotImplementedError("group without a symbol-table message (link-message groups are not read)")

    def _object(self, addr):
        msgs = self._messages(addr)
        types = {m for m, _ in msgs}
        if 0x0011 in types:
            btree, heap = next(struct.unpack_from("<QQ", body, 0) for m, body in msgs if m == 0x0011)
            return _Group(self, btree, heap)
        if 0x0008 not in types:
            raise NotImplementedError("object that is neither a symbol-table group nor a dataset")
        shape, dtype, layout, filters = (), None, None, []
        for mtype, body in msgs:
            if mtype == 0x0001:
                shape = self._dataspace(body)
            elif mtype == 0x0003:
                dtype = self._datatype(body)
            elif mtype == 0x0008:
                layout = self._layout(body)
            elif mtype == 0x000B:
                filters = self._pipeline(body)
        if layout[0] == "chunked":
            layout = (layout[0], layout[1], layout[2][:len(shape)])
        return _Dataset(self, shape, dtype, layout, filters)

    @staticmethod
    def _dataspace(body):
        version, rank, flags = struct.unpack_from("<BBB", body, 0)
        if version == 1:
            pos = 8
        elif version == 2:
            pos = 4
            if body[3] == 2:      # null dataspace
                return (0,)
        else:
            raise NotImplementedError(f"dataspace message version {version}")
        return struct.unpack_from("<%dQ" % rank, body, pos)

    @staticmethod
    def _datatype(body):
        cv, b0, b1, _b2, size = struct.unpack_from("<BBBBI", body, 0)
        cls = cv & 0x0F
        order = ">" if (b0 & 1) else "<"
        if cls == 0:
            return numpy.dtype(f"{order}{'i' if b0 & 0x08 else 'u'}{size}")
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte floating point")
            return numpy.dtype(f"{order}f{size}")
        if cls == 3:
            return numpy.dtype(f"S{size}")
        raise NotImplementedError(f"HDF5 datatype class {cls} (only integers, IEEE floats and fixed-length strings)")

    @staticmethod
    def _layout(body):
        version, cls = body[0], body[1]
        if version != 3:
            raise NotImplementedError(f"data layout message version {version}")
        if cls == 0:
            size = struct.unpack_from("<H", body, 2)[0]
            return ("compact", bytes(body[4:4 + size]))
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", body, 2)
            return ("contiguous", addr, size)
        if cls == 2:
            ndim = body[2]
            btree = struct.unpack_from("<Q", body, 3)[0]
            dims = struct.unpack_from("<%dI" % ndim, body, 11)
            return ("chunked", btree, tuple(dims))
        raise NotImplementedError(f"data layout class {cls}")

    @staticmethod
    def _pipeline(body):
        version, nfilters = body[0], body[1]
        pos = 8 if version == 1 else 2
        out = []
        for _ in range(nfilters):
            fid = struct.unpack_from("<H", body, pos)[0]
            if version == 1 or fid >= 256:
                namelen, _flags, ncd = struct.unpack_from("<HHH", body, pos + 2)
                pos += 8
            else:
                namelen = 0
                _flags, ncd = struct.unpack_from("<HH", body, pos + 2)
                pos += 6
            pos += (namelen + 7) // 8 * 8 if version == 1 else namelen
            cd = struct.unpack_from("<%dI" % ncd, body, pos)
            pos += 4 * ncd
            if version == 1 and ncd % 2:
                pos += 4
            out.append((fid, cd))
        return out

    # ---- group B-tree + heap -----------------------------------------------------------------
    def _group_links(self, btree, heap):
        b = self._buf
        if b[heap:heap + 4] != b"HEAP":
            raise IOError("bad local heap signature")
        data_addr = struct.unpack_from("<Q", b, heap + 24)[0]
        links = {}

        def name_at(off):
            end = b.index(b"\0", data_addr + off)
            return b[data_addr + off:end].decode()

        def walk(addr):
            if b[addr:addr + 4] == b"SNOD":
                nsym = struct.unpack_from("<H", b, addr + 6)[0]
                for i in range(nsym):
                    off, ohdr = struct.unpack_from("<QQ", b, addr + 8 + 40 * i)
                    links[name_at(off)] = ohdr
                return
            if b[addr:addr + 4] != b"TREE":
                raise IOError("bad group B-tree node")
            _ntype, _level, used = struct.unpack_from("<BBH", b, addr + 4)
            pos = addr + 24
            for i in range(used):
                child = struct.unpack_from("<Q", b, pos + 8)[0]       # key_i, child_i, key_i+1 ...
                walk(child)
                pos += 16

        if btree != UNDEF:
            walk(btree)
        return links

    def _chunks(self, addr, rank):
        """Yield (offset tuple, filter mask, raw bytes) of every chunk under a v1 chunk B-tree."""
        b = self._buf
        if b[addr:addr + 4] != b"TREE":
            raise IOError("bad chunk B-tree node")
        ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
        if ntype != 1:
            raise IOError("not a raw-data chunk B-tree")
        keysz = 8 + 8 * (rank + 1)
        pos = addr + 24
        for _ in range(used):
            size, fmask = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from("<%dQ" % (rank + 1), b, pos + 8)
            child = struct.unpack_from("<Q", b, pos + keysz)[0]
            if level == 0:
                yield offs[:rank], fmask, b[child:child + size]
            else:
                yield from self._chunks(child, rank)
            pos += keysz + 8


# =================================================================================================
# writer
# =================================================================================================
def _pad8(x):
    return x + b"\0" * (-len(x) % 8)


def _datatype_message(dt):
    dt = numpy.dtype(dt)
    if dt.byteorder == ">":
        raise NotImplementedError("big-endian datasets are not written")
    if dt.kind in "iu":
        return struct.pack("<BBBBIHH", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        e, m = (8, 23) if dt.itemsize == 4 else (11, 52)
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 8 * dt.itemsize - 1, 0, dt.itemsize, 0, 8 * dt.itemsize, m, e, 0, m, (1 << (e - 1)) - 1)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)      # null-padded ASCII, what h5py makes of numpy 'S'
    raise NotImplementedError(f"numpy dtype {dt} has no HDF5 mapping here")


class Writer:
    """Collect ``path -> array`` pairs, then serialise them in one pass: ``w = Writer(); w["plda/mean"] = a; w.save(name)``."""

    LEAF_K, NODE_K = 32, 16       # symbol-table node / group B-tree fan-out recorded in the superblock

    def __init__(self, gzip_level=4):
        self._tree = {}
        self._level = gzip_level

    def __setitem__(self, path, array):
        parts = [p for p in path.split("/") if p]
        node = self._tree
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise ValueError(f"{path}: {p} is a dataset")
        a = numpy.asarray(array)
        if a.dtype.kind == "U":
            a = a.astype("S")
        if a.dtype.kind == "b":
            a = a.astype("int8")
        if a.dtype.kind == "S" and a.dtype.itemsize == 0:
            a = a.astype("S1")
        node[parts[-1]] = numpy.require(a, requirements="C")     # ascontiguousarray would turn a 0-d array into shape (1,)

    def save(self, name):
        self._out = bytearray(96)                       # superblock (56) + root symbol-table entry (40), filled in last
        root_hdr, btree, heap = self._write_group(self._tree)
        eof = len(self._out)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.NODE_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", btree, heap)
        self._out[:96] = sb
        # written beside the target and moved over it in one step: a crash or an exception while saving never leaves a truncated
        # file where the previous one was (StatServer.write(mode='a') rewrites a file that holds earlier statistics)
        tmp = f"{name}.tmp{os.getpid()}"
        try:
            with open(tmp, "wb") as fh:
                fh.write(bytes(self._out))
                fh.flush()
                os.fsync(fh.fileno())
            os.replace(tmp, name)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    # ---- pieces ---------------------------------------------------------------------------------
    def _alloc(self, data):
        self._out += b"\0" * (-len(self._out) % 8)
        addr = len(self._out)
        self._out += data
        return addr

    def _object_header(self, messages):
        body = b"".join(struct.pack("<HHBBBB", t, len(_pad8(m)), f, 0, 0, 0) + _pad8(m) for t, m, f in messages)
        return self._alloc(struct.pack("<BBHII", 1, 0, len(messages), 1, len(body)) + b"\0" * 4 + body)

    def _write_group(self, tree):
        if len(tree) > 2 * self.LEAF_K:
            raise NotImplementedError(f"more than {2 * self.LEAF_K} entries in one group")
        names = sorted(tree, key=lambda s: s.encode())
        entries = []
        for n in names:
            child = tree[n]
            if isinstance(child, dict):
                hdr, bt, hp = self._write_group(child)
                entries.append((n, hdr, 1, bt, hp))
            else:
                entries.append((n, self._write_dataset(child), 0, 0, 0))
        # local heap: the empty string at offset 0, then the names, then one free block
        heap_data = bytearray(8)
        offs = []
        for n, *_ in entries:
            offs.append(len(heap_data))
            heap_data += _pad8(n.encode() + b"\0")
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 32) + b"\0" * 16           # free block: next = H5HL_FREE_NULL (1), 32 bytes long
        data_addr = self._alloc(bytes(heap_data))
        heap = self._alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), free_off, data_addr))
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(entries))
        for (n, hdr, ctype, bt, hp), off in zip(entries, offs):
            snod += struct.pack("<QQII", off, hdr, ctype, 0) + (struct.pack("<QQ", bt, hp) if ctype == 1 else b"\0" * 16)
        snod += b"\0" * (40 * (2 * self.LEAF_K - len(entries)))
        snod_addr = self._alloc(snod)
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if entries else 0, UNDEF, UNDEF)
        node += struct.pack("<QQQ", 0, snod_addr, offs[-1] if offs else 0)
        node += b"\0" * (24 + 8 * (2 * self.NODE_K + 1) + 8 * 2 * self.NODE_K - len(node))
        btree = self._alloc(node)
        hdr = self._object_header([(0x0011, struct.pack("<QQ", btree, heap), 0)])
        return hdr, btree, heap

    CHUNK_BYTES = 1 << 20          # h5py's guess_chunk aims at about this much per chunk
    CHUNK_K = 32                   # chunk B-tree fan-out (2K = 64 entries per node; the library default, not recorded in the file)

    def _chunk_rows(self, a):
        """Chunk = `rows` leading-dimension slices x the full trailing extent: about CHUNK_BYTES each, more only when a two-level
        B-tree (64 x 64 chunks) would not hold the dataset otherwise, and never 4 GiB (the chunk size field is 32 bits)."""
        row_bytes = max(1, a.dtype.itemsize * int(numpy.prod(a.shape[1:], dtype=numpy.int64)))
        target = max(self.CHUNK_BYTES, -(-a.nbytes // (4 * self.CHUNK_K * self.CHUNK_K - 64)))
        rows = max(1, min(a.shape[0], target // row_bytes))
        if rows * row_bytes >= 1 << 32:
            raise NotImplementedError("one leading-dimension slice of the dataset exceeds HDF5's 4 GiB chunk limit")
        return rows

    def _chunk_btree(self, entries, rank, end_key):
        """v1 B-tree (node type 1) over `entries` = [(key offsets, nbytes, address)], leaves of up to 2K chunks, one more level above
        them when needed.  A node's keys are the first offsets of its children plus one closing key."""
        keysz = 8 + 8 * (rank + 1)
        cap = 2 * self.CHUNK_K

        def node(level, items, closing):           # items: [(offsets, nbytes, child address)]
            body = b"TREE" + struct.pack("<BBHQQ", 1, level, len(items), UNDEF, UNDEF)
            for offs, nbytes, addr in items:
                body += struct.pack("<II", nbytes, 0) + struct.pack("<%dQ" % (rank + 1), *(list(offs) + [0])) + struct.pack("<Q", addr)
            body += struct.pack("<II", 0, 0) + struct.pack("<%dQ" % (rank + 1), *(list(closing) + [0]))
            body += b"\0" * (24 + (cap + 1) * keysz + cap * 8 - len(body))
            return self._alloc(body)

        if len(entries) <= cap:
            return node(0, entries, end_key)
        if len(entries) > cap * cap:
            raise NotImplementedError("more chunks than a two-level B-tree holds")
        groups = [entries[i:i + cap] for i in range(0, len(entries), cap)]
        leaves = []
        for gi, g in enumerate(groups):
            closing = groups[gi + 1][0][0] if gi + 1 < len(groups) else end_key
            leaves.append((g[0][0], g[0][1], node(0, g, closing)))
        # sibling pointers of the leaves are left undefined: readers (libhdf5, this module) descend from the root
        return node(1, leaves, end_key)

    def _write_dataset(self, a):
        rank = a.ndim
        space = struct.pack("<BBBBI", 1, rank, 1 if rank else 0, 0, 0) + struct.pack("<%dQ" % rank, *a.shape)
        space += struct.pack("<%dQ" % rank, *([UNDEF] * rank))            # maxshape=(None, ...)
        dtype = _datatype_message(a.dtype)
        if rank == 0:                                                     # scalar: contiguous, no filters
            fill = struct.pack("<BBBB", 2, 2, 0, 0)
            raw = a.tobytes()
            layout = struct.pack("<BBQQ", 3, 1, self._alloc(raw), len(raw))
            return self._object_header([(0x0001, space, 0), (0x0003, dtype, 1), (0x0005, fill, 1), (0x0008, layout, 0)])
        # Extendible (maxshape None) datasets must be chunked for libhdf5 to resize them -- the reference appends to its files in
        # place (StatServer.write(mode='a'), sidekit/statserver.py:430-489).  An empty dataset gets a chunk shape and no chunk
        # (B-tree address undefined), a non-empty one ~1-MiB chunks along the leading dimension (one chunk of the full shape used
        # to make every later append allocate a whole-dataset-sized chunk, and broke at 4 GiB).
        fill = struct.pack("<BBBB", 2, 3, 2, 0)                           # allocate incrementally, fill value undefined-at-write
        pipe = struct.pack("<BBHI", 1, 2, 0, 0)
        pipe += struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<II", self._level, 0)      # deflate (optional), one client value + padding
        pipe += struct.pack("<HHHH", 3, 0, 0, 0)                                            # fletcher32
        if a.size == 0:
            cdims = [max(1, min(d, 1024)) if d else 1 for d in a.shape]
            cdims[0] = max(1, min(1024, self.CHUNK_BYTES // max(1, a.dtype.itemsize * int(numpy.prod(cdims[1:], dtype=numpy.int64)))))
            btree = UNDEF
        else:
            rows = self._chunk_rows(a)
            cdims = [rows] + list(a.shape[1:])
            entries = []
            for r0 in range(0, a.shape[0], rows):
                block = a[r0:r0 + rows]
                if block.shape[0] < rows:                                  # edge chunks are stored whole
                    pad = numpy.zeros([rows] + list(a.shape[1:]), dtype=a.dtype)
                    pad[:block.shape[0]] = block
                    block = pad
                raw = zlib.compress(numpy.ascontiguousarray(block).tobytes(), self._level)
                raw += struct.pack("<I", fletcher32(raw))
                entries.append(([r0] + [0] * (rank - 1), len(raw), self._alloc(raw)))
            end_key = [-(-a.shape[0] // rows) * rows] + [0] * (rank - 1)
            btree = self._chunk_btree(entries, rank, end_key)
        layout = struct.pack("<BBBQ", 3, 2, rank + 1, btree) + struct.pack("<%dI" % (rank + 1), *(cdims + [a.dtype.itemsize]))
        return self._object_header([(0x0001, space, 0), (0x0003, dtype, 1), (0x0005, fill, 1), (0x000B, pipe, 0), (0x0008, layout, 0)])


def read_all(name):
    """Every dataset of a file as ``{path: array}`` (``sidekit_io.read_dict_hdf5`` style, any depth)."""
    out = {}

    def walk(group, prefix):
        for k in group.keys():
            obj = group[k]
            if isinstance(obj, _Group):
                walk(obj, prefix + k + "/")
            else:
                out[prefix + k] = obj[()]

    with File(name) as f:
        walk(f, "")
    return out
