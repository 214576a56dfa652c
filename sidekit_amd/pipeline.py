"""Streaming wav -> x-vector pipeline around ``Xtractor.forward`` (the caller's side of the hot path).

The reference driver (``sidekit/bin/extract_xvectors.py:130-150``) reads one file, runs the model on it, copies the
embedding back, and only then touches the next file.  At 38 k x-vectors/s the extractor needs 10 GB/s of waveform: the
host side has to run beside the GPU, not in front of it.  ``StreamingExtractor`` keeps four stages busy at once:

  probe    the headers of a window of files are walked by native threads (``sk_wav_probe``, csrc/wav_io.cpp): lengths are known
           before a sample is read; pipes, float / 8 / 32-bit files and arrays are decoded by a Python thread pool instead;
  batch    the window is sorted by length and cut into padded batches (every row is still computed over its own length,
           SURVEY N2);
  stage    native threads read the PCM16 payloads STRAIGHT INTO rows of a pinned staging buffer (``sk_wav_read_pcm16``: no
           interpreter lock, no intermediate copy) -- 16-bit audio stays int16 all the way to the device (half the PCIe
           bytes, no host conversion pass) -- while the main thread launches earlier batches;
  copy     host -> device on a copy stream; the forward waits on the copy's event, not on the host;
  compute  ``Xtractor.forward`` on the int16 rows themselves (``xt_forward_pcm16``: the STFT kernel's load widens them as
           ``x / 32768``, exact, the same numbers as the host conversion -- no cast kernel, no float copy of the batch),
           embeddings back into pinned memory with a non-blocking copy; the host collects a batch only when ``pending``
           newer ones are already queued behind it.

Memory is bounded by one window of decoded utterances and ``pending + stage_ahead + 1`` staging buffers, whatever the corpus.
On a CPU device (tests of the plumbing) the same code runs without streams.
"""
import collections
import concurrent.futures
import io
import itertools
import os
import struct
import subprocess

import numpy
import torch


def parse_wav(data, name="<bytes>"):
    """RIFF/WAVE bytes -> (samples, sample_rate): int16 array for 16-bit PCM (converted on the device later), float32 in
    [-1, 1) for 8 / 32-bit PCM and IEEE float -- the conversions ``bin/extract_xvectors.prepare`` applies.  Mono only."""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        return _scipy_wav(data, name)
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = pos + 8
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", data, body)
        elif cid == b"data":
            payload = memoryview(data)[body:min(body + size, len(data))]   # a streamed header may carry size 0xFFFFFFFF
            break
        pos = body + size + (size & 1)
    if fmt is None or payload is None:
        return _scipy_wav(data, name)
    tag, channels, rate, _, _, bits = fmt
    if channels != 1:
        raise IOError(f"{name}: expected a mono file, got {channels} channels")
    if tag == 1 and bits == 16:
        return numpy.frombuffer(payload, dtype="<i2", count=len(payload) // 2), rate
    if tag == 1 and bits == 32:
        return numpy.frombuffer(payload, dtype="<i4", count=len(payload) // 4).astype(numpy.float32) / 2147483648.0, rate
    if tag == 1 and bits == 8:
        return (numpy.frombuffer(payload, dtype=numpy.uint8).astype(numpy.float32) - 128.0) / 128.0, rate
    if tag == 3 and bits == 32:
        return numpy.frombuffer(payload, dtype="<f4", count=len(payload) // 4), rate
    return _scipy_wav(data, name)


def _scipy_wav(data, name):
    import scipy.io.wavfile
    try:
        rate, sample = scipy.io.wavfile.read(io.BytesIO(bytes(data)))
    except Exception as e:
        raise IOError(f"Error processing wav file: {name}\n{e}")
    if sample.ndim > 1:
        raise IOError(f"{name}: expected a mono file, got shape {sample.shape}")
    if sample.dtype == numpy.int32:
        sample = sample.astype(numpy.float32) / 2147483648.0
    elif sample.dtype == numpy.uint8:
        sample = (sample.astype(numpy.float32) - 128.0) / 128.0
    elif sample.dtype != numpy.int16:
        sample = numpy.ascontiguousarray(sample, dtype=numpy.float32)
    return sample, rate


def load_entry(source):
    """One wav.scp right-hand side (a path, or ``cmd |`` run through the shell as ``extract_xvectors.py:57-70`` does)."""
    source = source.strip()
    if source.endswith("|"):
        try:
            data = subprocess.run(source[:-1], shell=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        except Exception as e:
            raise IOError(f"Error processing wav file: {source}\n{e}")
    else:
        with open(source, "rb") as f:
            data = f.read()
    return parse_wav(data, source)


def plan_batches(lengths, batch_size, max_samples=None):
    """Indices of ``lengths`` sorted by length, cut into batches of at most ``batch_size`` utterances and (``max_samples``) at
    most that many PADDED samples (rows x longest row): bounded padding, and a window of hour-long files does not ask for a
    ``batch_size`` x hour staging buffer and activation workspace."""
    order = sorted(range(len(lengths)), key=lambda i: lengths[i])
    batches, cur = [], []
    for i in order:
        if cur and (len(cur) >= max(1, batch_size) or (max_samples and (len(cur) + 1) * lengths[i] > max_samples)):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def host_workers(cap=8):
    """Decode / staging threads of ONE rank: the cores this process may run on, divided by the ranks that share the host
    (``LOCAL_WORLD_SIZE`` under torch.distributed.run), at most ``cap`` -- eight ranks with eight threads each on a 64-core share would
    otherwise oversubscribe it 1:1 before the interpreter, the RCCL proxy and the HIP runtime threads get anything."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    return max(1, min(cap, cores // local_world))


class _Staging:
    """One slot of the pinned ring: int16 and float32 host buffers (grown on demand, never shrunk), the device copies,
    and the events that order copy -> compute -> read-back."""

    def __init__(self, device):
        self.device = device
        self.cuda = device.type == "cuda"
        self.host = {}
        self.dev = {}
        self.out_host = None
        self.copied = torch.cuda.Event() if self.cuda else None
        self.done = torch.cuda.Event() if self.cuda else None

    def buffers(self, dtype, rows, cols, copy_stream=None):
        """(rows, cols) views of the slot's flat pinned / device buffers: CONTIGUOUS on both sides, so the host -> device copy is
        one asynchronous DMA (a strided 2-D copy from pinned memory goes through a synchronous staging pass in torch)."""
        need = rows * cols
        h = self.host.get(dtype)
        if h is None or h.numel() < need:
            need_ = max(need, 0 if h is None else h.numel())
            h = torch.empty(need_, dtype=dtype, pin_memory=self.cuda)
            self.host[dtype] = h
            if self.cuda:
                # The caching allocator may hand out memory that a forward still queued on the compute stream is reading
                # (a freed temporary is reusable in COMPUTE-stream order only), and the copy stream is about to write it:
                # order the copy stream behind everything queued so far.  Happens only while the buffers grow.
                d = torch.empty(need_, dtype=dtype, device=self.device)
                copy_stream.wait_stream(torch.cuda.current_stream(self.device))
                d.record_stream(copy_stream)
                self.dev[dtype] = d
            else:
                self.dev[dtype] = h
        return self.host[dtype][:need].view(rows, cols), self.dev[dtype][:need].view(rows, cols)

    def out(self, rows, cols):
        if self.out_host is None or self.out_host.shape[0] < rows or self.out_host.shape[1] != cols:
            self.out_host = torch.empty((rows, cols), dtype=torch.float32, pin_memory=self.cuda)
        return self.out_host


def probe_wavs(paths, threads=8):
    """Native header walk of many files at once (``sk_wav_probe``): arrays ``kind`` (1 = PCM16 mono, 0 = other wav, -1 = cannot
    open), ``nsamples``, ``rate``, ``data_offset``."""
    import ctypes
    from . import _lib
    n = len(paths)
    kind, ns, rate = (numpy.zeros(n, dtype=numpy.int32) for _ in range(3))
    off = numpy.zeros(n, dtype=numpy.int64)
    if n:
        arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        _lib.check(_lib.lib().sk_wav_probe(arr, n, threads, ns.ctypes.data, rate.ctypes.data, off.ctypes.data, kind.ctypes.data))
    return kind, ns, rate, off


def read_pcm16(paths, offsets, nsamples, rows, dst, threads=8):
    """Native read of PCM16 payloads straight into rows of ``dst`` (2-D int16 numpy view of a pinned buffer)."""
    import ctypes
    from . import _lib
    n = len(paths)
    if not n:
        return
    assert dst.dtype == numpy.int16 and dst.ndim == 2 and dst.strides[1] == 2
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    off = numpy.ascontiguousarray(offsets, dtype=numpy.int64)
    ns = numpy.ascontiguousarray(nsamples, dtype=numpy.int32)
    rw = numpy.ascontiguousarray(rows, dtype=numpy.int32)
    status = numpy.zeros(n, dtype=numpy.int32)
    _lib.check(_lib.lib().sk_wav_read_pcm16(arr, off.ctypes.data, ns.ctypes.data, rw.ctypes.data, n, threads, dst.ctypes.data,
                                            dst.strides[0] // 2, dst.shape[0], status.ctypes.data))
    bad = numpy.nonzero(status)[0]
    if bad.size:
        raise IOError(f"Error processing wav file: {paths[int(bad[0])]} (short read, or row / length outside the staging buffer)")


class _Item:
    """One utterance of a window: either decoded samples or a PCM16 file the staging threads read in place."""
    __slots__ = ("key", "length", "samples", "path", "offset")

    def __init__(self, key, length, samples=None, path=None, offset=0):
        self.key, self.length, self.samples, self.path, self.offset = key, int(length), samples, path, int(offset)

    @property
    def int16(self):
        return self.path is not None or self.samples.dtype == numpy.int16


class StreamingExtractor:
    """``for key, vec in StreamingExtractor(model).run(entries)``: ``entries`` yields ``(key, source)`` with ``source`` a path /
    ``cmd |`` string, an already decoded 1-D array or a callable returning one; ``vec`` is the ``(1, E)`` float32 embedding.  Results arrive batch by
    batch (length-sorted inside a window of ``window * batch_size`` utterances), not in input order."""

    def __init__(self, model, batch_size=256, window=8, workers=None, pending=3, sample_rate=16000, norm_embedding=True, stage_ahead=6,
                 max_samples_per_batch=1 << 26):
        # pending / stage_ahead: batches waiting for their read-back / staged ahead of the launches.  Round 6 (scripts/pipeline_bench.py, 32 768 files, 8 decode
        # workers, three alternating runs each, k files/s): 2 / 2 (the default up to round 5) 40.3 / 38.8 / 40.8; 2 / 3 41.3 / 41.3 / 41.0; 3 / 6 41.6 / 41.7 / 41.3;
        # 4 / 8 41.1 / 40.9 / 40.9 -- a deeper queue rides out the host's jitter (a staging slot is one pinned + one device buffer of a batch of int16 samples)
        self.model = model
        self.batch_size = max(1, int(batch_size))
        self.window = max(1, int(window))
        self.workers = host_workers() if workers is None else max(1, int(workers))   # None: this rank's share of the host cores
        self.pending = max(1, int(pending))
        self.stage_ahead = max(1, int(stage_ahead))
        self.max_samples_per_batch = max_samples_per_batch   # 2^26 padded samples = 1024 x 4 s: caps staging and activation memory
        self.sample_rate = sample_rate
        self.norm_embedding = norm_embedding
        self.device = torch.device(model.device)
        self.cuda = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.pipelined = self.cuda and hasattr(model, "submit")   # whole batches in flight (Xtractor.submit / collect)
        # staging slots: the batches whose forwards are in flight beyond the newest one, those waiting for their read-back, those staged ahead, + 1
        in_flight = max(1, getattr(model, "pipeline_depth", 2) - 1) if self.pipelined else 1
        self.ring = [_Staging(self.device) for _ in range(self.pending + self.stage_ahead + in_flight)]
        self.stats = {"utterances": 0, "batches": 0, "samples": 0, "padded_samples": 0, "native_reads": 0}

    def _resample(self, sample, rate):
        """A file at another rate: the GPU resampler (``sk_resample``), result back on the host as float32 for the staging path.
        Rare by construction (a corpus is normally at the model's rate), so the round trip is not optimised."""
        if not self.cuda:
            raise ValueError(f"sample rate {rate} != {self.sample_rate}: resampling runs on the GPU (no CPU fallback)")
        from .resample import resample
        self.stats["resampled"] = self.stats.get("resampled", 0) + 1
        return resample(sample, rate, self.sample_rate, self.device).cpu().numpy()

    # ---- decode (whatever is not a canonical PCM16 file) ----------------------------------------------------------------
    def _decode(self, key, source):
        if isinstance(source, str):
            sample, rate = load_entry(source)
            if rate != self.sample_rate:       # extract_xvectors.py:141-143: Resample(orig_freq=sr, new_freq=sample_rate), on the device
                sample = self._resample(sample, rate)
        else:
            if callable(source):                       # a deferred decode (e.g. an IdMap row with start / stop), run on the pool
                source = source()
            sample = source.detach().cpu().numpy() if torch.is_tensor(source) else numpy.asarray(source)
        if sample.dtype != numpy.int16:
            sample = numpy.ascontiguousarray(sample, dtype=numpy.float32)
        if sample.ndim != 1:
            raise IOError(f"{key}: expected a mono signal, got shape {sample.shape}")
        return _Item(key, sample.shape[0], samples=sample)

    def _window_items(self, chunk, pool):
        """A window of (key, source) -> _Item list: plain paths are probed natively (header only), the rest is decoded."""
        items = [None] * len(chunk)
        plain = [i for i, (_, src) in enumerate(chunk) if isinstance(src, str) and not src.strip().endswith("|")]
        kind, ns, rate, off = probe_wavs([chunk[i][1].strip() for i in plain], self.workers)
        futures = {}
        for j, i in enumerate(plain):
            key, src = chunk[i]
            if kind[j] == 1 and rate[j] == self.sample_rate:        # any other rate goes through the decode pool, which resamples
                items[i] = _Item(key, ns[j], path=src.strip(), offset=off[j])
        for i, (key, src) in enumerate(chunk):
            if items[i] is None:
                futures[i] = pool.submit(self._decode, key, src)
        for i, f in futures.items():
            items[i] = f.result()
        return items

    # ---- staging (a background thread: file reads and copies into the slot's pinned buffer) ----------------------------------
    def _stage(self, slot, items, host, dev, as_int16):
        hv = host.numpy()
        fast = [(r, it) for r, it in enumerate(items) if it.path is not None]
        if as_int16:
            read_pcm16([it.path for _, it in fast], [it.offset for _, it in fast], [it.length for _, it in fast], [r for r, _ in fast],
                       hv, self.workers)
            for r, it in enumerate(items):            # padding keeps whatever the slot held before: rows are cut at their length
                if it.path is None:
                    hv[r, :it.length] = it.samples
        else:                                         # a float file in the batch: everything is staged as float32
            for r, it in enumerate(items):
                s = it.samples if it.path is None else load_entry(it.path)[0]
                hv[r, :it.length] = s.astype(numpy.float32) / 32768.0 if s.dtype == numpy.int16 else s
        return slot, items, host, dev, as_int16, len(fast) if as_int16 else 0

    # ---- one staged batch through copy + compute -----------------------------------------------------------------------------
    def _launch(self, slot, items, host, dev, as_int16, n_native):
        lens = [it.length for it in items]
        rows, cols = len(items), max(lens)
        if self.cuda:
            compute = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(slot.done)          # the slot's previous batch has left the device buffer
                dev.copy_(host, non_blocking=True)
                slot.copied.record(self.copy_stream)
            compute.wait_event(slot.copied)
        self.stats["utterances"] += rows
        self.stats["batches"] += 1
        self.stats["samples"] += sum(lens)
        self.stats["padded_samples"] += rows * cols
        self.stats["native_reads"] += n_native
        keys = [it.key for it in items]
        with torch.no_grad():                                   # int16 rows go in as they are: the front-end kernel widens them in its load
            if self.pipelined:
                # the batch's whole forward is queued on one of the handle's slot streams and NOT waited for: the next batch is submitted
                # before this one is collected, so two batches are in flight, half a step apart (Xtractor.submit)
                return slot, keys, self.model.submit(dev, lengths=lens, norm_embedding=self.norm_embedding), rows
            out = self.model(dev, is_eval=True, norm_embedding=self.norm_embedding, lengths=lens)
        return self._read_back(slot, keys, out, rows)

    def _finish(self, slot, keys, ticket, rows):
        """Second half of a pipelined launch: the compute stream waits for the batch's forward, then queues the read-back."""
        return self._read_back(slot, keys, self.model.collect(ticket), rows)

    def _read_back(self, slot, keys, out, rows):
        emb = out[1] if isinstance(out, tuple) else out
        oh = slot.out(max(rows, self.batch_size), emb.shape[1])
        oh[:rows].copy_(emb, non_blocking=self.cuda)
        if self.cuda:
            slot.done.record(torch.cuda.current_stream(self.device))
        return slot, keys, oh, rows

    def _collect(self, slot, keys, oh, rows):
        if self.cuda:
            slot.done.synchronize()
        vecs = oh[:rows].numpy().copy()
        for r, k in enumerate(keys):
            yield k, vecs[r:r + 1]

    # ---- the loop ------------------------------------------------------------------------------------------------------------
    def run(self, entries):
        entries = iter(entries)
        span = self.window * self.batch_size
        free = collections.deque(self.ring)
        staged = collections.deque()          # futures of _stage, submission order
        launched = collections.deque()        # (slot, keys, out_host, rows), launch order
        if self.cuda:
            for s in self.ring:
                s.done.record(torch.cuda.current_stream(self.device))   # "previous batch" of a fresh slot

        inflight = collections.deque()        # pipelined: (slot, keys, ticket, rows) submitted, forward not yet waited for
        depth = getattr(self.model, "pipeline_depth", 2) if self.pipelined else 1

        def finish_oldest():
            launched.append(self._finish(*inflight.popleft()))

        def launch_oldest():
            entry = self._launch(*staged.popleft().result())
            if not self.pipelined:
                launched.append(entry)
                return
            inflight.append(entry)
            while len(inflight) >= depth:       # submit(k), then collect(k - depth + 1): `depth` forwards were in flight meanwhile
                finish_oldest()

        def collect_oldest():
            if not launched:
                finish_oldest()
            slot, keys, oh, rows = launched.popleft()
            yield from self._collect(slot, keys, oh, rows)
            free.append(slot)

        try:
            yield from self._run(entries, span, free, staged, launched, inflight, launch_oldest, collect_oldest)
        finally:
            if self.pipelined and inflight:       # abandoned half way (an exception, or the consumer stopped iterating): leave no ticket behind
                inflight.clear()
                self.model.discard_pending()

    def _run(self, entries, span, free, staged, launched, inflight, launch_oldest, collect_oldest):
        with concurrent.futures.ThreadPoolExecutor(self.workers) as pool, concurrent.futures.ThreadPoolExecutor(self.stage_ahead) as stager:
            while True:
                chunk = list(itertools.islice(entries, span))
                if not chunk:
                    break
                items = self._window_items(chunk, pool)
                for idx in plan_batches([it.length for it in items], self.batch_size, self.max_samples_per_batch):
                    batch = [items[i] for i in idx]
                    while not free:                             # every slot is staged, in flight or waiting to be read back
                        if launched or inflight:
                            yield from collect_oldest()
                        else:
                            launch_oldest()
                    slot = free.popleft()
                    as_int16 = all(it.int16 for it in batch)
                    dtype = torch.int16 if as_int16 else torch.float32
                    host, dev = slot.buffers(dtype, len(batch), max(it.length for it in batch), self.copy_stream)
                    staged.append(stager.submit(self._stage, slot, batch, host, dev, as_int16))
                    while staged and (staged[0].done() or len(staged) > self.stage_ahead):
                        launch_oldest()
                        while len(launched) + len(inflight) > self.pending:
                            yield from collect_oldest()
            while staged:
                launch_oldest()
            while launched or inflight:
                yield from collect_oldest()
