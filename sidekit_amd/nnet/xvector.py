"""``Xtractor`` -- host-side mirror of ``sidekit.nnet.xvector.Xtractor`` over the HIP library.

Keeps the constructor / ``forward`` / ``load_state_dict`` / ``to`` / ``eval`` surface that
``sidekit/bin/extract_xvectors.py:74-89,146`` and ``sidekit/nnet/xvector.py:1835-1843,1893`` use
(reference ctor ``sidekit/nnet/xvector.py:424-431``, forward ``:876-907``), for the two
architectures in scope (``model_archi='halfresnet34'`` and ``'xvector'``).  No ``torch.nn`` is
involved: parameters are plain tensors kept in checkpoint layout, compute is
``libsidekit_amd.so`` through ctypes, torch tensors only carry the input/output device memory
and the stream.  There is no CPU fallback: running on a non-GPU device raises.
"""
import ctypes
import os
from collections import OrderedDict
from types import SimpleNamespace

import numpy
import torch

from .. import _lib
from .preprocessor import MelSpecFrontEnd, MfccFrontEnd
from .weights import seeded_state_dict, state_dict_spec


def _ptr(arr):
    return None if arr is None else arr.ctypes.data


class _Params(SimpleNamespace):
    """Attribute view on a group of checkpoint tensors (e.g. ``model.after_speaker_embedding.weight``)."""


class Xtractor:
    """x-vector extractor (inference).  Same positional/keyword arguments as the reference."""

    def __init__(self, speaker_number, model_archi="xvector", loss=None, norm_embedding=False, aam_margin=0.2, aam_s=30,
                 embedding_size=256, seed=None):
        self.speaker_number = speaker_number
        self.feature_size = None
        self.norm_embedding = norm_embedding
        self.embedding_size = embedding_size
        print(f"MODEL = {model_archi}")  # reference ctor prints this (xvector.py:451)
        if model_archi == "xvector":
            if loss not in ["cce", "aam"]:
                raise NotImplementedError("The valid loss are for now cce and aam ")
            self.loss = loss
            self.input_nbdim = 2
            self.preprocessor = MfccFrontEnd(self)
            self.feature_size = self.preprocessor.n_mfcc
            self._arch, self._aam_s = _lib.XT_ARCH_TDNN, 64.0  # xvector.py:493-497
        elif model_archi == "halfresnet34":
            if loss != "aam":
                raise NotImplementedError("halfresnet34 is built for loss='aam' (the released checkpoints); "
                                          f"got loss={loss!r}")
            self.loss = loss
            self.preprocessor = MelSpecFrontEnd(self)
            self._arch, self._aam_s = _lib.XT_ARCH_HALFRESNET34, 30.0  # xvector.py:586-591 (s=30, m=0.2 hard-coded)
        else:
            raise NotImplementedError(f"model_archi={model_archi!r}: only 'halfresnet34' and 'xvector' are built")
        self.model_archi = model_archi
        self.aam_margin, self.aam_s = aam_margin, self._aam_s
        self._spec = state_dict_spec(model_archi, speaker_number, embedding_size, self.loss)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self._sd = seeded_state_dict(model_archi, speaker_number, embedding_size, self.loss, seed)
        self.device = torch.device("cpu")
        self.training = True
        self.compute_dtype = None  # None: follow torch autocast (reduced precision -> bf16 trunk), 'fp32' or 'bf16'
        self._handles = {}
        self._reserved = {}
        self._slot_shapes, self._tickets, self._next_slot = {}, [], 0     # pipelined forwards (submit / collect)
        self._refresh_views()

    # ---- torch.nn.Module look-alikes -------------------------------------------------------------
    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._spec if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._spec]
        errors = []
        if strict and unexpected:
            errors.append("Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
        if strict and missing:
            errors.append("Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
        for k, (shape, _) in self._spec.items():
            if k in state_dict and tuple(state_dict[k].shape) != tuple(shape):
                errors.append(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)} from checkpoint, "
                              f"the shape in current model is {tuple(shape)}.")
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for Xtractor:\n\t" + "\n\t".join(errors))
        for k in self._spec:
            if k in state_dict:
                v = state_dict[k].detach().to("cpu")
                self._sd[k] = v.to(torch.int64 if k.endswith("num_batches_tracked") else torch.float32).contiguous().clone()
        self._drop_handles()
        self._refresh_views()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self.device:
            self._drop_handles()
            self.device = device
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is out of scope for sidekit_amd (inference-only hot path)")
        return self.eval()

    def parameters(self):
        return [v for k, v in self._sd.items() if self._spec[k][1] not in ("buffer", "count", "bn_m", "bn_v")]

    def context_size(self):
        """xvector.py:909-924: frames of context the conv stack named ``conv*`` consumes."""
        if self.model_archi == "xvector":
            return 1 + 4 * 1 + 2 * 2 + 2 * 3
        return 1 + 2  # only `sequence_network.conv1` (the 3x3 stem) matches name.startswith("conv")

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    # ---- compute ------------------------------------------------------------------------------
    def forward(self, x, is_eval=False, target=None, norm_embedding=True, lengths=None):
        """Reference ``forward`` (xvector.py:876-907) for extraction.

        :param x: float32 waveform ``(L,)`` or ``(B, L)`` on the model's device -- or int16 PCM as the files hold it, which the
            front-end kernel widens in its load (``xt_forward_pcm16``: the same numbers as ``x.float() / 32768``, no cast kernel)
        :param lengths: optional per-utterance sample counts for a zero-padded batch; each row is then
            processed exactly as if run alone (no padding semantics exist in the reference)
        :return: ``(s*cos logits (B, n_spk), x-vectors (B, E))`` for ``loss='aam'``; x-vectors for ``'cce'``
        """
        if not is_eval or target is not None:
            raise NotImplementedError("sidekit_amd.Xtractor runs extraction only: call with is_eval=True and no target")
        x = self._check_input(x, pcm16_ok=True)
        B, L = x.shape
        h = self._handle()
        self._reserve(h, B, L)
        lib = _lib.lib()
        _lib.check(lib.xt_set_norm_embedding(h, 1 if norm_embedding else 0))
        emb = torch.empty((B, self.embedding_size), dtype=torch.float32, device=x.device)
        logits = torch.empty((B, int(self.speaker_number)), dtype=torch.float32, device=x.device) if self.loss == "aam" else None
        lens = self._lengths(lengths, B, L)
        entry = lib.xt_forward_pcm16 if x.dtype == torch.int16 else lib.xt_forward
        _lib.check(entry(h, x.data_ptr(), x.stride(0) if B > 1 else L, _ptr(lens), B, L, emb.data_ptr(),
                         logits.data_ptr() if logits is not None else None, self._stream(x)))
        return (logits, emb) if self.loss == "aam" else emb

    # ---- pipelined forwards: two whole batches in flight (xt_forward_begin / xt_forward_end) ---------------------------------------
    @staticmethod
    def _depth_from_env():
        try:                                  # a malformed value must not break importing the package
            return min(4, max(1, int(os.environ.get("SIDEKIT_AMD_PIPELINE_DEPTH", "2"))))
        except ValueError:
            return 2

    # batches in flight.  Two is where the gain is (5.80 -> 5.56 ms per batch of 256, round 4).  A third adds 0.3-1.1 % on resident inputs (round 6, alternating
    # runs on two boxes: 46.12 / 46.14 -> 46.49 / 46.63 k and 45.24 / 45.39 -> 45.74 / 45.52 k x-vectors/s) and COSTS the streaming extractor 13 % (38.5 / 39.6 ->
    # 34.4 / 33.6 k files/s: its copy stream and read-backs then share the device with three forwards); four in flight lose 8 % in bench.py itself (42.4-42.5 k);
    # GPU_MAX_HW_QUEUES=8 changes none of it.  profiles/r06_pipeline_depth.txt
    pipeline_depth = _depth_from_env.__func__()
    # Workspaces are sized for a batch's longest utterance rounded UP to a whole second of samples: a corpus streams batches whose maxima creep up by a few
    # samples at a time, and every new record otherwise costs a device synchronisation + the reallocation of every buffer of every slot in the middle of the
    # run (round 6: 16 384 files of 3-5 s: 37-39 k -> see profiles/r06_pipeline_bench.json).  At most one second of activations per utterance more memory.
    reserve_round = 16000

    def _round_up(self, L):
        r = max(1, int(self.reserve_round))
        return -(-int(L) // r) * r

    def submit(self, x, lengths=None, norm_embedding=True):
        """Queue ``forward(x, is_eval=True)`` WITHOUT waiting for it on the caller's stream and return a ticket for :meth:`collect`.

        The reference driver runs one forward at a time (``extract_xvectors.py:130-150``); a corpus is many independent batches, and keeping
        ``pipeline_depth`` (2) of them in flight -- each on a stream the handle owns, half a step apart -- uses the chip better than the two
        halves of one batch side by side (5.67 vs 5.87 ms per batch of 256 in one run of bench.py, profiles/r05_bench_line.json).  Tickets must be collected in submission order; at most
        ``pipeline_depth`` may be outstanding.  The x-vectors are the bits ``forward`` returns."""
        x = self._check_input(x, pcm16_ok=True)
        B, L = x.shape
        h = self._handle()
        key = next(k for k, v in self._handles.items() if v is h)
        lib = _lib.lib()
        shapes = self._slot_shapes.setdefault(key, [])
        if not any(b >= B and l >= L for b, l in shapes):
            Lr = self._round_up(L)
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                _lib.check(lib.xt_reserve_slots(h, self.pipeline_depth, B, Lr))
            shapes[:] = [(b, l) for b, l in shapes if not (b <= B and l <= Lr)] + [(B, Lr)]
            self._reserved[key] = [(b, l) for b, l in self._reserved[key] if not (b <= B and l <= Lr)] + [(B, Lr)]
        if len(self._tickets) >= self.pipeline_depth:
            raise RuntimeError(f"submit: {self.pipeline_depth} batches are already in flight -- collect() the oldest first")
        _lib.check(lib.xt_set_norm_embedding(h, 1 if norm_embedding else 0))
        emb = torch.empty((B, self.embedding_size), dtype=torch.float32, device=x.device)
        logits = torch.empty((B, int(self.speaker_number)), dtype=torch.float32, device=x.device) if self.loss == "aam" else None
        lens = self._lengths(lengths, B, L)
        slot = self._next_slot
        rc = lib.xt_forward_begin(h, slot, x.data_ptr(), _lib.XT_I16 if x.dtype == torch.int16 else _lib.XT_F32,
                                  x.stride(0) if B > 1 else L, _ptr(lens), B, L, emb.data_ptr(),
                                  logits.data_ptr() if logits is not None else None, self._stream(x))
        if rc != _lib.SK_OK:
            # part of the forward may already be queued on the slot's stream (the library records the slot's completion event on its error
            # paths too): order the caller's stream behind it BEFORE emb / logits / x go back to the caching allocator with this frame
            msg = _lib.last_error()
            lib.xt_forward_end(h, slot, self._stream(x))
            raise (ValueError if rc == _lib.SK_EARG else RuntimeError)(msg)
        self._next_slot = (slot + 1) % self.pipeline_depth
        # x is kept alive until the forward that reads it has been waited for; the submitting stream is kept because emb / logits were taken from ITS pool
        ticket = (h, slot, x, logits, emb, torch.cuda.current_stream(x.device))
        self._tickets.append(ticket)
        return ticket

    def collect(self, ticket):
        """Make the current stream wait for the oldest submitted batch and return what ``forward`` returns for it."""
        if not self._tickets or self._tickets[0] is not ticket:
            raise RuntimeError("collect: tickets are collected in submission order")
        h, slot, x, logits, emb, sub = self._tickets.pop(0)
        lib = _lib.lib()
        cur = torch.cuda.current_stream(emb.device)
        _lib.check(lib.xt_forward_end(h, slot, ctypes.c_void_p(cur.cuda_stream)))
        if cur != sub:
            # Collected on another stream than the one it was submitted on.  emb / logits (and, usually, x) are blocks of the SUBMITTING stream's
            # pool: once the caller drops them the caching allocator hands them out again on that stream at once -- which so far knows neither
            # of the slot's forward (still writing emb, still reading x) nor of what the collecting stream does with the results.  So the
            # submitting stream waits for the slot as well (one hipStreamWaitEvent), and the outputs are recorded as in use on this stream.
            _lib.check(lib.xt_forward_end(h, slot, ctypes.c_void_p(sub.cuda_stream)))
            emb.record_stream(cur)
            if logits is not None:
                logits.record_stream(cur)
        return (logits, emb) if self.loss == "aam" else emb

    def discard_pending(self):
        """Collect (and drop) every outstanding ticket: the state after this is that of a model nothing was submitted to.  For callers that
        abandon a pipelined loop half way (an exception between ``submit`` and ``collect``)."""
        while self._tickets:
            self.collect(self._tickets[0])

    def forward_features(self, feats, frames=None, norm_embedding=True):
        """Everything after ``xvector.py:885``: ``feats`` is the ``(B, 80, T)`` front-end output."""
        feats = self._check_input(feats, dims=3)
        B, F, T = feats.shape
        if F != 80:
            raise RuntimeError(f"expected (B, 80, T) features, got {tuple(feats.shape)}")
        h = self._handle()
        self._reserve(h, B, (T - 1) * self.preprocessor.hop_length + self.preprocessor.n_fft)
        lib = _lib.lib()
        _lib.check(lib.xt_set_norm_embedding(h, 1 if norm_embedding else 0))
        emb = torch.empty((B, self.embedding_size), dtype=torch.float32, device=feats.device)
        logits = torch.empty((B, int(self.speaker_number)), dtype=torch.float32, device=feats.device) if self.loss == "aam" else None
        lens = self._lengths(frames, B, T)
        _lib.check(lib.xt_forward_features(h, feats.data_ptr(), _ptr(lens), B, T, emb.data_ptr(),
                                           logits.data_ptr() if logits is not None else None, self._stream(feats)))
        return (logits, emb) if self.loss == "aam" else emb

    def features(self, x, lengths=None):
        """Front-end only (``MelSpecFrontEnd.forward(is_eval=True)`` / ``MfccFrontEnd.forward``): ``(B, 80, T)``."""
        x = self._check_input(x)
        B, L = x.shape
        h = self._handle("fp32")
        self._reserve(h, B, L)
        T = 1 + L // self.preprocessor.hop_length
        out = torch.empty((B, 80, T), dtype=torch.float32, device=x.device)
        lens = self._lengths(lengths, B, L)
        _lib.check(_lib.lib().xt_features(h, x.data_ptr(), x.stride(0) if B > 1 else L, _ptr(lens), B, L, out.data_ptr(), self._stream(x)))
        return out

    def debug_taps(self, names, dtype=None):
        """Intermediate activations of the last forward (after ``set_debug(True)``), as numpy arrays."""
        h = self._handle(dtype)
        lib = _lib.lib()
        out = {}
        for n in names:
            nbytes = ctypes.c_size_t(0)
            _lib.check(lib.xt_debug_tap(h, n.encode(), None, 0, ctypes.byref(nbytes)))
            buf = numpy.empty(nbytes.value, dtype=numpy.uint8)
            _lib.check(lib.xt_debug_tap(h, n.encode(), buf.ctypes.data, buf.nbytes, ctypes.byref(nbytes)))
            out[n] = buf
        return out

    def set_profile(self, on, dtype=None, slots=None):
        """Bracket the kernel launches of ``forward`` with HIP events on the launch stream (measurement only): all of
        them, or only the named classes (``slots``, names as returned by ``get_profile``) -- an event pair costs stream time."""
        mask = 1 if on else 0
        if on and slots is not None:
            mask = 0
            for name in slots:
                mask |= 1 << (_lib.PROF_NAMES.index(name) + 1)
        _lib.check(_lib.lib().xt_set_profile(self._handle(dtype), mask))

    def get_profile(self, dtype=None, reset=True):
        """``{kernel class: (device ms, launches)}`` accumulated since the last reset."""
        ms = (ctypes.c_double * _lib.XT_PROF_SLOTS)()
        n = (ctypes.c_int64 * _lib.XT_PROF_SLOTS)()
        _lib.check(_lib.lib().xt_get_profile(self._handle(dtype), ms, n, 1 if reset else 0))
        return {name: (ms[i], n[i]) for i, name in enumerate(_lib.PROF_NAMES) if n[i]}

    def set_lanes(self, lanes, dtype=None):
        """1: serial forward, 2: a batch of >= 128 utterances runs as two halves on two HIP streams (``xt_set_lanes``; same bits)."""
        _lib.check(_lib.lib().xt_set_lanes(self._handle(dtype), int(lanes)))

    def get_lanes(self, dtype=None):
        return int(_lib.lib().xt_get_lanes(self._handle(dtype)))

    def set_debug(self, on, dtype=None):
        _lib.check(_lib.lib().xt_set_debug(self._handle(dtype), 1 if on else 0))

    # ---- plumbing ----------------------------------------------------------------------------
    def _refresh_views(self):
        groups = {}
        for k, v in self._sd.items():
            top, _, rest = k.partition(".")
            groups.setdefault(top, {})[rest.replace(".", "_")] = v
        for top in ("sequence_network", "stat_pooling", "before_speaker_embedding", "after_speaker_embedding"):
            setattr(self, top, _Params(**groups.get(top, {})))

    def _check_input(self, x, dims=2, pcm16_ok=False):
        if not torch.is_tensor(x):
            raise TypeError("input must be a torch tensor")
        if self.device.type != "cuda":
            raise RuntimeError("sidekit_amd.Xtractor computes on the GPU only (no CPU fallback): call .to('cuda') first")
        if x.dim() == dims - 1:
            x = x.unsqueeze(0)
        if x.dim() != dims:
            raise RuntimeError(f"expected a {dims - 1}-D or {dims}-D input, got shape {tuple(x.shape)}")
        if x.device != self.device:
            raise RuntimeError(f"input is on {x.device} but the model is on {self.device}")
        if x.dtype != torch.float32 and not (pcm16_ok and x.dtype == torch.int16):
            x = x.float()
        if x.stride(-1) != 1 or (dims == 3 and not x.is_contiguous()):
            x = x.contiguous()
        return x

    @staticmethod
    def _lengths(lengths, B, limit):
        if lengths is None:
            return None
        arr = numpy.ascontiguousarray(torch.as_tensor(lengths).cpu().numpy().astype(numpy.int32))
        if arr.shape != (B,):
            raise ValueError(f"lengths must have shape ({B},)")
        return arr

    @staticmethod
    def _stream(t):
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def _dtype_key(self, dtype=None):
        d = dtype or self.compute_dtype
        if d is None:
            d = "bf16" if (torch.is_autocast_enabled() and self.model_archi == "halfresnet34") else "fp32"
        if d in ("bf16", torch.bfloat16):
            if self.model_archi != "halfresnet34":
                raise NotImplementedError("the TDNN runs in fp32 only")
            return "bf16"
        if d in ("fp32", "float32", torch.float32):
            return "fp32"
        raise ValueError(f"compute_dtype={d!r}: use 'fp32' or 'bf16'")

    def _handle(self, dtype=None):
        key = self._dtype_key(dtype)
        if key in self._handles:
            return self._handles[key]
        if self.device.type != "cuda":
            raise RuntimeError("sidekit_amd.Xtractor computes on the GPU only (no CPU fallback): call .to('cuda') first")
        lib = _lib.lib()
        cfg = _lib.XtConfig(self._arch, _lib.XT_BF16 if key == "bf16" else _lib.XT_F32,
                            _lib.XT_LOSS_AAM if self.loss == "aam" else _lib.XT_LOSS_CCE, int(self.speaker_number),
                            int(self.embedding_size), float(self._aam_s))
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.xt_create(ctypes.byref(cfg), ctypes.byref(h)))
            try:
                for k, v in self._sd.items():
                    a = v.numpy()
                    shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
                    _lib.check(lib.xt_set_tensor(h, k.encode(), a.ctypes.data, shape, a.ndim,
                                                 _lib.XT_I64 if a.dtype == numpy.int64 else _lib.XT_F32))
                _lib.check(lib.xt_finalize(h))
            except Exception:
                lib.xt_destroy(h)
                raise
        self._handles[key] = h
        self._reserved[key] = []
        return h

    def _reserve(self, h, B, L):
        """Size the workspace for a (B, L) batch unless a shape reserved earlier covers it in both dimensions: buffers grow to
        the largest B x L product actually requested, never to max(B) x max(L) of unrelated calls."""
        key = next(k for k, v in self._handles.items() if v is h)
        shapes = self._reserved[key]
        if any(b >= B and l >= L for b, l in shapes):
            return
        L = self._round_up(L)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _lib.check(_lib.lib().xt_reserve(h, B, L))
        self._reserved[key] = [(b, l) for b, l in shapes if not (b <= B and l <= L)] + [(B, L)]

    def _drop_handles(self):
        if self._handles:
            lib = _lib.lib()
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            for h in self._handles.values():
                lib.xt_destroy(h)
        self._handles, self._reserved = {}, {}
        self._slot_shapes, self._tickets, self._next_slot = {}, [], 0

    def __del__(self):
        try:
            self._drop_handles()
        except Exception:
            pass


# ---- library-level extraction driver --------------------------------------------------------------
def _load_segment(data_root_name, seg_id, file_extension, start_cs, stop_cs, sample_rate, min_duration):
    """One IdMap row -> (float32 waveform tensor, start sample, stop sample): the ``__getitem__`` of the reference's
    ``IdMapSet`` (``sidekit/nnet/xsets.py:420-466``; start / stop are centiseconds, ``None`` = whole file; a segment
    shorter than ``min_duration`` seconds is widened around its middle).  PCM wav files are read with ``scipy.io.wavfile``
    (torchaudio is not installed)."""
    import scipy.io.wavfile
    sr, x = scipy.io.wavfile.read(f"{data_root_name}/{seg_id}.{file_extension}")
    # xsets.py:434-448: a whole file at another rate is resampled, a start / stop segment of one asserts
    assert sr == sample_rate or stop_cs is None, f"{seg_id}: sample rate {sr} != {sample_rate} for a start / stop segment"
    if x.ndim > 1:
        x = x[:, 0]
    if x.dtype == numpy.int16:
        x = x.astype(numpy.float32) / 32768.0
    elif x.dtype == numpy.int32:
        x = x.astype(numpy.float32) / 2147483648.0
    x = numpy.ascontiguousarray(x, dtype=numpy.float32)
    if sr != sample_rate:
        from ..resample import resample
        x = resample(x, sr, sample_rate).cpu().numpy()
    start = 0 if start_cs is None else int(start_cs * 0.01 * sample_rate)
    if stop_cs is None:
        duration = int(x.shape[0] - start)          # the reference keeps the whole file in this branch
    else:
        duration = int(stop_cs * 0.01 * sample_rate) - start
        if duration <= min_duration * sample_rate:
            middle = start + duration // 2
            start = int(max(0, int(middle - (min_duration * sample_rate / 2))))
            duration = int(min_duration * sample_rate)
        x = x[start:start + duration]
    return torch.from_numpy(x), start, start + duration


def extract_embeddings(idmap_name, model_filename, data_root_name, device, batch_size=1, file_extension="wav", transform_pipeline={},
                       sliding_window=False, win_duration=3., win_shift=1.5, num_thread=1, sample_rate=16000, mixed_precision=False,
                       norm_embeddings=True):
    """``sidekit.nnet.xvector.extract_embeddings`` (``sidekit/nnet/xvector.py:1796-1916``): x-vectors of every segment
    of an IdMap, returned as a ``StatServer`` (``stat1`` = x-vectors, ``stat0`` = ones, ids as unicode arrays).

    Same arguments.  Differences on the driver side only: ``batch_size`` utterances go through one padded forward
    (every row computed over its own length), sliding windows of a file form one batch, ``mixed_precision`` selects the
    bf16 trunk (the reference's fp16 autocast), augmentation pipelines are training-time and refused; for sliding windows
    ``stop`` is ``start + window length`` in samples (the reference stores ``start + 1`` there, ``:1901-1911``)."""
    from ..bosaris import IdMap
    from ..statserver import StatServer
    if transform_pipeline:
        raise NotImplementedError("augmentation pipelines are training-time (out of scope)")
    if isinstance(model_filename, str):
        checkpoint = torch.load(model_filename, map_location="cpu", weights_only=False)
        model_opts = checkpoint["model_archi"]
        model = Xtractor(checkpoint["speaker_number"], model_archi=model_opts["model_type"], loss=model_opts["loss"]["type"],
                         embedding_size=256)
        model.load_state_dict(checkpoint["model_state_dict"])
    else:
        model = model_filename
    idmap = idmap_name if isinstance(idmap_name, IdMap) else IdMap(idmap_name)
    model.eval()
    model.to(device)
    prev_dtype = model.compute_dtype
    if mixed_precision and model.model_archi == "halfresnet34":
        model.compute_dtype = "bf16"
    win_len, win_hop = int(win_duration * sample_rate), int(win_shift * sample_rate)
    embed, modelset, segset, starts, stops = [], [], [], [], []

    def run(batch, lens):
        out = model(batch.to(model.device), is_eval=True, norm_embedding=norm_embeddings, lengths=lens)
        return (out[1] if isinstance(out, tuple) else out).detach().cpu()

    try:
        with torch.no_grad():
            if sliding_window:
                for i in range(idmap.leftids.shape[0]):
                    speech, start, _ = _load_segment(data_root_name, idmap.rightids[i], file_extension, idmap.start[i], idmap.stop[i],
                                                     sample_rate, win_duration)
                    windows = speech.unfold(0, win_len, win_hop)                      # (n_windows, win_len)
                    for j in range(0, windows.shape[0], max(1, 100)):
                        embed.append(run(windows[j:j + 100].contiguous(), None))
                    n = windows.shape[0]
                    modelset.extend([idmap.leftids[i]] * n)
                    segset.extend([idmap.rightids[i]] * n)
                    w0 = numpy.arange(n) * win_hop + start
                    starts.extend(w0.tolist())
                    stops.extend((w0 + win_len).tolist())
            else:
                # whole-file rows that are canonical PCM16 go file -> pinned staging natively; everything else (start / stop
                # windows, other sample formats) is decoded by _load_segment on the pool; batches of `batch_size` length-sorted
                # utterances stream through copy + forward while the next ones are staged (sidekit_amd/pipeline.py)
                from ..pipeline import StreamingExtractor, probe_wavs
                n = idmap.leftids.shape[0]
                workers = max(4, int(num_thread))
                path = [f"{data_root_name}/{idmap.rightids[i]}.{file_extension}" for i in range(n)]
                whole = [i for i in range(n) if idmap.start[i] is None and idmap.stop[i] is None]
                kind, ns, _, _ = probe_wavs([path[i] for i in whole], workers)
                span = {i: (0, int(ns[j])) for j, i in enumerate(whole) if kind[j] == 1}

                def deferred(i):
                    def load():
                        speech, a, b = _load_segment(data_root_name, idmap.rightids[i], file_extension, idmap.start[i], idmap.stop[i],
                                                     sample_rate, win_duration)
                        span[i] = (a, b)
                        return speech
                    return load

                stream = StreamingExtractor(model, batch_size=max(1, batch_size), workers=workers, sample_rate=sample_rate,
                                            norm_embedding=norm_embeddings)
                vec = [None] * n
                for i, v in stream.run((i, path[i] if i in span else deferred(i)) for i in range(n)):
                    vec[i] = torch.from_numpy(v)
                embed = vec
                modelset = list(idmap.leftids)
                segset = list(idmap.rightids)
                starts = [span[i][0] for i in range(n)]
                stops = [span[i][1] for i in range(n)]
    finally:
        model.compute_dtype = prev_dtype
    embeddings = StatServer()
    embeddings.stat1 = numpy.concatenate([e.numpy() for e in embed])
    embeddings.modelset = numpy.array(modelset).astype('>U')
    embeddings.segset = numpy.array(segset).astype('>U')
    embeddings.start = numpy.array(starts).squeeze()
    embeddings.stop = numpy.array(stops).squeeze()
    embeddings.stat0 = numpy.ones((embeddings.modelset.shape[0], 1))
    return embeddings


def _load_model(model_filename):
    """A checkpoint file name -> Xtractor (``extract_embeddings`` :1822-1829); an Xtractor is returned as is."""
    if isinstance(model_filename, str):
        checkpoint = torch.load(model_filename, map_location="cpu", weights_only=False)
        model_opts = checkpoint["model_archi"]
        model = Xtractor(checkpoint["speaker_number"], model_archi=model_opts["model_type"], loss=model_opts["loss"]["type"],
                         embedding_size=256)
        model.load_state_dict(checkpoint["model_state_dict"])
        return model
    return model_filename


def extract_embeddings_per_speaker(idmap_name, model_filename, data_root_name, device, file_extension="wav", transform_pipeline={},
                                   sample_rate=16000, mixed_precision=False, num_thread=1, dither=10e-6):
    """``sidekit.nnet.xvector.extract_embeddings_per_speaker`` (``sidekit/nnet/xvector.py:1919-1999``): ONE x-vector per
    speaker from the concatenation of all of the speaker's segments (``IdMapSetPerSpeaker``, ``xsets.py:483-590``,
    ``min_duration`` 1 s), truncated to 20 000 000 samples, always ``norm_embedding=True``.

    Same arguments plus ``dither``: the reference adds ``10e-6 * randn`` to every segment before concatenating
    (``xsets.py:572``); the same is done here from torch's global generator, ``dither=0`` makes the call deterministic.
    Speakers come out in ``numpy.unique`` order; ``start`` / ``stop`` are ``None`` per row (the reference leaves them as
    empty object arrays).  A whole-file segment is read from its OWN file (the reference indexes ``rightids`` by the
    speaker index there, ``xsets.py:558``, which reads the wrong file whenever ids are not aligned)."""
    from ..bosaris import IdMap
    from ..statserver import StatServer
    if transform_pipeline:
        raise NotImplementedError("augmentation pipelines are training-time (out of scope)")
    model = _load_model(model_filename)
    idmap = idmap_name if isinstance(idmap_name, IdMap) else IdMap(idmap_name)
    model.eval()
    model.to(device)
    prev_dtype = model.compute_dtype
    if mixed_precision and model.model_archi == "halfresnet34":
        model.compute_dtype = "bf16"
    speakers = numpy.unique(idmap.leftids)
    stat1 = numpy.ones((speakers.shape[0], model.embedding_size))
    try:
        with torch.no_grad():
            for idx, spk in enumerate(speakers):
                parts = []
                for sid, seg_id, seg_start, seg_stop in zip(idmap.leftids, idmap.rightids, idmap.start, idmap.stop):
                    if sid != spk:
                        continue
                    speech, _, _ = _load_segment(data_root_name, seg_id, file_extension, seg_start, seg_stop, sample_rate, 1.)
                    if dither:
                        speech = speech + dither * torch.randn(speech.shape)
                    parts.append(speech)
                data = torch.cat(parts)[:20000000].unsqueeze(0)
                out = model(data.to(model.device), is_eval=True, norm_embedding=True)
                stat1[idx, :] = (out[1] if isinstance(out, tuple) else out).detach().cpu().numpy()
    finally:
        model.compute_dtype = prev_dtype
    embeddings = StatServer()
    embeddings.modelset = speakers
    embeddings.segset = speakers
    embeddings.start = numpy.empty(speakers.shape[0], "|O")
    embeddings.stop = numpy.empty(speakers.shape[0], "|O")
    embeddings.stat0 = numpy.ones((speakers.shape[0], 1))
    embeddings.stat1 = stat1
    return embeddings


def test_metrics(model, device, model_opts, data_opts, train_opts, as_norm=True):
    """``sidekit.nnet.xvector.test_metrics`` (``sidekit/nnet/xvector.py:212-271``): x-vectors of the test IdMap, all-vs-all
    cosine, the trials of the Ndx, EER from ``rocch`` / ``rocch2eer``; with ``as_norm`` also the EER of the adaptive
    s-normalised scores (cohort = rows of ``after_speaker_embedding.weight``), returned as ``(eer, norm_eer)``.

    ``data_opts["test"]["ndx"]`` / ``["key"]`` may be ``Ndx`` / ``Key`` objects or file names (HDF5 as the reference writes
    it, or text trial lists).  The N x N cosine matrix and the normalisation run on the GPU."""
    from ..bosaris import Key, Ndx
    from ..bosaris.detplot import rocch, rocch2eer
    from ..iv_scoring import cosine_matrix
    from ..score_normalization import asnorm
    model = _load_model(model)
    xv_stat = extract_embeddings(idmap_name=data_opts["test"]["idmap"], model_filename=model,
                                 data_root_name=data_opts["test"]["data_path"], device=device, transform_pipeline={},
                                 num_thread=train_opts.get("num_cpu", 1), mixed_precision=train_opts.get("mixed_precision", False),
                                 batch_size=train_opts.get("batch_size", 1))

    def load(cls, x):
        if isinstance(x, cls):
            return x
        return cls.read_txt(x) if str(x).endswith(".txt") else cls(x)

    ndx, key = load(Ndx, data_opts["test"]["ndx"]), load(Key, data_opts["test"]["key"])
    from ..iv_scoring import normalize_rows_device
    tsr = normalize_rows_device(torch.as_tensor(xv_stat.stat1, dtype=torch.float32))       # F.normalize, on the device (stays there)
    scores = cosine_matrix(tsr, tsr)[ndx.trialmask]
    tar, non = key.tar[ndx.trialmask], key.non[ndx.trialmask]
    pmiss, pfa = rocch(scores[tar], scores[non])
    if not as_norm:
        return rocch2eer(pmiss, pfa)
    cohort = torch.as_tensor(model.state_dict()["after_speaker_embedding.weight"], dtype=torch.float32)
    s_scores = asnorm(tsr, normalize_rows_device(cohort), ndx)[ndx.trialmask]               # xvector.py:258; asnorm normalises once more, as in the reference (score_normalization.py:128)
    norm_pmiss, norm_pfa = rocch(s_scores[tar], s_scores[non])
    return rocch2eer(pmiss, pfa), rocch2eer(norm_pmiss, norm_pfa)


test_metrics.__test__ = False   # a library function with the reference's name, not a pytest case
