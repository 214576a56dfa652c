"""Sample-rate conversion on the GPU: what ``sidekit/bin/extract_xvectors.py:141-143`` does with
``torchaudio.transforms.Resample(orig_freq=sr, new_freq=sample_rate)`` when a file's rate differs from the model's.

``sk_resample`` (csrc/resample.hip) restates torchaudio 0.8.2's windowed-sinc interpolation; torchaudio is not vendored, so
parity is UNPINNED at this boundary (as for the mel front-end).  No CPU fallback.
"""
import ctypes

import numpy
import torch

from . import _lib


def resample(signal, orig_freq, new_freq, device=None):
    """1-D float32 / int16 tensor or array -> float32 tensor of ``ceil(new * n / orig)`` samples on the GPU."""
    if int(orig_freq) == int(new_freq):
        x = torch.as_tensor(signal)
        return (x.float() / 32768.0 if x.dtype == torch.int16 else x.float()).to(device or "cuda")
    if not torch.cuda.is_available():
        raise RuntimeError("sidekit_amd.resample computes on the GPU only (no CPU fallback) and no GPU is visible")
    x = torch.as_tensor(numpy.ascontiguousarray(signal) if isinstance(signal, numpy.ndarray) else signal)
    if x.dim() != 1:
        raise RuntimeError(f"expected a 1-D signal, got shape {tuple(x.shape)}")
    dev = torch.device(device) if device is not None else (x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    if x.dtype != torch.int16:
        x = x.float()
    x = x.to(dev).contiguous()
    lib = _lib.lib()
    n_out = ctypes.c_int64(0)
    dt = _lib.XT_I16 if x.dtype == torch.int16 else _lib.XT_F32
    _lib.check(lib.sk_resample(None, dt, x.shape[0], int(orig_freq), int(new_freq), None, 0, ctypes.byref(n_out), None))
    out = torch.empty(n_out.value, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.sk_resample(x.data_ptr(), dt, x.shape[0], int(orig_freq), int(new_freq), out.data_ptr(), out.shape[0], ctypes.byref(n_out),
                                   ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out
