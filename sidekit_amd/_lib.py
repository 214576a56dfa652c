"""ctypes binding of ``libsidekit_amd.so`` (the C ABI declared in ``include/sidekit_amd.h``).

There is no CPU fallback: if the HIP library is missing or cannot be loaded the import of any
compute entry point raises.  PyTorch is imported first so that the library binds to the same
``libamdhip64`` runtime instance torch uses (device pointers and streams are then interchangeable).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsidekit_amd.so")
if os.environ.get("SIDEKIT_AMD_LIB"):   # tuning aid: another build of the SAME library (A/B runs of kernel variants); never a different backend
    LIB_PATH = os.path.abspath(os.environ["SIDEKIT_AMD_LIB"])
    import warnings
    warnings.warn(f"SIDEKIT_AMD_LIB is set: loading {LIB_PATH} instead of the shipped csrc/libsidekit_amd.so (A/B tuning aid)", RuntimeWarning)
# The product library reads SIDEKIT_AMD_LANES and SIDEKIT_AMD_SMALL_GRID (csrc/common.h); every other SIDEKIT_AMD_* switch exists only in the A/B
# build (csrc/Makefile `make ab` -> libsidekit_amd_ab.so, loaded through SIDEKIT_AMD_LIB).  Say so instead of silently ignoring or obeying one.
_PRODUCT_VARS = {"SIDEKIT_AMD_LIB", "SIDEKIT_AMD_LANES", "SIDEKIT_AMD_SMALL_GRID", "SIDEKIT_AMD_PIPELINE_DEPTH"}
_IS_AB_BUILD = os.path.basename(LIB_PATH) != "libsidekit_amd.so"
for _tuning in sorted(k for k in os.environ if k.startswith("SIDEKIT_AMD_") and k not in _PRODUCT_VARS):
    import warnings
    if _IS_AB_BUILD:
        warnings.warn(f"{_tuning}={os.environ[_tuning]!r} is set: kernels may differ from the product configuration (A/B tuning aid)", RuntimeWarning)
    else:
        warnings.warn(f"{_tuning} is set but the product library ignores it: A/B switches exist only in libsidekit_amd_ab.so "
                      f"(make -C sidekit_amd/csrc ab; SIDEKIT_AMD_LIB=<that file>)", RuntimeWarning)

SK_OK, SK_EARG, SK_ESHAPE, SK_EHIP, SK_EWORKSPACE, SK_ESTATE = 0, -1, -2, -3, -4, -5
XT_ARCH_HALFRESNET34, XT_ARCH_TDNN = 0, 1
XT_F32, XT_BF16, XT_F64, XT_I64, XT_I16 = 0, 1, 2, 3, 4
XT_LOSS_AAM, XT_LOSS_CCE = 0, 1
XT_PROF_SLOTS = 17
PROF_NAMES = ("conv_L1", "conv_L1S", "conv_L2A", "conv_L2S", "conv_L2", "conv_L3A", "conv_L3S", "conv_L3", "conv_L4A", "conv_L4S",
              "conv_L4", "frontend", "stem", "se_residual", "pool_tail", "tdnn", "conv_pair_L1")


class XtConfig(ctypes.Structure):
    _fields_ = [("arch", ctypes.c_int32), ("dtype", ctypes.c_int32), ("loss", ctypes.c_int32), ("n_spk", ctypes.c_int32),
                ("emb_dim", ctypes.c_int32), ("aam_s", ctypes.c_float)]


_P = ctypes.c_void_p
_I32, _I64, _SZ, _F64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_double

# name -> (restype, argtypes); every symbol of include/sidekit_amd.h
SIGNATURES = {
    "xt_create": (ctypes.c_int, [ctypes.POINTER(XtConfig), ctypes.POINTER(_P)]),
    "xt_destroy": (ctypes.c_int, [_P]),
    "xt_set_tensor": (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.POINTER(_I64), _I32, _I32]),
    "xt_num_keys": (ctypes.c_int, [_P]),
    "xt_key_name": (ctypes.c_char_p, [_P, _I32]),
    "xt_finalize": (ctypes.c_int, [_P]),
    "xt_reserve": (ctypes.c_int, [_P, _I32, _I64]),
    "xt_forward": (ctypes.c_int, [_P, _P, _I64, _P, _I32, _I64, _P, _P, _P]),
    "xt_forward_pcm16": (ctypes.c_int, [_P, _P, _I64, _P, _I32, _I64, _P, _P, _P]),
    "xt_reserve_slots": (ctypes.c_int, [_P, _I32, _I32, _I64]),
    "xt_forward_begin": (ctypes.c_int, [_P, _I32, _P, _I32, _I64, _P, _I32, _I64, _P, _P, _P]),
    "xt_forward_end": (ctypes.c_int, [_P, _I32, _P]),
    "xt_forward_features": (ctypes.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _P]),
    "xt_features": (ctypes.c_int, [_P, _P, _I64, _P, _I32, _I64, _P, _P]),
    "xt_set_norm_embedding": (ctypes.c_int, [_P, _I32]),
    "xt_set_lanes": (ctypes.c_int, [_P, _I32]),
    "xt_get_lanes": (ctypes.c_int, [_P]),
    "xt_set_profile": (ctypes.c_int, [_P, _I32]),
    "xt_get_profile": (ctypes.c_int, [_P, ctypes.POINTER(_F64), ctypes.POINTER(_I64), _I32]),
    "xt_set_debug": (ctypes.c_int, [_P, _I32]),
    "xt_debug_tap": (ctypes.c_int, [_P, ctypes.c_char_p, _P, _SZ, ctypes.POINTER(_SZ)]),
    "xt_last_error": (ctypes.c_char_p, []),
    "sk_resample": (ctypes.c_int, [_P, _I32, _I64, _I32, _I32, _P, _I64, ctypes.POINTER(_I64), _P]),
    "sc_cosine": (ctypes.c_int, [_P, _I32, _P, _I32, _I32, _P, _P]),
    "sc_plda_fast": (ctypes.c_int, [_P, _I32, _P, _I32, _I32, _P, _P, _F64, _F64, _P, _P]),
    "sc_release_workspace": (ctypes.c_int, []),
    "sc_normalize_rows": (ctypes.c_int, [_P, _I32, _I32, _P, _P]),
    "sc_cosine_hist": (ctypes.c_int, [_P, _I32, _P, _I32, _I32, _P, _P, _I32, ctypes.c_float, ctypes.c_float, _I32, _P, _P, _P]),
    "sc_cosine_trials": (ctypes.c_int, [_P, _P, _I32, _P, _P, _I64, _P, _P]),
    "sc_topk_stats": (ctypes.c_int, [_P, _I32, _I32, _I32, _P, _P, _P]),
    "sc_snorm_apply": (ctypes.c_int, [_P, _I32, _I32, _P, _P, _P, _P, _P]),
    "sk_bench_conv": (ctypes.c_int, [_I32, _I32, _I32, _I32, _I32, _I32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_F64)]),
    "sk_pavx": (ctypes.c_int, [_P, _I64, _P, _P, _P, ctypes.POINTER(_I64)]),
    "sk_rocch_vertices": (ctypes.c_int, [_P, _I64, _I64, _I64, _P, _I64, _P, _P]),
    "sk_wav_probe": (ctypes.c_int, [_P, _I32, _I32, _P, _P, _P, _P]),
    "sk_wav_read_pcm16": (ctypes.c_int, [_P, _P, _P, _P, _I32, _I32, _P, _I64, _I32, _P]),
}

_lib = None


def lib():
    """Load (once) and return the HIP library.  Raises ImportError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(or `make -C sidekit_amd/csrc`). sidekit_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7; ours resolves to the same runtime instance)
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
    cdll = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = cdll
    import atexit
    atexit.register(_release_at_exit)
    return _lib


def _release_at_exit():
    """Free the scoring workspace cache (scoring.hip); only if this process ever used the GPU."""
    try:
        import torch
        if _lib is not None and torch.cuda.is_initialized():
            _lib.sc_release_workspace()
    except Exception:   # interpreter teardown: nothing to report to
        pass


def last_error():
    msg = lib().xt_last_error()
    return msg.decode() if msg else ""


def check(rc, arg_exc=ValueError):
    """Turn an ABI error class into the Python exception the reference would raise."""
    if rc == SK_OK:
        return
    msg = last_error()
    if rc == SK_EARG:
        raise arg_exc(msg)
    raise RuntimeError(msg)
