"""The C-ABI library loads on a CPU-only host and exports every symbol include/*.h declares."""
import ctypes
import os
import re

from sidekit_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names += re.findall(r"^\s*(?:const\s+)?(?:int|char\s*\*|void)\s*\**\s*((?:xt|sc|sk)_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    declared = _declared()
    assert len(declared) >= 20
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(cdll, name), f"{name} declared in include/ but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes binding table and header disagree"
    _lib.lib()   # binds argtypes for all of them


def test_error_plumbing_without_gpu():
    lib = _lib.lib()
    assert lib.xt_create(None, None) == _lib.SK_EARG
    assert "null" in _lib.last_error()
    assert lib.sc_cosine(None, 0, None, 0, 0, None, None) == _lib.SK_EARG
    assert lib.xt_destroy(None) == _lib.SK_OK
