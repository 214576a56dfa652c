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


def test_ab_build_exports_the_same_symbols():
    """`make ab` (-DSK_AB: every A/B switch and alternative kernel) is the same C ABI: the tests that load it through SIDEKIT_AMD_LIB bind the same table."""
    ab = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsidekit_amd_ab.so")
    if not os.path.exists(ab):
        import pytest
        pytest.skip("the A/B build has not been made here (__graft_entry__.build() makes it)")
    cdll = ctypes.CDLL(ab)
    for name in _declared():
        assert hasattr(cdll, name), f"{name} missing from the A/B build"


def test_product_library_reads_two_environment_variables():
    """No tuning switch in what users load: the product library's strings name SIDEKIT_AMD_LANES and SIDEKIT_AMD_SMALL_GRID only."""
    blob = open(_lib.LIB_PATH if os.path.basename(_lib.LIB_PATH) == "libsidekit_amd.so" else os.path.join(os.path.dirname(_lib.LIB_PATH), "libsidekit_amd.so"), "rb").read()
    names = set(m.decode() for m in re.findall(rb"SIDEKIT_AMD_[A-Z0-9_]+", blob))
    assert names == {"SIDEKIT_AMD_LANES", "SIDEKIT_AMD_SMALL_GRID"}, names


def test_error_plumbing_without_gpu():
    lib = _lib.lib()
    assert lib.xt_create(None, None) == _lib.SK_EARG
    assert "null" in _lib.last_error()
    assert lib.sc_cosine(None, 0, None, 0, 0, None, None) == _lib.SK_EARG
    assert lib.xt_destroy(None) == _lib.SK_OK
