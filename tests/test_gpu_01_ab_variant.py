"""The A/B partners of the product paths, in the one process that loads the A/B build of the library.

The product `libsidekit_amd.so` reads two environment variables and holds thirteen convolution shapes; every tuning switch
(`SIDEKIT_AMD_SHORTCUT_TENSOR`, `_MEL_GEMM`, `_ATT_SEPARATE`, `_MFCC_DFT_GEMM`, `_GEMM64`, `_GATE_PROLOGUE`, `_SHAPE_MAP`, ...), the
alternative shapes of `sk_bench_conv` and the in-convolution SE-gate forms (`csrc/se_gate_inl.h`) exist only in
`libsidekit_amd_ab.so` = the same sources built with `-DSK_AB` (`make -C sidekit_amd/csrc ab`, `__graft_entry__.build()`).  The tests
that compare a product path with such a partner are marked `ab_variant`; they are deselected everywhere (tests/conftest.py) except in
the child started here with `SIDEKIT_AMD_LIB` pointing at that build.  Without a switch set the A/B build runs the product's code path,
so "default vs partner" inside that child is the comparison the tests always made.

Sorts right behind test_gpu_00_multirank.py: the child is created by a pytest process that has not touched the GPU yet.
"""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB_LIB = os.path.join(ROOT, "sidekit_amd", "csrc", "libsidekit_amd_ab.so")
pytestmark = pytest.mark.gpu


def test_ab_partners_agree_with_the_product_paths():
    if torch.cuda.device_count() < 1:      # counting devices does not initialise the GPU in this process
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback)")
    assert os.path.exists(AB_LIB), f"{AB_LIB} is missing: __graft_entry__.build() / `make -C sidekit_amd/csrc ab` builds it"
    env = dict(os.environ, SIDEKIT_AMD_LIB=AB_LIB, SK_AB_CHILD="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    proc = subprocess.run([sys.executable, "-W", "ignore::RuntimeWarning", "-m", "pytest", "tests", "-x", "-q", "-m", "gpu and ab_variant", "-p", "no:cacheprovider"],
                          env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = proc.stdout[-3000:]
    print(tail)
    assert proc.returncode == 0, f"A/B partner tests failed in the child ({proc.returncode}):\n{tail}\n{proc.stderr[-2000:]}"
    m = re.search(r"(\d+) passed", proc.stdout)
    assert m and int(m.group(1)) >= 5, f"expected the five ab_variant tests to run:\n{tail}"
