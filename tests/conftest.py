import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


AB_LIB = os.path.join(ROOT, "sidekit_amd", "csrc", "libsidekit_amd_ab.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ab_variant: compares a product path with an A/B partner that only the -DSK_AB build of the library holds "
                                       "(csrc/Makefile `make ab`); run by tests/test_gpu_01_ab_variant.py in a child process that loads that build")


def pytest_collection_modifyitems(config, items):
    """The product library carries no A/B switch (csrc/common.h): tests marked ``ab_variant`` run only in the child process that
    tests/test_gpu_01_ab_variant.py starts with SIDEKIT_AMD_LIB = the A/B build; everywhere else they are deselected."""
    if os.environ.get("SK_AB_CHILD") == "1":
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("ab_variant") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback)")
    return torch.device("cuda", 0)
