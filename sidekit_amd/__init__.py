"""sidekit_amd -- MI355X-native x-vector extraction and trial scoring behind SIDEKIT's Python surface.

Sub-modules mirror the reference package layout for the hot path only:
``sidekit_amd.nnet.xvector.Xtractor``, ``sidekit_amd.iv_scoring``, ``sidekit_amd.statserver``,
``sidekit_amd.bosaris``, ``sidekit_amd.score_normalization``, ``sidekit_amd.sidekit_io``.
``install_as_sidekit()`` registers them under the ``sidekit`` names so that reference-style drivers
(``extract_xvectors.py``, scoring scripts) import them unchanged.
"""
import importlib
import sys

import numpy

PARALLEL_MODULE = 'multiprocessing'  # sidekit/__init__.py:56
PARAM_TYPE = numpy.float32           # sidekit/__init__.py:57
STAT_TYPE = numpy.float64            # sidekit/__init__.py:58

__version__ = "0.2.0"

# the reference package re-exports these names at top level (sidekit/__init__.py:61-121); resolved on first use so that
# `import sidekit_amd` alone loads neither torch nor the HIP library
_LAZY = {
    "IdMap": "bosaris", "Ndx": "bosaris", "Key": "bosaris", "Scores": "bosaris", "effective_prior": "bosaris",
    "logit_effective_prior": "bosaris", "fast_minDCF": "bosaris",
    "StatServer": "statserver",
    "cosine_scoring": "iv_scoring", "PLDA_scoring": "iv_scoring", "fast_PLDA_scoring": "iv_scoring", "full_PLDA_scoring": "iv_scoring",
    "mahalanobis_scoring": "iv_scoring", "two_covariance_scoring": "iv_scoring",
    "asnorm": "score_normalization",
    "write_matrix_hdf5": "sidekit_io", "read_plda_hdf5": "sidekit_io", "write_plda_hdf5": "sidekit_io",
}

# every module of the mirror, by its reference name
SUBMODULES = ("bosaris", "bosaris.idmap", "bosaris.ndx", "bosaris.key", "bosaris.scores", "bosaris.detplot", "statserver", "iv_scoring",
              "score_normalization", "sidekit_io", "nnet", "nnet.xvector", "nnet.preprocessor")


def __getattr__(name):
    if name in _LAZY:
        value = getattr(importlib.import_module(f"{__name__}.{_LAZY[name]}"), name)
        globals()[name] = value
        return value
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def install_as_sidekit():
    """Alias this package as ``sidekit`` in ``sys.modules``: afterwards ``from sidekit.nnet.xvector import
    Xtractor``, ``from sidekit.iv_scoring import cosine_scoring``, ``sidekit.bosaris.detplot.rocch``,
    ``sidekit.score_normalization.asnorm`` ... resolve to the MI355X implementations.  A module that cannot be imported
    (e.g. the HIP library is not built) raises here, loudly, instead of leaving a partial alias behind."""
    pkg = sys.modules[__name__]
    if sys.modules.get("sidekit", pkg) is not pkg:
        raise ImportError("another package is already imported as `sidekit`; install_as_sidekit() must run before it")
    sys.modules["sidekit"] = pkg
    for sub in SUBMODULES:
        sys.modules["sidekit." + sub] = importlib.import_module(f"{__name__}.{sub}")
    return pkg
