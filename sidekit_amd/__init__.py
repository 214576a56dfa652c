"""sidekit_amd -- MI355X-native x-vector extraction and trial scoring behind SIDEKIT's Python surface.

Sub-modules mirror the reference package layout for the hot path only:
``sidekit_amd.nnet.xvector.Xtractor``, ``sidekit_amd.iv_scoring``, ``sidekit_amd.statserver``,
``sidekit_amd.bosaris``.  ``install_as_sidekit()`` registers them under the ``sidekit`` names so
that reference-style drivers (``extract_xvectors.py``, scoring scripts) import them unchanged.
"""
import sys

STAT_TYPE = "float64"   # sidekit/__init__.py:59
PARAM_TYPE = "float32"  # sidekit/__init__.py:58

__version__ = "0.1.0"


def install_as_sidekit():
    """Alias this package as ``sidekit`` in ``sys.modules``."""
    import importlib
    pkg = sys.modules[__name__]
    sys.modules.setdefault("sidekit", pkg)
    for sub in ("nnet", "nnet.xvector", "nnet.preprocessor", "bosaris", "statserver", "iv_scoring"):
        try:
            sys.modules.setdefault("sidekit." + sub, importlib.import_module(__name__ + "." + sub))
        except ImportError:
            pass
    return pkg
