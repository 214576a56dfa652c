"""HDF5 helpers shared by the containers' ``read`` / ``write`` (layout of ``sidekit/bosaris/idmap.py:84-116,283-310``,
``ndx.py:92-112,184-204``, ``key.py:128-149,238-260``, ``scores.py:94-116,316-341``, ``statserver.py:392-489``): identifier
arrays as fixed-length byte strings, masks as int8, ``start`` / ``stop`` as int32 with -1 for "not set", every dataset gzip +
Fletcher-32 with unlimited maximum shape.  Files go through :mod:`sidekit_amd.hdf5_lite` (no h5py on the GPU box)."""
import numpy

from .. import hdf5_lite


def is_hdf5(file_name):
    try:
        with open(file_name, "rb") as f:
            return f.read(8) == hdf5_lite.SIGNATURE
    except OSError:
        return False


def ids_from_file(arr, width=None):
    """Byte-string dataset -> unicode array (the reference converts with ``astype('U')`` / ``'U100'`` / ``'U255'``)."""
    return numpy.asarray(arr).astype("U" if width is None else f"U{width}")


def bounds_from_file(arr):
    """int32 with -1 sentinels -> object array with ``None`` holes (``idmap.py:300-305``, ``statserver.py:412-417``)."""
    arr = numpy.asarray(arr)
    out = numpy.empty(arr.shape, "|O")
    out[arr != -1] = arr[arr != -1]
    return out


def bounds_to_file(arr):
    """Object array with ``None`` / NaN holes -> int32 with -1 (``idmap.py:101-107``)."""
    vals = numpy.array([numpy.nan if v is None else float(v) for v in numpy.asarray(arr, dtype="|O").ravel()], dtype=float)
    out = numpy.where(numpy.isnan(vals), -1, vals)
    return out.astype("int32").reshape(numpy.asarray(arr).shape)
