"""HDF5 parameter files -- mirror of the part of ``sidekit/sidekit_io.py`` the scoring path exchanges: PLDA models
(``write_plda_hdf5`` / ``read_plda_hdf5``, :282-324), normalisation parameters (``write_norm_hdf5`` / ``read_norm_hdf5``,
:246-280), single matrices and dictionaries (``write_matrix_hdf5`` / ``read_matrix_hdf5`` / ``write_dict_hdf5`` /
``read_dict_hdf5``).  Same dataset names and types as the reference, written and read with
:mod:`sidekit_amd.hdf5_lite` (gzip + Fletcher-32 chunks), so the files open in h5py and vice versa."""
import numpy

from . import hdf5_lite


def write_matrix_hdf5(M, filename):
    """One array under the dataset name ``matrix``."""
    w = hdf5_lite.Writer()
    w["matrix"] = M
    w.save(filename)


def read_matrix_hdf5(filename):
    with hdf5_lite.File(filename) as f:
        return f["matrix"][()]


def write_dict_hdf5(data, output_filename):
    """``{"group/name": array}`` -> one dataset per key."""
    w = hdf5_lite.Writer()
    for key, value in data.items():
        w[key] = value
    w.save(output_filename)


def read_dict_hdf5(input_filename):
    """Every ``group/name`` dataset of a two-level file, as the reference returns it."""
    data = {}
    with hdf5_lite.File(input_filename) as f:
        for key in f.keys():
            for key2 in f[key].keys():
                data[key + '/' + key2] = f[key][key2][()]
    return data


def write_norm_hdf5(data, output_filename):
    """``(means, covs)`` lists (one entry per normalisation iteration) -> ``norm/means``, ``norm/covs``."""
    w = hdf5_lite.Writer()
    w["norm/means"] = numpy.asarray(data[0])
    w["norm/covs"] = numpy.asarray(data[1])
    w.save(output_filename)


def read_norm_hdf5(input_filename):
    with hdf5_lite.File(input_filename) as f:
        return f["norm/means"][()], f["norm/covs"][()]


def write_plda_hdf5(data, output_filename):
    """``(mean, F, G, Sigma)`` -> ``plda/mean``, ``plda/f``, ``plda/g``, ``plda/sigma``."""
    w = hdf5_lite.Writer()
    for name, value in zip(("mean", "f", "g", "sigma"), data):
        w["plda/" + name] = numpy.asarray(value)
    w.save(output_filename)


def read_plda_hdf5(input_filename):
    """-> ``(mean, F, G, Sigma)``, the arguments of ``PLDA_scoring`` (``sidekit/iv_scoring.py:215``)."""
    with hdf5_lite.File(input_filename) as f:
        return f["plda/mean"][()], f["plda/f"][()], f["plda/g"][()], f["plda/sigma"][()]
