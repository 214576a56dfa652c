"""``Ndx`` -- in-memory mirror of ``sidekit/bosaris/ndx.py:48-91,115-182,208-237``: the list of
(model, test segment) trials as a boolean mask over sorted unique identifier sets."""
import logging

import numpy

from . import _h5

from ._sets import as_ids, member_mask, read_columns, sorted_difference


def _trial_matrix(models, testsegs, modelset, segset):
    mask = numpy.zeros((modelset.shape[0], segset.shape[0]), dtype="bool")
    if len(models):
        mi = numpy.searchsorted(modelset, models)
        si = numpy.searchsorted(segset, testsegs)
        mask[mi, si] = True
    return mask


class Ndx:
    """Trial index: ``modelset`` / ``segset`` (sorted, unique) and ``trialmask[model, segment]``."""

    def __init__(self, ndx_file_name='', models=numpy.array([]), testsegs=numpy.array([])):
        self.modelset = numpy.empty(0, dtype="|O")
        self.segset = numpy.empty(0, dtype="|O")
        self.trialmask = numpy.array([], dtype="bool")
        if ndx_file_name == '':
            models, testsegs = numpy.asarray(models), numpy.asarray(testsegs)
            self.modelset = numpy.unique(models)
            self.segset = numpy.unique(testsegs)
            self.trialmask = _trial_matrix(models, testsegs, self.modelset, self.segset)
            assert self.validate(), "Wrong Ndx format"
        else:   # the reference reads HDF5 here (ndx.py:85-90); a text trial list is recognised by its first bytes
            tmp = Ndx.read(ndx_file_name) if _h5.is_hdf5(ndx_file_name) else Ndx.read_txt(ndx_file_name)
            self.modelset, self.segset, self.trialmask = tmp.modelset, tmp.segset, tmp.trialmask

    def write(self, output_file_name):
        """HDF5 form of ``ndx.py:92-112``: ``modelset`` / ``segset`` byte strings, ``trial_mask`` int8."""
        assert self.validate(), "Error: wrong Ndx format"
        w = _h5.hdf5_lite.Writer()
        w["modelset"] = self.modelset.astype('S')
        w["segset"] = self.segset.astype('S')
        w["trial_mask"] = self.trialmask.astype('int8')
        w.save(output_file_name)

    @staticmethod
    def read(input_file_name):
        """``ndx.py:184-204``."""
        with _h5.hdf5_lite.File(input_file_name) as f:
            ndx = Ndx()
            ndx.modelset = _h5.ids_from_file(f["modelset"][()], 100)
            ndx.segset = _h5.ids_from_file(f["segset"][()], 100)
            ndx.trialmask = f["trial_mask"][()].astype("bool")
        assert ndx.validate(), "Error: wrong Ndx format"
        return ndx

    def validate(self):
        ok = isinstance(self.modelset, numpy.ndarray) and isinstance(self.segset, numpy.ndarray)
        ok &= isinstance(self.trialmask, numpy.ndarray)
        ok &= self.modelset.ndim == 1 and self.segset.ndim == 1 and self.trialmask.ndim == 2
        ok &= self.trialmask.shape == (self.modelset.shape[0], self.segset.shape[0])
        return bool(ok)

    def filter(self, modlist, seglist, keep):
        """Keep (or, with ``keep=False``, discard) the listed models and segments."""
        if keep:
            keepmods, keepsegs = modlist, seglist
        else:
            keepmods = sorted_difference(self.modelset, modlist)
            keepsegs = sorted_difference(self.segset, seglist)
        km = member_mask(self.modelset, keepmods)
        ks = member_mask(self.segset, keepsegs)
        out = Ndx()
        out.modelset = self.modelset[km]
        out.segset = self.segset[ks]
        out.trialmask = self.trialmask[km, :][:, ks]
        assert out.validate(), "Wrong Ndx format"
        if self.modelset.shape[0] > out.modelset.shape[0]:
            logging.info('Number of models reduced from %d to %d', self.modelset.shape[0], out.modelset.shape[0])
        if self.segset.shape[0] > out.segset.shape[0]:
            logging.info('Number of test segments reduced from %d to %d', self.segset.shape[0], out.segset.shape[0])
        return out

    def save_txt(self, output_file_name):
        with open(output_file_name, 'w') as f:
            for m in range(self.modelset.shape[0]):
                for s in self.segset[self.trialmask[m, ]]:
                    f.write('{} {}\n'.format(self.modelset[m], s))

    @classmethod
    def read_txt(cls, input_filename):
        models, testsegs = read_columns(input_filename, 2)
        return cls(models=as_ids(models), testsegs=as_ids(testsegs))

    def merge(self, ndx_list):
        """Union of this Ndx with a list of others."""
        models = [numpy.repeat(n.modelset, n.trialmask.sum(1)) for n in [self] + list(ndx_list)]
        segs = [n.segset[numpy.nonzero(n.trialmask)[1]] for n in [self] + list(ndx_list)]
        merged = Ndx(models=numpy.concatenate(models), testsegs=numpy.concatenate(segs))
        self.modelset, self.segset, self.trialmask = merged.modelset, merged.segset, merged.trialmask
