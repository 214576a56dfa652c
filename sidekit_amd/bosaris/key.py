"""``Key`` -- in-memory mirror of ``sidekit/bosaris/key.py:49-126,151-236,262-303``: which trials are
target and which are non-target."""
import logging

import numpy

from . import _h5

from ._sets import as_ids, member_mask, read_columns, sorted_difference
from .ndx import Ndx


class Key:
    """``tar`` / ``non`` boolean matrices over sorted unique ``modelset`` x ``segset``."""

    def __init__(self, key_file_name=None, models=numpy.array([]), testsegs=numpy.array([]), trials=numpy.array([])):
        self.modelset = numpy.empty(0, dtype="|O")
        self.segset = numpy.empty(0, dtype="|O")
        self.tar = numpy.array([], dtype="bool")
        self.non = numpy.array([], dtype="bool")
        if key_file_name is None and models is None and testsegs is None and trials is None:
            return
        if key_file_name is None:
            self._fill(numpy.asarray(models), numpy.asarray(testsegs), numpy.asarray(trials))
        else:   # the reference reads HDF5 here (key.py:84-90); a text key is recognised by its first bytes
            tmp = Key.read(key_file_name) if _h5.is_hdf5(key_file_name) else Key.read_txt(key_file_name)
            self.modelset, self.segset, self.tar, self.non = tmp.modelset, tmp.segset, tmp.tar, tmp.non

    def write(self, output_file_name):
        """HDF5 form of ``key.py:128-149``: ``trial_mask`` int8 = +1 target, -1 non-target, 0 no trial."""
        assert self.validate(), "Error: wrong Key format"
        w = _h5.hdf5_lite.Writer()
        w["modelset"] = self.modelset.astype('S')
        w["segset"] = self.segset.astype('S')
        w["trial_mask"] = numpy.array(self.tar, dtype='int8') - numpy.array(self.non, dtype='int8')
        w.save(output_file_name)

    @staticmethod
    def read(input_file_fame):
        """``key.py:238-260``."""
        with _h5.hdf5_lite.File(input_file_fame) as f:
            key = Key(None, None, None, None)
            key.modelset = _h5.ids_from_file(f["modelset"][()], 100)
            key.segset = _h5.ids_from_file(f["segset"][()], 100)
            trialmask = f["trial_mask"][()]
            key.tar = (trialmask == 1)
            key.non = (trialmask == -1)
        assert key.validate(), "Error: wrong Key format"
        return key

    def _fill(self, models, testsegs, trials):
        self.modelset = numpy.unique(models)
        self.segset = numpy.unique(testsegs)
        self.tar = numpy.zeros((self.modelset.shape[0], self.segset.shape[0]), dtype="bool")
        self.non = numpy.zeros_like(self.tar)
        if len(models):
            mi = numpy.searchsorted(self.modelset, models)
            si = numpy.searchsorted(self.segset, testsegs)
            # a (model, segment) pair listed twice keeps its LAST label, as the reference's dict(zip(...)) does
            last = {}
            for i, pair in enumerate(zip(mi.tolist(), si.tolist())):
                last[pair] = i
            idx = numpy.fromiter(last.values(), dtype=numpy.int64, count=len(last))
            self.tar[mi[idx], si[idx]] = trials[idx] == 'target'
            self.non[mi[idx], si[idx]] = trials[idx] == 'nontarget'
        assert self.validate(), "Wrong Key format"

    @classmethod
    def create(cls, modelset, segset, tar, non):
        key = cls()
        key.modelset, key.segset, key.tar, key.non = modelset, segset, tar, non
        assert key.validate(), "Wrong Key format"
        return key

    def validate(self):
        ok = all(isinstance(a, numpy.ndarray) for a in (self.modelset, self.segset, self.tar, self.non))
        ok = ok and self.modelset.ndim == 1 and self.segset.ndim == 1 and self.tar.ndim == 2 and self.non.ndim == 2
        ok = ok and self.tar.shape == self.non.shape
        ok = ok and self.tar.shape == (self.modelset.shape[0], self.segset.shape[0])
        return bool(ok)

    def to_ndx(self):
        ndx = Ndx()
        ndx.modelset, ndx.segset, ndx.trialmask = self.modelset, self.segset, self.tar | self.non
        return ndx

    def filter(self, modlist, seglist, keep):
        if keep:
            keepmods, keepsegs = modlist, seglist
        else:
            keepmods = sorted_difference(self.modelset, modlist)
            keepsegs = sorted_difference(self.segset, seglist)
        km = member_mask(self.modelset, keepmods)
        ks = member_mask(self.segset, keepsegs)
        out = Key()
        out.modelset, out.segset = self.modelset[km], self.segset[ks]
        out.tar = self.tar[km, :][:, ks]
        out.non = self.non[km, :][:, ks]
        assert out.validate(), "Wrong Key format"
        if self.modelset.shape[0] > out.modelset.shape[0]:
            logging.info('Number of models reduced from %d to %d', self.modelset.shape[0], out.modelset.shape[0])
        if self.segset.shape[0] > out.segset.shape[0]:
            logging.info('Number of test segments reduced from %d to %d', self.segset.shape[0], out.segset.shape[0])
        return out

    def write_txt(self, output_file_name):
        with open(output_file_name, 'w') as f:
            for m in range(self.modelset.shape[0]):
                for s in self.segset[self.tar[m, ]]:
                    f.write('{} {} {}\n'.format(self.modelset[m], s, 'target'))
                for s in self.segset[self.non[m, ]]:
                    f.write('{} {} {}\n'.format(self.modelset[m], s, 'nontarget'))

    @staticmethod
    def read_txt(input_file_name):
        """``model segment target|nontarget`` per line (tools/compute_metrics.py key format)."""
        models, testsegs, trials = read_columns(input_file_name, 3)
        key = Key()
        key._fill(as_ids(models).astype('U'), as_ids(testsegs).astype('U'), as_ids(trials).astype('U'))
        return key
