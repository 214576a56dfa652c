"""``Scores`` -- in-memory mirror of ``sidekit/bosaris/scores.py:52-81,118-133,157-313,371-412``."""
import logging
import os

import numpy

from . import _h5

from ._sets import as_ids, first_index, member_mask, read_columns, sorted_difference
from .key import Key
from .ndx import Ndx


class Scores:
    """``scoremat[model, segment]`` with ``scoremask`` flagging the trials that were scored."""

    def __init__(self, scores_file_name=''):
        self.modelset = numpy.empty(0, dtype="|O")
        self.segset = numpy.empty(0, dtype="|O")
        self.scoremask = numpy.array([], dtype="bool")
        self.scoremat = numpy.array([])
        if scores_file_name != '':   # the reference reads HDF5 here (scores.py:75-81); a text score file is recognised by its first bytes
            tmp = Scores.read(scores_file_name) if _h5.is_hdf5(scores_file_name) else Scores.read_txt(scores_file_name)
            self.modelset, self.segset, self.scoremask, self.scoremat = tmp.modelset, tmp.segset, tmp.scoremask, tmp.scoremat

    def write(self, output_file_name):
        """HDF5 form of ``scores.py:94-116``: ``score_mask`` int8, ``scores`` in the matrix's own float type."""
        w = _h5.hdf5_lite.Writer()
        w["modelset"] = self.modelset.astype('S')
        w["segset"] = self.segset.astype('S')
        w["score_mask"] = self.scoremask.astype('int8')
        w["scores"] = self.scoremat
        w.save(output_file_name)

    @staticmethod
    def read(input_file_name):
        """``scores.py:316-341``."""
        with _h5.hdf5_lite.File(input_file_name) as f:
            scores = Scores()
            scores.modelset = _h5.ids_from_file(f["modelset"][()], 100)
            scores.segset = _h5.ids_from_file(f["segset"][()], 100)
            scores.scoremask = f["score_mask"][()].astype('bool')
            scores.scoremat = f["scores"][()]
        assert scores.validate(), "Error: wrong Scores format"
        return scores

    def validate(self):
        ok = self.scoremat.shape == self.scoremask.shape
        ok &= self.scoremat.shape[0] == self.modelset.shape[0]
        ok &= self.scoremat.shape[1] == self.segset.shape[0]
        return bool(ok)

    def get_tar_non(self, key):
        """Target and non-target score vectors according to ``key`` (scores.py:157-177)."""
        same = (key.modelset.shape == self.modelset.shape and key.segset.shape == self.segset.shape
                and bool((key.modelset == self.modelset).all()) and bool((key.segset == self.segset).all())
                and self.scoremask.shape == key.tar.shape)
        if same:
            return self.scoremat[key.tar & self.scoremask], self.scoremat[key.non & self.scoremask]
        new_score = self.align_with_ndx(key)
        return new_score.scoremat[key.tar & new_score.scoremask], new_score.scoremat[key.non & new_score.scoremask]

    def align_with_ndx(self, ndx):
        """Resize / reorder to the model and segment order of ``ndx`` (a Key or an Ndx), scores.py:179-240."""
        out = Scores()
        out.modelset, out.segset = ndx.modelset, ndx.segset
        hasmodel = member_mask(ndx.modelset, self.modelset)
        hasseg = member_mask(ndx.segset, self.segset)
        rindx = first_index(self.modelset, ndx.modelset[hasmodel])
        cindx = first_index(self.segset, ndx.segset[hasseg])
        rows, cols = numpy.where(hasmodel)[0][:, None], numpy.where(hasseg)[0]
        out.scoremat = numpy.zeros((ndx.modelset.shape[0], ndx.segset.shape[0]))
        out.scoremask = numpy.zeros((ndx.modelset.shape[0], ndx.segset.shape[0]), dtype='bool')
        if rindx.size and cindx.size:
            out.scoremat[rows, cols] = self.scoremat[rindx[:, None], cindx]
            out.scoremask[rows, cols] = self.scoremask[rindx[:, None], cindx]
        wanted = ndx.trialmask if isinstance(ndx, Ndx) else (ndx.tar | ndx.non)
        out.scoremask = out.scoremask & wanted
        if hasmodel.sum() < ndx.modelset.shape[0]:
            logging.info('models reduced from %d to %d', ndx.modelset.shape[0], hasmodel.sum())
        if hasseg.sum() < ndx.segset.shape[0]:
            logging.info('testsegs reduced from %d to %d', ndx.segset.shape[0], hasseg.sum())
        missing = int(wanted.sum() - (wanted & out.scoremask).sum())
        if missing > 0:
            logging.info('%d of %d trials missing', missing, int(wanted.sum()))
        assert numpy.isfinite(out.scoremat[out.scoremask]).all(), 'Inifinite or Nan value in the scoremat'
        assert out.validate(), 'Wrong Score format'
        return out

    def set_missing_to_value(self, ndx, value):
        if isinstance(ndx, Key):
            ndx = ndx.to_ndx()
        new_scr = self.align_with_ndx(ndx)
        missing = ndx.trialmask & ~new_scr.scoremask
        new_scr.scoremat[missing] = value
        new_scr.scoremask[missing] = True
        assert new_scr.validate(), "Wrong format of Scores"
        return new_scr

    def filter(self, modlist, seglist, keep):
        if keep:
            keepmods, keepsegs = modlist, seglist
        else:
            keepmods = sorted_difference(self.modelset, modlist)
            keepsegs = sorted_difference(self.segset, seglist)
        km = member_mask(self.modelset, keepmods)
        ks = member_mask(self.segset, keepsegs)
        out = Scores()
        out.modelset, out.segset = self.modelset[km], self.segset[ks]
        out.scoremat = self.scoremat[km, :][:, ks]
        out.scoremask = self.scoremask[km, :][:, ks]
        return out

    def sort(self):
        """Sort models and segments alphabetically (scores.py:469-478)."""
        im, isg = numpy.argsort(self.modelset), numpy.argsort(self.segset)
        self.modelset, self.segset = self.modelset[im], self.segset[isg]
        self.scoremat = self.scoremat[im, :][:, isg]
        self.scoremask = self.scoremask[im, :][:, isg]

    def get_score(self, modelID, segID):
        m = numpy.argwhere(self.modelset == modelID)
        s = numpy.argwhere(self.segset == segID)
        if m.shape[0] == 0:
            raise Exception('No such model as: %s' % modelID)
        if s.shape[0] == 0:
            raise Exception('No such segment as: %s' % segID)
        return self.scoremat[m, s]

    def write_txt(self, output_file_name):
        """``model segment score`` per scored trial (the format tools/compute_metrics.py reads)."""
        d = os.path.dirname(output_file_name)
        if d and not os.path.exists(d):
            os.makedirs(d)
        with open(output_file_name, 'w') as f:
            for m in range(self.modelset.shape[0]):
                segs = self.segset[self.scoremask[m, ]]
                scores = self.scoremat[m, self.scoremask[m, ]]
                for s in range(segs.shape[0]):
                    f.write('{} {} {}\n'.format(self.modelset[m], segs[s], scores[s]))

    @classmethod
    def read_txt(cls, input_file_name):
        models, testsegs, scores = read_columns(input_file_name, 3)
        models, testsegs = as_ids(models), as_ids(testsegs)
        s = cls()
        s.modelset, s.segset = numpy.unique(models), numpy.unique(testsegs)
        s.scoremask = numpy.zeros((s.modelset.shape[0], s.segset.shape[0]), dtype="bool")
        s.scoremat = numpy.zeros((s.modelset.shape[0], s.segset.shape[0]))
        mi, si = numpy.searchsorted(s.modelset, models), numpy.searchsorted(s.segset, testsegs)
        s.scoremask[mi, si] = True
        s.scoremat[mi, si] = scores.astype(float)
        assert s.validate(), "Wrong Scores format"
        return s
