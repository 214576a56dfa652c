"""``IdMap`` -- in-memory mirror of ``sidekit/bosaris/idmap.py:40-72,243-282`` (+ text I/O :118-126,312-339).
HDF5 persistence is out of scope."""
import copy
import logging

import numpy

from . import _h5

from ._sets import as_ids, member_mask, read_columns


class IdMap:
    """Map between two lists of identifiers: ``leftids`` (classes / models) and ``rightids`` (segments),
    with optional ``start`` / ``stop`` frame indices."""

    def __init__(self, idmap_filename=''):
        self.leftids = numpy.empty(0, dtype="|O")
        self.rightids = numpy.empty(0, dtype="|O")
        self.start = numpy.empty(0, dtype="|O")
        self.stop = numpy.empty(0, dtype="|O")
        if idmap_filename != '':   # the reference reads HDF5 here (idmap.py:62-67); a text file is recognised by its first bytes
            tmp = IdMap.read(idmap_filename) if _h5.is_hdf5(idmap_filename) else IdMap.read_txt(idmap_filename)
            self.leftids, self.rightids, self.start, self.stop = tmp.leftids, tmp.rightids, tmp.start, tmp.stop

    def write(self, output_file_name):
        """HDF5 form of ``idmap.py:84-116``: ``leftids`` / ``rightids`` byte strings, ``start`` / ``stop`` int32 (-1 = unset)."""
        assert self.validate(), "Error: wrong IdMap format"
        w = _h5.hdf5_lite.Writer()
        w["leftids"] = self.leftids.astype('S')
        w["rightids"] = self.rightids.astype('S')
        w["start"] = _h5.bounds_to_file(self.start)
        w["stop"] = _h5.bounds_to_file(self.stop)
        w.save(output_file_name)

    @staticmethod
    def read(input_file_name):
        """``idmap.py:283-310``."""
        with _h5.hdf5_lite.File(input_file_name) as f:
            idmap = IdMap()
            idmap.leftids = _h5.ids_from_file(f["leftids"][()], 255)
            idmap.rightids = _h5.ids_from_file(f["rightids"][()], 255)
            idmap.start = _h5.bounds_from_file(f["start"][()])
            idmap.stop = _h5.bounds_from_file(f["stop"][()])
        assert idmap.validate(), "Error: wrong IdMap format"
        return idmap

    def __repr__(self):
        return ('-' * 30 + '\nleft ids:' + repr(self.leftids) + '\nright ids:' + repr(self.rightids) + '\nseg start:'
                + repr(self.start) + '\nseg stop:' + repr(self.stop) + '\n' + '-' * 30 + '\n')

    def set(self, left, right, start=None, stop=None):
        self.leftids = copy.deepcopy(left)
        self.rightids = copy.deepcopy(right)
        self.start = copy.deepcopy(start) if start is not None else numpy.empty(self.rightids.shape, '|O')
        self.stop = copy.deepcopy(stop) if stop is not None else numpy.empty(self.rightids.shape, '|O')

    def validate(self, warn=False):
        ok = (self.leftids.shape == self.rightids.shape == self.start.shape == self.stop.shape) & self.leftids.ndim == 1
        if warn and self.leftids.shape != numpy.unique(self.leftids).shape:
            logging.warning('The left id list contains duplicate identifiers')
        if warn and self.rightids.shape != numpy.unique(self.rightids).shape:
            logging.warning('The right id list contains duplicate identifiers')
        return ok

    def map_left_to_right(self, leftidlist):
        table = dict(zip(self.leftids.tolist(), self.rightids.tolist()))
        return numpy.array([table[i] for i in leftidlist if i in table], dtype=object)

    def map_right_to_left(self, rightidlist):
        table = dict(zip(self.rightids.tolist(), self.leftids.tolist()))
        return numpy.array([table[i] for i in rightidlist if i in table], dtype=object)

    def _filter(self, ids, idlist, keep):
        mask = member_mask(ids, idlist)
        if not keep:
            mask = ~mask
        out = IdMap()
        out.leftids, out.rightids = self.leftids[mask], self.rightids[mask]
        out.start, out.stop = self.start[mask], self.stop[mask]
        return out

    def filter_on_left(self, idlist, keep):
        return self._filter(self.leftids, idlist, keep)

    def filter_on_right(self, idlist, keep):
        return self._filter(self.rightids, idlist, keep)

    def merge(self, idmap2):
        out = IdMap()
        out.leftids = numpy.concatenate((self.leftids, idmap2.leftids))
        out.rightids = numpy.concatenate((self.rightids, idmap2.rightids))
        out.start = numpy.concatenate((self.start, idmap2.start))
        out.stop = numpy.concatenate((self.stop, idmap2.stop))
        return out

    def write_txt(self, output_file_name):
        with open(output_file_name, 'w') as f:
            for left, right, start, stop in zip(self.leftids, self.rightids, self.start, self.stop):
                f.write(' '.join(filter(None, (left, right, str(start), str(stop)))) + '\n')

    @classmethod
    def read_txt(cls, input_file_name):
        with open(input_file_name, "r") as f:
            columns = len(f.readline().split())
        idmap = cls()
        if columns >= 4:
            left, right, start, stop = read_columns(input_file_name, 4)
            idmap.leftids, idmap.rightids = as_ids(left), as_ids(right)
            idmap.start, idmap.stop = start.astype(int), stop.astype(int)
        else:
            left, right = read_columns(input_file_name, 2)
            idmap.leftids, idmap.rightids = as_ids(left), as_ids(right)
            idmap.start = numpy.empty(idmap.rightids.shape, '|O')
            idmap.stop = numpy.empty(idmap.rightids.shape, '|O')
        if not idmap.validate():
            raise Exception('Wrong format of IdMap')
        return idmap
