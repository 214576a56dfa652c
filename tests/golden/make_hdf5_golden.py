"""Generate the HDF5 interchange fixtures under tests/golden/hdf5/ with the REAL reference writers.

Runs only in the build container, under /opt/conda/bin/python3.9 (the one interpreter here that has h5py -- 3.3.0 on
HDF5 1.10.6; it has no torch, so this script is separate from make_golden.py):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py

The reference's own files are imported unchanged (package shell + stand-ins for the absent soundfile / torch, the recipe of
SURVEY.md 8c): ``StatServer.write`` (sidekit/statserver.py:427-489), ``Ndx.write`` (bosaris/ndx.py:92-112), ``Key.write``
(key.py:128-149), ``Scores.write`` (scores.py:94-116), ``IdMap.write`` (idmap.py:84-116), ``write_plda_hdf5`` /
``write_norm_hdf5`` / ``write_matrix_hdf5`` (sidekit_io.py).  The arrays that went in are kept in ``expected.npz``.
"""
import importlib
import os
import sys
import types

import numpy

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "hdf5")
REF = "/root/reference"


def import_reference():
    pkg = types.ModuleType("sidekit")
    pkg.__path__ = [os.path.join(REF, "sidekit")]
    pkg.PARALLEL_MODULE, pkg.PARAM_TYPE, pkg.STAT_TYPE = 'multiprocessing', numpy.float32, numpy.float64
    sys.modules["sidekit"] = pkg
    for name in ("soundfile", "torch"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    mods = {m: importlib.import_module(m) for m in ("sidekit.bosaris", "sidekit.sidekit_io", "sidekit.statserver")}
    return mods


def main():
    mods = import_reference()
    bos, sio, sts = mods["sidekit.bosaris"], mods["sidekit.sidekit_io"], mods["sidekit.statserver"]
    rs = numpy.random.RandomState(11)
    os.makedirs(OUT, exist_ok=True)
    exp = {}

    # StatServer: 7 segments of 5 models, start/stop partly None (-> -1 on disk), 3 x 4 statistics
    s = sts.StatServer()
    s.modelset = numpy.array(["spk_b", "spk_a", "spk_b", "spk_c", "spk_d", "spk_e", "spk_a"], dtype="|O")
    s.segset = numpy.array([f"seg{i:02d}" for i in range(7)], dtype="|O")
    s.start = numpy.array([0, 150, None, 30, None, 7, 99], dtype="|O")
    s.stop = numpy.array([100, 300, None, 90, None, 77, 199], dtype="|O")
    s.stat0 = rs.rand(7, 3)
    s.stat1 = rs.randn(7, 12)
    s.write(os.path.join(OUT, "statserver.h5"))
    s.write(os.path.join(OUT, "statserver_prefix.h5"), prefix="enrol/")
    exp.update(ss_modelset=s.modelset.astype(str), ss_segset=s.segset.astype(str), ss_stat0=s.stat0, ss_stat1=s.stat1,
               ss_start=numpy.array([-1 if v is None else v for v in s.start]), ss_stop=numpy.array([-1 if v is None else v for v in s.stop]))
    # x-vector sized StatServer (what extract_embeddings hands to scoring): 40 x 256, compressed into several chunks
    x = sts.StatServer()
    n = 40
    x.modelset = numpy.array([f"id{i % 9:03d}" for i in range(n)], dtype="|O")
    x.segset = numpy.array([f"utt{i:04d}" for i in range(n)], dtype="|O")
    x.start = numpy.empty(n, dtype="|O")
    x.stop = numpy.empty(n, dtype="|O")
    x.stat0 = numpy.ones((n, 1))
    x.stat1 = rs.randn(n, 256)
    x.write(os.path.join(OUT, "xvectors.h5"))
    exp.update(xv_modelset=x.modelset.astype(str), xv_segset=x.segset.astype(str), xv_stat1=x.stat1)

    models = numpy.array(["m1", "m2", "m1", "m3", "m2", "m4"], dtype="|O")
    segs = numpy.array(["s1", "s1", "s2", "s3", "s3", "s2"], dtype="|O")
    ndx = bos.Ndx(models=models, testsegs=segs)
    ndx.write(os.path.join(OUT, "ndx.h5"))
    exp.update(ndx_modelset=ndx.modelset.astype(str), ndx_segset=ndx.segset.astype(str), ndx_trialmask=ndx.trialmask)
    key = bos.Key(models=models, testsegs=segs, trials=numpy.array(["target", "nontarget", "nontarget", "target", "nontarget", "target"], dtype="|O"))
    key.write(os.path.join(OUT, "key.h5"))
    exp.update(key_modelset=key.modelset.astype(str), key_segset=key.segset.astype(str), key_tar=key.tar, key_non=key.non)
    sc = bos.Scores()
    sc.modelset, sc.segset = ndx.modelset, ndx.segset
    sc.scoremask = ndx.trialmask
    sc.scoremat = rs.randn(*ndx.trialmask.shape)
    sc.write(os.path.join(OUT, "scores.h5"))
    exp.update(sc_scoremat=sc.scoremat, sc_scoremask=sc.scoremask)
    im = bos.IdMap()
    im.leftids = numpy.array(["spk1", "spk1", "spk2"], dtype="|O")
    im.rightids = numpy.array(["file_a", "file_b", "file_c"], dtype="|O")
    im.start = numpy.array([None, 10, 0], dtype="|O")
    im.stop = numpy.array([None, 250, 400], dtype="|O")
    im.write(os.path.join(OUT, "idmap.h5"))
    exp.update(im_leftids=im.leftids.astype(str), im_rightids=im.rightids.astype(str), im_start=numpy.array([-1, 10, 0]), im_stop=numpy.array([-1, 250, 400]))

    D, r = 24, 6
    mean, F, G = rs.randn(D), rs.randn(D, r), numpy.zeros((D, 0))
    A = rs.randn(D, D)
    Sigma = A.dot(A.T) / D + numpy.eye(D)
    sio.write_plda_hdf5((mean, F, G, Sigma), os.path.join(OUT, "plda.h5"))
    exp.update(plda_mean=mean, plda_F=F, plda_G=G, plda_Sigma=Sigma)
    sio.write_norm_hdf5(([mean, 2 * mean], [Sigma, 3 * Sigma]), os.path.join(OUT, "norm.h5"))
    sio.write_matrix_hdf5(F.astype(numpy.float32), os.path.join(OUT, "matrix.h5"))
    numpy.savez_compressed(os.path.join(OUT, "expected.npz"), **exp)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
