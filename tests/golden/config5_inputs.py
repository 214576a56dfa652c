"""BASELINE config 5's synthetic trial set, shared by make_golden.py and the tests.

Data only -- no reference code: 250 synthetic speakers, ``emb = normalize(c_spk + noise * randn)``, 1000 enrolment models x 1000
test segments with a full trial mask (a trial is a target when both sides are the same synthetic speaker), and a DISJOINT PLDA
training set (other speakers, other seed).  ``numpy.random.RandomState`` streams are frozen by numpy's compatibility policy, so
the arrays regenerate bit for bit; the fixture stores digests of them anyway and the tests assert those first.
"""
import numpy

D = 256
N_SPK, NE, NT = 250, 1000, 1000
NOISE = 1.8                 # sets the cosine EER near 3 % (SURVEY: "tune noise for EER 1-5 %")
TRAIN_SPK, TRAIN_SESS = 400, 8
PLDA_RANK = 128


def _norm(x):
    return x / numpy.linalg.norm(x, axis=1, keepdims=True)


def trial_set(seed=0):
    """-> E (1000, 256) f64, T (1000, 256) f64, enrolment speaker (1000,), test speaker (1000,)."""
    rs = numpy.random.RandomState(seed)
    c = rs.randn(N_SPK, D)
    spk_e, spk_t = rs.randint(0, N_SPK, NE), rs.randint(0, N_SPK, NT)
    E = _norm(c[spk_e] + NOISE * rs.randn(NE, D))
    T = _norm(c[spk_t] + NOISE * rs.randn(NT, D))
    return E, T, spk_e, spk_t


def plda_training_set(seed=1):
    """-> X (3200, 256) f64, speaker label (3200,): speakers drawn from another stream, so disjoint from the trial set's."""
    rs = numpy.random.RandomState(seed)
    c = rs.randn(TRAIN_SPK, D)
    lab = numpy.repeat(numpy.arange(TRAIN_SPK), TRAIN_SESS)
    return _norm(c[lab] + NOISE * rs.randn(lab.shape[0], D)), lab


def ids(prefix, n):
    return numpy.array([f"{prefix}{i:04d}" for i in range(n)], dtype="|O")


def digest(a):
    """Order-sensitive summary of an array (sum, weighted sum, strided sample)."""
    a = numpy.asarray(a, dtype=numpy.float64).ravel()
    w = numpy.cos(numpy.arange(a.shape[0]) * 0.001)
    return numpy.concatenate([[a.sum(), (a * w).sum(), numpy.abs(a).max()], a[::max(1, a.shape[0] // 61)][:61]])
