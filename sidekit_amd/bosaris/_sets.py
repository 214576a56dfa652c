"""Shared helpers for the trial-bookkeeping classes: hashed membership / index lookup instead of
the reference's per-element ``numpy.argwhere`` scans (O(N^2) -> O(N))."""
import numpy


def as_ids(x):
    """1-D object array of identifiers (the dtype the reference uses for modelset / segset)."""
    a = numpy.asarray(x)
    if a.dtype != object:
        a = a.astype(object)
    return a.reshape(-1)


def member_mask(items, pool):
    """``[item in pool for item in items]`` as a bool array."""
    pool = set(pool.tolist() if isinstance(pool, numpy.ndarray) else pool)
    return numpy.fromiter((i in pool for i in items), dtype=bool, count=len(items))


def first_index(container, wanted):
    """Index in ``container`` of the FIRST occurrence of every element of ``wanted`` (KeyError-free:
    raises IndexError like the reference's ``argwhere(...)[0][0]`` when an element is absent)."""
    first = {}
    for i, v in enumerate(container.tolist() if isinstance(container, numpy.ndarray) else container):
        first.setdefault(v, i)
    try:
        return numpy.fromiter((first[v] for v in wanted), dtype=numpy.int64, count=len(wanted))
    except KeyError as e:
        raise IndexError(f"index 0 is out of bounds for axis 0 with size 0 (identifier {e.args[0]!r} not found)")


def sorted_difference(a, b):
    """Elements of ``a`` not in ``b``, sorted (the reference's ``diff``)."""
    bs = set(b.tolist() if isinstance(b, numpy.ndarray) else b)
    out = [x for x in a if x not in bs]
    out.sort()
    return out


def read_columns(path, n):
    """Whitespace separated text file -> n object arrays (extra columns ignored)."""
    cols = [[] for _ in range(n)]
    with open(path, "r") as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            for c in range(n):
                cols[c].append(parts[c])
    return [numpy.array(c, dtype=object) for c in cols]
