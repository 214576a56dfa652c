"""The library's build graph: every header an object was compiled from must trigger its rebuild.

Round 5 listed the prerequisites by hand and `se_gate_inl.h` (included by conv3x3.hip) was not among them: an edit to it left
`make` saying "Nothing to be done" and would have shipped a stale library to the GPU box.  The Makefile now includes the
compiler-written dependency files (-MMD -MP); this test touches each header under csrc/ (and the public one) and asks `make -n`
of both builds (the product and the -DSK_AB one) whether it would recompile."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sidekit_amd", "csrc")
BUILDS = {"product": ([], "."), "ab": (["ab"], "obj_ab")}


def _make_n(target):
    return subprocess.run(["make", "-C", CSRC, "-n"] + target, capture_output=True, text=True, check=True).stdout


def _users(objdir, header):
    """objects of a build whose dependency file names the header"""
    out = []
    for d in glob.glob(os.path.join(CSRC, objdir, "*.d")):
        words = open(d).read().replace("\\\n", " ").split()
        if any(os.path.normpath(os.path.join(CSRC, w.rstrip(":"))) == header for w in words):
            out.append(os.path.basename(d)[:-2])
    return out


def test_every_header_triggers_a_rebuild():
    if not glob.glob(os.path.join(CSRC, "*.d")):
        pytest.fail("no dependency files beside the objects: build with __graft_entry__.build() / make -C sidekit_amd/csrc")
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "sidekit_amd.h")]
    assert any(h.endswith("se_gate_inl.h") for h in headers)
    seen = set()
    for h in headers:
        st = os.stat(h)
        try:
            os.utime(h, None)      # touch
            for name, (target, objdir) in BUILDS.items():
                users = _users(objdir, h)
                if not users:
                    continue
                seen.add(h)
                plan = _make_n(target)
                for u in users:
                    assert f"{u}.o" in plan, f"{os.path.basename(h)} was touched but `make {' '.join(target)}` would not rebuild {u}.o:\n{plan}"
        finally:
            os.utime(h, ns=(st.st_atime_ns, st.st_mtime_ns))
    missing = [os.path.basename(h) for h in headers if h not in seen and (h.endswith("se_gate_inl.h") is False or glob.glob(os.path.join(CSRC, "obj_ab", "*.d")))]
    assert not missing, f"headers no object depends on (dead, or the dependency files do not see them): {missing}"
