"""Build guard for the round-3 two-lane hazard (profiles/r04_hazard_isa_diff.txt).

On MI355X / ROCm 7.2 the SLP vectoriser's packed-f32 instructions -- `v_pk_add_f32` / `v_pk_mul_f32` with `op_sel` / `neg_lo` /
`neg_hi` modifiers, issued back to back as a dependent chain -- gave wrong spectrum bins while another stream's kernel issued dense
bf16 MFMAs on the same SIMDs.  The library is therefore built without them.  This test disassembles the shipped
`libsidekit_amd.so` (llvm-objdump cross-disassembles gfx950 on a CPU-only host) and fails if the family comes back: a compiler bump,
a dropped flag, an explicit float2 expression, a new file.

The rule since round 5: NO packed-f32 arithmetic (`v_pk_add_f32` / `v_pk_mul_f32` / `v_pk_fma_f32`) anywhere in the library.  Rounds
3-4 kept two exceptions -- `stem_kernel`'s explicit two-element FMAs and `se_pre_kernel`, the one file built with SLP -- on the evidence
"never seen to misbehave" and a speed argument measured in a serial forward (13.7 vs 18.5 us per SE gate).  Judged inside the
pipelined product schedule on one box (scripts/ab_pipelined.py, profiles/r05_ab_scalar_forms.txt) the scalar forms cost nothing:
5.533 / 5.530 / 5.525 ms per batch of 256 (shipped scalar stem + SLP gate / both scalar / both packed), batch 1 0.671 ms for all
three -- so the exposure went and the exception with it.
"""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from sidekit_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK_F32 = re.compile(r"\b(v_pk_(?:add|mul|fma)_f32)\b(.*)")


def _disassemble(workdir):
    so = os.path.join(workdir, "lib.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([OBJDUMP, "--offloading", so], cwd=workdir, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(glob.glob(os.path.join(workdir, "lib.so.*gfx950")))
    assert objs, "no gfx950 code object found inside libsidekit_amd.so"
    for o in objs:
        yield o, subprocess.run([OBJDUMP, "-d", o], check=True, capture_output=True, text=True).stdout


def _scan(text):
    """-> list of (kernel symbol, mnemonic, operand text) for every packed-f32 arithmetic instruction."""
    out, kernel = [], "?"
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        m = PK_F32.search(line)
        if m:
            out.append((kernel, m.group(1), m.group(2).split("//")[0].strip()))
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_no_modified_packed_f32_in_the_shipped_library():
    with tempfile.TemporaryDirectory() as td:
        found, n_mfma = [], 0
        for _, dis in _disassemble(td):
            found += _scan(dis)
            n_mfma += len(re.findall(r"\bv_mfma_", dis))
    assert n_mfma > 10000, f"disassembly looks empty ({n_mfma} MFMA instructions): the guard would be vacuous"
    bad = [(kernel, op, operands, "modified packed add / mul: the round-3 hazard family" if (op != "v_pk_fma_f32" and any(t in operands for t in ("op_sel", "neg_lo", "neg_hi")))
            else "packed f32 arithmetic") for kernel, op, operands in found]
    assert not bad, "packed-f32 instructions are back in libsidekit_amd.so (build every file with -fno-slp-vectorize, no explicit float2 arithmetic):\n" + "\n".join(
        f"  {k}: {o} {a}   [{why}]" for k, o, a, why in bad[:20]) + (f"\n  ... {len(bad)} in all" if len(bad) > 20 else "")


def test_scanner_sees_the_hazardous_forms():
    """The parser itself: the round-3 sequence (profiles/r04_hazard_isa_diff.txt) and the forms rounds 3-4 tolerated are all seen."""
    sample = """
0000000000001900 <_ZN2sk21stft_power_fft_kernelENS_7FftArgsE>:
	v_pk_add_f32 v[14:15], v[14:15], v[16:17] neg_lo:[0,1] neg_hi:[0,1]   // 000000001A2C: D3B2400E 1802210E
	v_pk_mul_f32 v[16:17], v[12:13], v[18:19] op_sel:[0,1]
	v_pk_add_f32 v[20:21], v[12:13], v[16:17]
0000000000002900 <_ZN2sk13se_pre_kernelItLi64EEEvNS_6SeArgsE>:
	v_pk_fma_f32 v[2:3], s[4:5], v[6:7], v[2:3] op_sel_hi:[1,0,1]
	v_pk_add_f32 v[2:3], v[2:3], v[4:5]
"""
    got = _scan(sample)
    assert [g[1] for g in got] == ["v_pk_add_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_add_f32"]
    assert "neg_lo" in got[0][2] and "op_sel" in got[1][2] and got[2][2] == "v[20:21], v[12:13], v[16:17]"
    assert "stft_power_fft_kernel" in got[0][0] and "se_pre_kernel" in got[3][0]
