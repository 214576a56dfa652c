"""Per-shape conv kernel timing with ablation variants (GPU box)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd import _lib
if os.environ.get('SK_LIB'):  # A/B against another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ['SK_LIB'])
elif len(sys.argv) > 1 and any(int(x) >= 11 and int(x) not in (42, 43) for x in sys.argv[1].split(",")):
    # the alternative shapes (11-41, 44-47) and the layer-1 pair kernel (48) exist in the A/B build of the library only (csrc/Makefile `make ab`)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsidekit_amd_ab.so")
lib = _lib.lib()
torch.cuda.init(); torch.zeros(1).cuda()
names = list(_lib.PROF_NAMES[:11]) + ['X0(L1 3WG)', 'X1(L2 2WG)', 'X2(L3 2WG padded)', 'X3(L4 TH16)', 'X4(L3 r01 linear)', 'X5(L4 r01 linear)', 'X6(L2 r01 linear)', 'X7(L1 r01 linear)', 'X8(L3 GRID 32x32x16)', 'X9(L4 DENSE 32x32x16)', 'X10(L2 GRID 32x32x16)', 'X11(L1 GRID 32x32x16)', 'X12(L1 direct-store epilogue)', 'X13(L4A TH4 3WG)', 'X14(L3A TH2 3WG)', 'X15(L4A TH4 4WG)', 'X16(L3A TH2 4WG)', 'X17(L2A planar 2x8)', 'X18(L3A planar 4x4)', 'X19(L4A planar 8x2)', 'X20(L1 TH4 2 waves)', 'X21(L1 TH4 4 waves)', 'X22(L2 ring 3)', 'X23(L3 ring 3)', 'X24(L4 8 waves)', 'X25(L2A r03 row-major)', 'X26(L3A r03 row-major)', 'X27(L4A r03 row-major)', 'X28(L3A planar M16 occ3)', 'X29(L2A planar M16 occ2)', 'X30(L4A planar M16 occ3)']
Ts = {0: 401, 1: 401, 2: 401, 3: 401, 4: 201, 5: 201, 6: 201, 7: 101, 8: 101, 9: 101, 10: 51, 11: 401, 12: 201, 13: 101, 14: 51, 15: 101, 16: 51, 17: 201, 18: 401, 19: 101, 20: 51, 21: 201, 22: 401, 23: 401, 24: 101, 25: 201, 26: 101, 27: 201, 28: 401, 29: 201, 30: 101, 31: 401, 32: 401, 33: 201, 34: 101, 35: 51, 36: 401, 37: 201, 38: 101, 39: 201, 40: 401, 41: 101}
names += [f'shape{i}' for i in range(len(names), 49)]; names[48] = 'PAIR_L1 (conv2 + next conv1)'; Ts[48] = 401; names += ['X34(L3A TH8 1WG)']; Ts[49] = 201
shapes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4, 7, 10]
STAMPS = os.environ.get("STAMPS", "0") == "1"
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 6]
for sh in shapes:
    row = []
    for v in variants:
        ms = ctypes.c_float(0)
        ph = (ctypes.c_double * 16)()
        try:
            _lib.check(lib.sk_bench_conv(sh, 1, 256, Ts[sh], 20, v, ctypes.byref(ms), ph if STAMPS else None))
        except ValueError:   # e.g. a 1x1 shape has no statistics / residual form
            row.append(f"v{v}=n/a")
            continue
        txt = f"v{v}={ms.value*1e3:.0f}us"
        if STAMPS and sh == 48:
            txt += " [cyc: DMA issue %d | land+bar %d | k-loops A + park %d | bar + Y to image + shortcut fetch %d | bar + RMW + Y store %d | bar %d | k-loop B %d | bar + epi B + bar %d | edge + out %d | item %d | clock %.2f GHz]" % (
                ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], ph[8], ph[9], ph[9] / max(ph[10], 1.0) * 0.1)
        elif STAMPS:
            txt += " [cyc: issue %d | land+bar %d | kloop %d | bar+epi %d | bar %d | out %d | WG total %d | clock %.2f GHz]" % (
                ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[7], ph[7] / max(ph[6], 1.0) * 0.1)
        row.append(txt)
    print(names[sh], " ".join(row), flush=True)
