"""GPU box: where the small-grid tilings of conv2 (layers 3-4, csrc/conv3x3.hip) stop paying: ms per bf16 forward of B x 4 s with
SIDEKIT_AMD_SMALL_GRID=0 (never) and =2 (always), each in its own child process.  usage: python scripts/small_grid_sweep.py [B ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from sidekit_amd.nnet import Xtractor
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to("cuda:0").eval()
m.compute_dtype = "bf16"
for B in %r:
    wav = 0.1 * torch.randn(B, 64000, device="cuda:0")
    for _ in range(5): m(wav, is_eval=True)
    torch.cuda.synchronize(); best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100): m(wav, is_eval=True)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
    print("B=%%d %%.3f" %% (B, best), flush=True)
'''
Bs = [int(x) for x in sys.argv[1:]] or [1, 4, 8, 12, 16, 24, 32, 48, 64, 96]
res = {}
for mode in ("0", "2"):
    out = subprocess.run([sys.executable, "-W", "ignore", "-c", CHILD % (ROOT, Bs)], env=dict(os.environ, SIDEKIT_AMD_SMALL_GRID=mode), capture_output=True, text=True).stdout
    res[mode] = {int(l.split()[0][2:]): float(l.split()[1]) for l in out.splitlines() if l.startswith("B=")}
print("batch   product tilings   small-grid tilings   (ms per forward, bf16, 4 s)")
for B in Bs:
    print(f"{B:5d}   {res['0'].get(B, float('nan')):15.3f}   {res['2'].get(B, float('nan')):18.3f}")
