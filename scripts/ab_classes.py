"""Same-box A/B of two builds of the library, per kernel class: serial lanes, every class bracketed (3 x 10 steps each).
usage: python scripts/ab_classes.py libA.so libB.so   (each build runs in its own child process)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
wavs = [0.1 * torch.randn(256, 64000, device=dev, generator=g) for _ in range(5)]
m.set_lanes(1)
for _ in range(5): m(wavs[0], is_eval=True)
best = None
for rep in range(3):
    m.set_profile(True); m.get_profile(reset=True)
    for i in range(10): m(wavs[i %% 5], is_eval=True)
    torch.cuda.synchronize()
    p = {k: v[0] / 10 for k, v in m.get_profile(reset=True).items() if v[1]}
    best = p if best is None else {k: min(best[k], p[k]) for k in p}
m.set_profile(False)
m.set_lanes(2)
for _ in range(5): m(wavs[0], is_eval=True)
torch.cuda.synchronize()
t2 = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for i in range(20): m(wavs[i %% 5], is_eval=True)
    torch.cuda.synchronize()
    t2 = min(t2, (time.perf_counter() - t0) / 20 * 1e3)
best["two_lane_step"] = t2
import json; print("RESULT " + json.dumps(best))
''' % ROOT
res = {}
for lib in sys.argv[1:]:
    env = dict(os.environ, SIDEKIT_AMD_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(lib, "FAILED", out.stderr[-2000:]); continue
    res[lib] = json.loads(line[-1][7:])
keys = sorted({k for r in res.values() for k in r})
print("%-16s" % "class" + "".join("%28s" % os.path.basename(l)[-26:] for l in res))
for k in keys:
    print("%-16s" % k + "".join("%28.4f" % res[l].get(k, float("nan")) for l in res))
print("%-16s" % "sum(serial)" + "".join("%28.4f" % sum(v for k, v in res[l].items() if k != "two_lane_step") for l in res))
