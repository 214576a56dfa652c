"""GPU box: socket power and shader clock while the benchmark forward runs (evidence for DESIGN.md section 5: the MFMA-bound layers are
power-limited).  A child process loops the bf16 B = 256 forward (or one convolution shape through sk_bench_conv) for a few seconds; this
process samples `rocm-smi` (reading needs no privileges) every ~0.2 s.

    python scripts/power_probe.py            # idle, whole forward, then the four stride-1 convolution shapes alone
"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FWD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
wavs = [0.1 * torch.randn(256, 64000, device=dev, generator=g) for _ in range(5)]
for _ in range(5): m(wavs[0], is_eval=True)
torch.cuda.synchronize(); print("READY", flush=True)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < %f:
    for i in range(20): m(wavs[i %% 5], is_eval=True)
    torch.cuda.synchronize(); n += 20
print("DONE %%.3f ms per step" %% ((time.perf_counter() - t0) / n * 1e3), flush=True)
'''
CONV = r'''
import sys, time, ctypes, torch
sys.path.insert(0, %r)
from sidekit_amd import _lib
lib = _lib.lib(); torch.zeros(1).cuda()
ms = ctypes.c_float(0)
_lib.check(lib.sk_bench_conv(%d, 1, 256, %d, 20, %d, ctypes.byref(ms), None)); print("READY", flush=True)
t0 = time.perf_counter(); tot = 0.0; n = 0
while time.perf_counter() - t0 < %f:
    _lib.check(lib.sk_bench_conv(%d, 1, 256, %d, 400, %d, ctypes.byref(ms), None)); tot += ms.value; n += 1
print("DONE %%.1f us per launch" %% (tot / n * 1e3), flush=True)
'''


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--json"], capture_output=True, text=True).stdout
    try:
        d = json.loads(out)
    except Exception:
        return None
    c = d.get("card0", next(iter(d.values())) if d else {})
    rec = {}
    for k, v in c.items():
        kl = k.lower()
        if "power" in kl and "socket" in kl or "average graphics package power" in kl or kl.startswith("current socket"):
            rec["power_w"] = v
        elif kl.startswith("sclk clock speed"):
            rec["sclk"] = v
        elif kl.startswith("mclk clock speed"):
            rec["mclk"] = v
        elif "gpu use" in kl:
            rec["use"] = v
    return rec or {"raw": list(c.items())[:12]}


def run(label, code, seconds):
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    line = p.stdout.readline()
    while line and not line.startswith("READY"):
        line = p.stdout.readline()
    samples, t0 = [], time.time()
    while time.time() - t0 < seconds - 0.5:
        s = sample()
        if s:
            samples.append(s)
        time.sleep(0.15)
    rest = p.stdout.read()
    p.wait()
    done = [l for l in rest.splitlines() if l.startswith("DONE")]
    print(f"== {label}: {done[-1] if done else ''}")
    for s in samples[2:14]:
        print("   ", s)
    sys.stdout.flush()


if __name__ == "__main__":
    print("== idle:", sample(), flush=True)
    run("bf16 B=256 forward, two lanes", FWD % (ROOT, 6.0), 6.0)
    for name, shape, T, var in (("conv_L1 statistics form", 0, 401, 8), ("conv_L2 statistics form", 4, 201, 8), ("conv_L3 statistics form", 7, 101, 8),
                                ("conv_L4 statistics form", 10, 51, 8), ("conv_L3 with the MFMA loop skipped", 7, 101, 10), ("conv_L1 with the MFMA loop skipped", 0, 401, 10)):
        run(name, CONV % (ROOT, shape, T, var, 5.0, shape, T, var), 5.0)
