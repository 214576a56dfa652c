"""GPU box: the Winograd inner-loop probe (scripts/probe_winograd.hip, built as build_alt/probe_winograd) beside the product's direct layer-3
convolution on the same problem size (256 x 101 x 20 positions, C = 128), each running alone for a few seconds while this process samples
`rocm-smi`: microseconds per launch, mean socket power while it runs, and joules per launch = the number that decides on a chip that runs
the MFMA-bound layers at its power limit (round-5 verdict item 2, measurements (a) and (c)).

    python scripts/winograd_probe_run.py [seconds per leg]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from power_probe import CONV, sample   # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0


def watts(s):
    v = s.get("power_w") if s else None
    try:
        return float(v)
    except (TypeError, ValueError):
        return None


IDLE = [240.0]


def leg(label, cmd, pattern):
    """Run `cmd` (it loops its kernel gap-free for SECONDS after printing READY), sample power meanwhile: us per launch, the PEAK-quartile
    mean of the samples (the legs that allocate between bursts have idle gaps; the upper quartile is the power while the kernel runs) and the
    energy per launch above idle."""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = p.stdout.readline()
    while line and not line.startswith("READY"):
        line = p.stdout.readline()
    samples = []
    while p.poll() is None:
        w = watts(sample())
        if w:
            samples.append(w)
        time.sleep(0.1)
    out = p.stdout.read()
    p.wait()
    m = re.search(pattern, out)
    us = float(m.group(1)) if m else float("nan")
    top = sorted(samples)[-max(1, len(samples) // 4):]
    w = sum(top) / len(top) if top else float("nan")
    print(f"{label:62s} {us:8.1f} us per launch   {w:7.0f} W while running   {us * (w - IDLE[0]) * 1e-3:7.2f} mJ per launch above idle   ({len(samples)} samples)", flush=True)
    return us, w


if __name__ == "__main__":
    idle = watts(sample())
    IDLE[0] = idle or 240.0
    print(f"idle: {idle} W; every leg runs alone for ~{SECONDS:g} s; layer-3 problem size: 256 utterances x 101 x 20 positions, 128 -> 128 channels, bf16", flush=True)
    probe = os.path.join(ROOT, "build_alt", "probe_winograd")
    for shape, var, name in ((7, 0, "direct conv3x3_kernel<L3> (product), plain form"), (7, 8, "direct conv3x3_kernel<L3> (product), statistics form"),
                             (7, 2, "direct, MFMA loop skipped (staging + epilogue only)")):
        code = CONV.replace(", 400, %d", ", 20000, %d") % (ROOT, shape, 101, var, SECONDS, shape, 101, var)    # 20 000 launches per call: seconds of gap-free kernel time
        leg(name, [sys.executable, "-W", "ignore", "-c", code], r"DONE ([0-9.]+) us per launch")
    for mode, name in ((0, "Winograd probe MODE 0: full transform per wave"), (1, "Winograd probe MODE 1: transform split over the wave pair"),
                       (2, "Winograd probe MODE 2: as 1, weights held in registers"), (3, "Winograd probe MODE 3: as 0, no MFMAs")):
        leg(name, [probe, str(int(SECONDS)), str(mode)], r"([0-9.]+) us per launch")
