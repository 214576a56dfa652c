"""Same-box A/B of builds of the library INSIDE the product schedule (two whole batches in flight, Xtractor.submit / collect):
    python scripts/ab_pipelined.py [--b1] libA.so libB.so ...        ("HEAD" = the shipped library)
Every build runs in its own child process, the builds are visited round-robin `--rounds` times (boxes drift by a percent within minutes:
A B A B, not A A B B); per build: min and median of rounds x 5 regions of 20 steps (ms per batch of 256), the one-forward-at-a-time
figure beside it, and with --b1 the batch-1 latency (ms per 4-s utterance, 200 forwards).  Boxes differ by +-3 %: a change is judged on ONE box."""
import json, os, statistics, subprocess, sys
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
def region(n):
    pend = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        pend.append(m.submit(wavs[i %% 5]))
        if len(pend) == m.pipeline_depth: m.collect(pend.pop(0))
    while pend: m.collect(pend.pop(0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def plain(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): m(wavs[i %% 5], is_eval=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
region(6)
out = {"pipelined": [region(20) for _ in range(5)]}
plain(4)
out["plain"] = [plain(20) for _ in range(3)]
if %r:
    one = wavs[0][:1].contiguous()
    plain_b1 = lambda n: [m(one, is_eval=True) for _ in range(n)]
    plain_b1(20); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plain_b1(200); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200 * 1e3)
    out["b1"] = ts
import json; print("RESULT " + json.dumps(out))
'''
args = [a for a in sys.argv[1:] if not a.startswith("--")]
b1 = "--b1" in sys.argv
rounds = 3
for a in sys.argv[1:]:
    if a.startswith("--rounds="): rounds = int(a.split("=")[1])
res = {l: {} for l in args}
for r in range(rounds):
    for lib in args:
        env = dict(os.environ)
        if lib != "HEAD": env["SIDEKIT_AMD_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-W", "ignore", "-c", CHILD % (ROOT, b1)], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(lib, "FAILED", out.stderr[-2000:], flush=True); continue
        for k, v in json.loads(line[-1][7:]).items(): res[lib].setdefault(k, []).extend(v)
        print(f"round {r} {lib}: " + " ".join(f"{k} {min(v):.3f}" for k, v in json.loads(line[-1][7:]).items()), flush=True)
def name(l): return l if l == "HEAD" else l.split("build_alt/")[-1].split("/")[0]
print("%-28s" % "build" + "".join("%24s" % k for k in ("pipelined min/med", "plain min/med", "b1 min/med")))
for lib in args:
    row = "%-28s" % name(lib)
    for k in ("pipelined", "plain", "b1"):
        v = res[lib].get(k)
        row += "%24s" % (f"{min(v):.3f} / {statistics.median(v):.3f}" if v else "-")
    print(row, flush=True)
