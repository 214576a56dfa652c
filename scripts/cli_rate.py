"""GPU box: end-to-end rate of the extraction CLI's main() (wav.scp -> StreamingExtractor -> ark / scp files, one rank) beside the bare streaming extractor on
the same files.  usage: python scripts/cli_rate.py [n_files]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, scipy.io.wavfile, torch
from sidekit_amd.nnet import Xtractor
from sidekit_amd.pipeline import StreamingExtractor
from sidekit_amd.bin import extract_xvectors

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = tempfile.mkdtemp(prefix="skcli_", dir="/tmp")
rs = numpy.random.RandomState(0)
base = (rs.randn(80000) * 3000).astype(numpy.int16)
entries = []
with open(os.path.join(d, "wav.scp"), "w") as f:
    for i in range(N):
        n = 64000 if i % 4 else int(rs.randint(48000, 80000))
        p = os.path.join(d, f"u{i:06d}.wav")
        scipy.io.wavfile.write(p, 16000, numpy.roll(base, i)[:n])
        entries.append((f"u{i:06d}", p))
        f.write(f"u{i:06d} {p}\n")
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
dict(StreamingExtractor(m, batch_size=256).run(iter(entries[:2048])))          # warm-up: workspace, page cache
out = {"files": N}
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = dict(StreamingExtractor(m, batch_size=256).run(iter(entries)))
    out.setdefault("streaming_files_per_s", []).append(round(N / (time.perf_counter() - t0)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    extract_xvectors.main(m, os.path.join(d, "wav.scp"), os.path.join(d, "out.scp"), "cuda:0", 16000, "", "", 256, "bf16", None, 8, False)
    out.setdefault("cli_main_files_per_s", []).append(round(N / (time.perf_counter() - t0)))
t0 = time.perf_counter()
u2w = extract_xvectors.read_wav_scp(os.path.join(d, "wav.scp"))
out["read_wav_scp_s"] = round(time.perf_counter() - t0, 4)
for w in (8, 16):
    t0 = time.perf_counter()
    extract_xvectors.precheck(m, [(k, ' '.join(v)) for k, v in u2w.items()], 16000, w)
    out[f"precheck_s_{w}_threads"] = round(time.perf_counter() - t0, 4)
print(json.dumps(out), flush=True)
for _, p in entries: os.remove(p)
for fn in os.listdir(d): os.remove(os.path.join(d, fn))
os.rmdir(d)
