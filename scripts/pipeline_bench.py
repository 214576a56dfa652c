"""End-to-end rate of wav files -> x-vectors on one GPU (GPU box): N synthetic 16-bit 4 s wav files on local disk through
(a) the reference's loop -- one file, one forward, one read-back at a time (extract_xvectors.py:130-150) -- and
(b) sidekit_amd.pipeline.StreamingExtractor (decode threads, length-sorted batches, pinned staging, copy stream).
Prints one JSON line."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, scipy.io.wavfile, torch
from sidekit_amd.nnet import Xtractor
from sidekit_amd.pipeline import StreamingExtractor, load_entry

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if len(sys.argv) > 3:      # A/B: granularity (samples) the workspace reservation rounds a batch's longest utterance up to (1 = exact, the behaviour up to round 5)
    Xtractor.reserve_round = int(sys.argv[3])
d = tempfile.mkdtemp(prefix="skwav_", dir="/tmp")
rs = numpy.random.RandomState(0)
t0 = time.perf_counter()
base = (rs.randn(80000) * 3000).astype(numpy.int16)
entries = []
for i in range(N):
    n = 64000 if i % 4 else int(rs.randint(48000, 80000))        # three quarters exactly 4 s, the rest 3-5 s
    p = os.path.join(d, f"u{i:06d}.wav")
    scipy.io.wavfile.write(p, 16000, numpy.roll(base, i)[:n])
    entries.append((f"u{i:06d}", p))
t_write = time.perf_counter() - t0
dev = torch.device("cuda", 0)
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
out = {"files": N, "seconds_of_audio": None, "write_s": t_write, "decode_workers": workers, "reserve_round": Xtractor.reserve_round}
# (a) one file at a time
n_a = min(N, 512)
m(torch.zeros(1, 64000, device=dev), is_eval=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k, p in entries[:n_a]:
    s, _ = load_entry(p)
    x = torch.from_numpy(s.astype(numpy.float32) / 32768.0)[None].to(dev)
    e = m(x, is_eval=True)[1].cpu().numpy()
out["per_file_loop_files_per_s"] = n_a / (time.perf_counter() - t0)
# (b) streaming
for bs in (256,):
    ex = StreamingExtractor(m, batch_size=bs, window=8, workers=workers)
    dict(ex.run(iter(entries[:1024])))                      # warm-up: workspace, pinned buffers, page cache
    ex = StreamingExtractor(m, batch_size=bs, window=8, workers=workers)
    acc = {}
    def timed(name):                                     # where the host time goes (wall time inside each stage, threads overlap)
        fn = getattr(ex, name)
        def wrap(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            if name == "_collect": r = list(r)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
            return r
        setattr(ex, name, wrap)
    for name in ("_window_items", "_stage", "_launch", "_collect"): timed(name)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = dict(ex.run(iter(entries)))
    dt = time.perf_counter() - t0
    out["stage_seconds"] = {k: round(v, 3) for k, v in acc.items()}
    out["total_seconds"] = round(dt, 3)
    assert len(got) == N
    out[f"streaming_b{bs}_files_per_s"] = N / dt
    out["padding_overhead"] = ex.stats["padded_samples"] / ex.stats["samples"] - 1.0
    out["seconds_of_audio"] = ex.stats["samples"] / 16000.0
# decode alone (what the host can feed)
import concurrent.futures
t0 = time.perf_counter()
with concurrent.futures.ThreadPoolExecutor(workers) as pool:
    tot = sum(s.shape[0] for s, _ in pool.map(load_entry, [p for _, p in entries]))
out["decode_only_files_per_s"] = N / (time.perf_counter() - t0)
print(json.dumps(out), flush=True)
for _, p in entries: os.remove(p)
os.rmdir(d)
