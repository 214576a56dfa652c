"""bench.py plumbing that needs no GPU: `--gpus N` started bare launches its own ranks (fresh child processes, gloo dry
run here), and the roofline object is stable when two kernel classes are within noise of each other."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1",
                          "--batch", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout   # rank 0 prints ONE JSON line, relayed by the launcher
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_launcher_propagates_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--arch", "nope"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0


def test_roofline_object_is_stable_for_co_dominant_classes():
    sys.path.insert(0, ROOT)
    import bench
    B, T = 256, 401
    per_class = {"conv_L1": 1.70, "conv_L2A": 0.19, "conv_L2": 1.33, "conv_L3A": 0.14, "conv_L3": 1.69, "conv_L4A": 0.13, "conv_L4": 0.80,
                 "frontend": 0.3}
    a = bench.roofline({"conv_L1": (1.700 * 20, 120), "conv_L3": (1.690 * 20, 220)}, B, T, "bf16", per_class)
    b = bench.roofline({"conv_L1": (1.690 * 20, 120), "conv_L3": (1.700 * 20, 220)}, B, T, "bf16", per_class)
    # whichever class wins the coin flip, both are reported with their own bound and fraction ...
    assert {a["kernel"], a["co_dominant"]["kernel"]} == {b["kernel"], b["co_dominant"]["kernel"]}
    by_kernel = lambda r: {r["kernel"]: r["bound"], r["co_dominant"]["kernel"]: r["co_dominant"]["bound"]}
    assert by_kernel(a) == by_kernel(b) == {"conv3x3_kernel<L1, bf16>": "hbm", "conv3x3_kernel<L3, bf16>": "mfma"}
    # ... and the time-weighted trunk fraction does not depend on the order at all
    assert a["trunk"] == b["trunk"] and 0.2 < a["trunk"]["frac_time_weighted"] < 1.0
    for r in (a, a["co_dominant"]):
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["unit"] in ("GB/s", "TFLOP/s")


def test_committed_bench_lines_carry_roofline_and_cpu_baseline():
    """The bench lines kept under profiles/ for this round (configs[1]: HalfResNet34 bf16 B = 256; configs[3]: TDNN fp32, 512 ragged
    utterances) are complete: the contract keys, a `roofline` object with a consistent fraction and its provenance, and a
    `cpu_baseline` with both core counts."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_bench_line*.json")))
    assert any("tdnn_config4" in p for p in paths), "profiles/r03_bench_line_tdnn_config4.json is missing"
    for path in paths:
        with open(path) as f:
            d = json.loads(f.read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (path, k)
        assert "workload" in d["config"] and "model" not in d["config"] and d["vs_baseline"] is None
        assert abs(d["value"] - d["config"]["batch_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["peak"] in (8000.0, 2500.0, 157.3) and "traffic" in r and "traffic_source" in r and r["measured_in"]
        if "tdnn_config4" in path:
            assert d["dtype"] == "fp32" and "configs[3]" in d["config"]["workload"] and r["bound"] == "mfma" and r["peak"] == 157.3
            assert abs(r["achieved"] - r["alg_flops_per_forward"] / (r["tdnn_gemm_ms_per_step"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["cores_all"] >= c["cores"] and c["value"] > 0 and c["sample"]
