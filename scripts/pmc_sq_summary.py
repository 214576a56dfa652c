"""Summarise the SQ counter passes of scripts/capture_profiles.sh per kernel: matrix-core busy cycles, the MfmaUtil figure as
rocprofv3 derives it (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 4 SIMDs x CUs) -- reads LOW for sub-0.3-ms dispatches
because GRBM_GUI_ACTIVE over-counts them, MI355X_MICROARCH.md DVFS note), wave-cycle split and the LDS bank-conflict share.

usage: python scripts/pmc_sq_summary.py gpurun_out/<tag>_pmc_mfma gpurun_out/<tag>_pmc_lds gpurun_out/<tag>_trace > profiles/<tag>_pmc_sq.json
"""
import collections, csv, glob, json, re, sys


def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(n):
    m = re.search(r"ConvCfg<sk::bf16_t, (\d+), (\d+), (\d+), (\d+), (\d+),.*?>, (\w+), (\d)>", n)
    if m:
        a = m.groups()
        return f"conv3x3 {a[0]}->{a[1]} s{a[2]} W{a[3]} TH{a[4]} form{a[6]}"
    return re.sub(r"\(.*", "", n).replace("void ", "")[-60:]


mfma, lds = collect(sys.argv[1]), collect(sys.argv[2])
dur = {}
for f in glob.glob(sys.argv[3] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"])
out = {}
for k, v in mfma.items():
    mean = {c: sum(x) / len(x) for c, x in v.items()}
    if not mean.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        continue
    rec = {"launches": len(v["SQ_VALU_MFMA_BUSY_CYCLES"]), "avg_us_under_profiler": dur.get(k, 0) / 1e3,
           "mfma_busy_cycles": mean["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_bf16_mops": mean.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
           "grbm_gui_active_sum_xcd": mean.get("GRBM_GUI_ACTIVE")}
    if mean.get("GRBM_GUI_ACTIVE"):
        rec["MfmaUtil_rocprof_definition"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8 * 4 * 256)
    if dur.get(k):
        # busy cycles per SIMD over the dispatch's wall time = the clock the matrix pipes would need to be 100 % busy
        rec["mfma_busy_cycles_per_simd_per_us"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (dur[k] / 1e3)
    l = lds.get(k)
    if l:
        lm = {c: sum(x) / len(x) for c, x in l.items()}
        rec.update({"lds_bank_conflict_share": lm["SQ_LDS_BANK_CONFLICT"] / max(lm["SQ_LDS_IDX_ACTIVE"], 1.0),
                    "wave_cycles_split": {c: lm[c] for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if c in lm}})
    out[short(k)] = rec
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"] * kv[1]["launches"])), sys.stdout, indent=1)
