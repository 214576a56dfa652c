#!/bin/bash
# dynamic instruction mix of the front-end kernels (GPU box): scripts/frontend_pmc.sh <arch> <tag>
set -e -o pipefail
ARCH=${1:-halfresnet34}; TAG=${2:-fe}
OUT=gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_a -- python3 scripts/frontend_bench.py $ARCH > $OUT/${TAG}_pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_b -- python3 scripts/frontend_bench.py $ARCH > $OUT/${TAG}_pmc_b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_c -- python3 scripts/frontend_bench.py $ARCH > $OUT/${TAG}_pmc_c.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("a", "b", "c"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$OUT/${TAG}_pmc_%s/*/*counter_collection.csv" % p):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "fft" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / n[(k, c)], 1) for c, v in d.items()})
PY
