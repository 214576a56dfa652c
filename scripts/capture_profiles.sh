#!/bin/bash
# GPU-box capture of the evidence bench.py's roofline object is judged against (run through gpurun):
#   scripts/capture_profiles.sh <tag>            -> gpurun_out/<tag>_*  (copy the summaries into profiles/)
# Passes: kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE; PMC matrix-core / busy counters -- each in its own run,
# with `python3 bench.py` directly after `--` (no env/bash hop under the profiler).
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ARGS > $OUT/${TAG}_trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $ARGS > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $ARGS > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -- python3 $ARGS > $OUT/${TAG}_pmc_mfma.log 2>&1
echo "mfma done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_lds -- python3 $ARGS > $OUT/${TAG}_pmc_lds.log 2>&1
echo "lds done"
