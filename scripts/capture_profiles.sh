#!/bin/bash
# GPU-box capture of the evidence bench.py's roofline object is judged against (run through gpurun):
#   scripts/capture_profiles.sh <tag>            -> gpurun_out/<tag>_*  (copy the summaries into profiles/)
# Passes: kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE; PMC matrix-core / busy counters; PMC LDS -- each in its own run,
# with `python3 bench.py` directly after `--` (no env/bash hop under the profiler).  Every pass runs ONE forward at a time with SERIAL lanes
# (--pipeline 1 --lanes 1): in the two-lane product configuration kernels of the two half batches overlap and one kernel's duration or counter
# says nothing about that kernel; one extra trace of the two-lane run is kept for the record.
set -e -o pipefail
TAG=${1:-r05}
OUT=gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --pipeline 1 --lanes 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ARGS > $OUT/${TAG}_trace.log 2>&1
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_lanes2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --pipeline 1 --lanes 2 > $OUT/${TAG}_trace_lanes2.log 2>&1
echo "two-lane trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_pipelined -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/${TAG}_trace_pipelined.log 2>&1
echo "pipelined (product configuration) trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $ARGS > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $ARGS > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -- python3 $ARGS > $OUT/${TAG}_pmc_mfma.log 2>&1
echo "mfma done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_lds -- python3 $ARGS > $OUT/${TAG}_pmc_lds.log 2>&1
echo "lds done"
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_hbm_bytes_bf16_b256.json
python3 scripts/pmc_sq_summary.py $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_lds $OUT/${TAG}_trace > $OUT/${TAG}_pmc_sq_bf16_b256.json
cp $(ls $OUT/${TAG}_trace/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_bf16_b256.csv
cp $(ls $OUT/${TAG}_trace_lanes2/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_bf16_b256_two_lanes.csv
cp $(ls $OUT/${TAG}_trace_pipelined/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_bf16_b256_pipelined.csv
# wav files on disk -> x-vectors (PCM16 straight into the front-end): the rate, and a kernel trace that shows which kernels ran
python3 scripts/pipeline_bench.py 16384 14 > $OUT/${TAG}_pipeline_bench.json 2> $OUT/${TAG}_pipeline_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_pipeline -- python3 scripts/pipeline_bench.py 2048 14 > $OUT/${TAG}_trace_pipeline.log 2>&1
cp $(ls $OUT/${TAG}_trace_pipeline/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_pipeline.csv
echo "pipeline done"
