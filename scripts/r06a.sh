#!/bin/bash
# round-6 first GPU session: GPU suite, bench line, host-side worker sweep, layer-1 phase stamps (statistics vs residual form)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06a_pytest.log
tail -5 gpurun_out/r06a_pytest.log
python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err || exit 1
tail -c 600 gpurun_out/r06a_bench.json
for W in 14 8 4; do python scripts/pipeline_bench.py 16384 $W >> gpurun_out/r06a_pipeline.json 2>> gpurun_out/r06a_pipeline.err || exit 1; done
cut -c1-400 gpurun_out/r06a_pipeline.json
STAMPS=1 python scripts/conv_bench.py 0 8,16,9,17,10,18 > gpurun_out/r06a_stamps_l1.txt 2>&1 || exit 1
cat gpurun_out/r06a_stamps_l1.txt
