# round-6 final capture at HEAD (GPU box): profiles/ evidence + bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/capture_profiles.sh r06 > gpurun_out/r06_capture.log 2>&1; echo "capture rc $?"; tail -3 gpurun_out/r06_capture.log
python3 scripts/traffic_from_pmc.py gpurun_out/r06_pmc_hbm_bytes_bf16_b256.json > gpurun_out/r06_traffic.json 2>gpurun_out/r06_traffic.err; echo "traffic rc $?"
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; echo "bench rc $?"
python3 bench.py --arch xvector --dtype fp32 --batch 512 --ragged > gpurun_out/r06_bench_line_tdnn_config4.json 2> gpurun_out/r06_bench_tdnn.err; echo "bench tdnn rc $?"
python3 scripts/scoring_bench.py > gpurun_out/r06_scoring_bench.json 2> gpurun_out/r06_scoring.err; echo "scoring rc $?"
python3 scripts/latency_b1.py > gpurun_out/r06_latency_b1.txt 2>&1; echo "latency rc $?"
grep -o '"value": [0-9.]*' gpurun_out/r06_bench_line.json | head -3; grep "B=" gpurun_out/r06_latency_b1.txt
