cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_halfresnet.py tests/test_gpu_config1.py -x -q > gpurun_out/r05e_pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r05e_pytest.log | cut -c1-300
for cfg in "0 0 -" "1 0 -" "1 0 42=44;43=45" "1 0 43=46" "1 1 -" ; do set -- $cfg
  echo "small_grid=$1 gate_prologue=$2 shape_map=$3"
  if [ "$3" = "-" ]; then SIDEKIT_AMD_SMALL_GRID=$1 SIDEKIT_AMD_GATE_PROLOGUE=$2 python -W ignore scripts/latency_b1.py 2>&1 | grep "bf16"
  else SIDEKIT_AMD_SHAPE_MAP="$3" SIDEKIT_AMD_SMALL_GRID=$1 SIDEKIT_AMD_GATE_PROLOGUE=$2 python -W ignore scripts/latency_b1.py 2>&1 | grep "bf16"; fi
done > gpurun_out/r05e_latency_matrix.txt 2>&1
cat gpurun_out/r05e_latency_matrix.txt
