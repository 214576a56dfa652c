cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 240 python -m pytest tests/test_gpu_halfresnet.py -x -q -k "small_grid or profile_slots or batch_size or short_clips" > gpurun_out/r05g_pytest.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/r05g_pytest.log | cut -c1-300
for cfg in "0" "1" "3"; do
  echo "gate_prologue=$cfg"
  SIDEKIT_AMD_GATE_PROLOGUE=$cfg timeout -k 10 120 python -W ignore scripts/latency_b1.py 2>&1 | grep "bf16"
done > gpurun_out/r05g_latency_matrix.txt 2>&1
cat gpurun_out/r05g_latency_matrix.txt
