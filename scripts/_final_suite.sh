cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -rx > gpurun_out/r06h_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06h_pytest.log
tail -4 gpurun_out/r06h_pytest.log
bash scripts/_final_r06.sh
