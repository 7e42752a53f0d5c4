cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_e.txt 2>&1; tail -3 gpurun_out/pytest_gpu_e.txt
