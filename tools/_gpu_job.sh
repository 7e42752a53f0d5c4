cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dp_gpu.py -q -x -k "rccl" 2>&1 | tail -15
