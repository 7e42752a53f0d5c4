cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_misc_gpu.py tests/test_layer_fused_gpu.py tests/test_dp_gpu.py -q -x -k "self_diagnosis or deepcopy or dp_gpu or eight_ranks" 2>&1 | tail -25 > gpurun_out/r05_t11.txt; cat gpurun_out/r05_t11.txt
