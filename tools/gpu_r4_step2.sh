cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_layer_fused_gpu.py tests/test_large_e2e_gpu.py tests/test_bf16_e2e_gpu.py "tests/test_model_gpu.py::test_base_width_vs_oracle" -x -q -s 2>&1 | grep -v Warning | tail -60 > gpurun_out/r4_parity_b.txt; cat gpurun_out/r4_parity_b.txt
