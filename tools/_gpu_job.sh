cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_large_e2e_gpu.py -q -k "layernorm or rowops or large or dropout_exact" 2>&1 | tail -3
python tools/ln_conv_bench.py 2>&1 | grep -v amdgpu
LN_ROWS=256000 python tools/ln_conv_bench.py 2>&1 | grep -v amdgpu
python tools/ln_bench.py 2>&1 | grep -v amdgpu | head -2
timeout 600 python bench.py --config large --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
