cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_layer_fused_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r4_fused_test.txt; cat gpurun_out/r4_fused_test.txt
timeout 300 python tools/host_time.py 8 > gpurun_out/r4_host_time.txt 2>&1; tail -3 gpurun_out/r4_host_time.txt
WAVLM_LAYER_FUSED=0 timeout 300 python tools/host_time.py 8 > gpurun_out/r4_host_time_composed.txt 2>&1; tail -3 gpurun_out/r4_host_time_composed.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench_a.txt 2>&1; tail -1 gpurun_out/r4_bench_a.txt | cut -c1-400
