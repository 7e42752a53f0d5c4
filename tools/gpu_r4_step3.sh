cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_layer_fused_gpu.py tests/test_fp16_recipe_gpu.py "tests/test_kernels_gpu.py" -x -q 2>&1 | tail -5 > gpurun_out/r4_step3_tests.txt; cat gpurun_out/r4_step3_tests.txt
timeout 300 bash tools/gpu_prof.sh r4c 2>&1 | tail -3
head -1 gpurun_out/prof_r4c/summary.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench_c.txt 2>&1; tail -1 gpurun_out/r4_bench_c.txt | cut -c1-420
