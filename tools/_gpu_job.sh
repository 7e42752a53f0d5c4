cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "gemm or linear_ffn or convstack or large or Large or bias" 2>&1 | tail -2
lib() { [ $1 = new ] && echo $PWD/unispeech_amd/lib/libwavlm_hip.so || echo $PWD/tools/probe/lib/libwavlm_hip_var$1.so; }
: > gpurun_out/r05_gemm_bias_ab3.txt
for c in base sat_large; do for L in head new head new; do echo "== $c $L" >> gpurun_out/r05_gemm_bias_ab3.txt; WAVLM_HIP_LIB=$(lib $L) python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])" >> gpurun_out/r05_gemm_bias_ab3.txt; done; done
cat gpurun_out/r05_gemm_bias_ab3.txt
