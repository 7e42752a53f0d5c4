cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or linear_ffn or convstack" 2>&1 | tail -3 > gpurun_out/r05_t16.txt
timeout 300 python tools/gemm_aux_bound.py 2>&1 | grep -v amdgpu > gpurun_out/r05_gemm_aux_bound2.txt
cat gpurun_out/r05_t16.txt gpurun_out/r05_gemm_aux_bound2.txt
for i in 1 2; do for L in unispeech_amd/lib/libwavlm_hip.so tools/probe/lib/libwavlm_hip_probenopre2.so; do echo "== $L"; WAVLM_HIP_LIB=$PWD/$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.2f  gpu_busy %.2f  gemm ms %.3f frac %.4f' % (d['ms_per_step'], d['gpu_busy_ms_per_step'], r['gemm_ms_per_step'], r['frac']))"; done; done > gpurun_out/r05_step_ab_auxpre.txt 2>&1
cat gpurun_out/r05_step_ab_auxpre.txt
