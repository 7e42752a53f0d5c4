cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_pp3 or gemm_colsum or linear_ffn" 2>&1 | tail -3 > gpurun_out/r05_t13.txt
for i in 1 2; do
echo "== prefetch (default)"; timeout 300 python tools/gemm_aux_bound.py 2>&1 | grep -v amdgpu
echo "== no prefetch"; WAVLM_HIP_LIB=$PWD/tools/probe/lib/libwavlm_hip_probenopf.so timeout 300 python tools/gemm_aux_bound.py 2>&1 | grep -v amdgpu
done > gpurun_out/r05_gemm_aux_prefetch_ab.txt
cat gpurun_out/r05_t13.txt gpurun_out/r05_gemm_aux_prefetch_ab.txt
