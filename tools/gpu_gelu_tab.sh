# usage (GPU box): bash tools/gpu_gelu_tab.sh -> gpurun_out/gelu_tab.txt: GELU-epilogue GEMMs, this build against tools/probe/lib/libwavlm_hip_prev.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/gelu_tab.txt; : > $O
timeout 600 python -m pytest tests -m gpu -x -q -k "gemm_pp or gemm_w4 or convstack or posconv or gemm_h2" 2>&1 | tail -2 >> $O
python tools/h2_ab.py time > /dev/null 2>&1
for rep in 1 2; do
for L in "" $PWD/tools/probe/lib/libwavlm_hip_prev.so; do
echo "== lib ${L:-this build}" >> $O
WAVLM_HIP_LIB=$L python tools/h2_ab.py time 2>/dev/null | grep "gelu" >> $O
done; done
bash tools/gpu_bench_ab.sh "WAVLM_HIP_LIB=" "WAVLM_HIP_LIB=$PWD/tools/probe/lib/libwavlm_hip_prev.so" 2 >> $O 2>&1
