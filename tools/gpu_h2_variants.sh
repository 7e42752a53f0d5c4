# usage (GPU box): bash tools/gpu_h2.sh  -> gpurun_out/h2_variants.txt: the gemm_h2 lab builds (tools/probe/build_probe.py h2:...) x start skew
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/h2_variants.txt; : > $O
L=$PWD/tools/probe/lib
python tools/h2_ab.py time > /dev/null 2>&1   # box warm-up
WAVLM_HIP_LIB=$L/libwavlm_hip_probeh2pipe.so python tools/h2_ab.py check 2>&1 | tail -1 >> $O
for V in h2old h2oldprio h2pipe h2pipeprio; do
  for S in 0 2 4; do
    echo "== $V skew $S" >> $O
    WAVLM_H2_SKEW=$S WAVLM_HIP_LIB=$L/libwavlm_hip_probe$V.so timeout 300 python tools/h2_ab.py time 2>&1 | grep -v Warn | tail -9 >> $O
  done
done
