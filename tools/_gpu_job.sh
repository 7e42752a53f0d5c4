cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or dropout_exact" 2>&1 | tail -5) > gpurun_out/r05_t10.txt 2>&1
L=tools/probe/lib/libwavlm_hip_probe
timeout 900 bash tools/gpu_ab.sh tilesrc "attn_" "python tools/attn_bench.py" unispeech_amd/lib/libwavlm_hip.so ${L}notilesrc.so > /dev/null 2>&1
grep -v "rocprofv3\|domain_stats\|^W2\|^E2" gpurun_out/ab_tilesrc.txt > gpurun_out/r05_ab_tilesrc.txt; cat gpurun_out/r05_t10.txt gpurun_out/r05_ab_tilesrc.txt
