# usage (GPU box): bash tools/gpu_sk.sh -> gpurun_out/sk.txt: balanced grouped weight gradients: parity, then same-box A/B of the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/sk.txt; : > $O
timeout 900 python -m pytest tests -m gpu -x -q -k "gemm_grouped or gemm_w4 or layer_fused or fallbacks" 2>&1 | tail -5 >> $O
bash tools/gpu_bench_ab.sh "WAVLM_WGRAD_STREAMK=0" "WAVLM_WGRAD_STREAMK=1" 2 >> $O 2>&1
python tools/gemm_step_table.py 2>/dev/null | head -8 >> $O
