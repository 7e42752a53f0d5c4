cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "layernorm or conv_ln_block or large or Large or sat" 2>&1 | tail -2
for s in 1 0 1 0; do echo "== WAVLM_LN_GELU_TAB=$s"; WAVLM_LN_GELU_TAB=$s python tools/ln_conv_bench.py 2>&1 | grep -v amdgpu | grep -i "fwd" | head -4; done > gpurun_out/r05_ln_gelu_tab.txt 2>&1; cat gpurun_out/r05_ln_gelu_tab.txt
for s in 1 0 1 0; do WAVLM_LN_GELU_TAB=$s python bench.py --config large --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tab $s', d['ms_per_step'])"; done | tee -a gpurun_out/r05_ln_gelu_tab.txt
