cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tests/gpu_checks.py conv_ln_block 2>&1 | grep -v "^ok" | tail -20
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "layernorm or large or Large or layer_norm or preln or sat or conv" 2>&1 | tail -3
for s in 1 0 1 0; do WAVLM_LN_SEG=$s timeout 600 python bench.py --config large --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-140; done
