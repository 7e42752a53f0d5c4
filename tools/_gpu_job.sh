cd $GRAFT_REPO_ROOT; python tools/_dbg_attn.py 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_layer_fused_gpu.py -q -k "attention or dropout_exact or stored_p" 2>&1 | tail -4
