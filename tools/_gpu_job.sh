# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fallbacks_gpu.py -q -m gpu 2>&1 | tail -12
