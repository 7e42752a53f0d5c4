# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_misc_gpu.py -q -m gpu -s -k "force_dp or self_diagnosis" > gpurun_out/pytest_forcedp.txt 2>&1; grep "plain\|passed\|failed\|Error" gpurun_out/pytest_forcedp.txt | tail -8
