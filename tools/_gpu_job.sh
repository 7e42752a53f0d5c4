cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for g in conv0 conv0_ln; do timeout 600 python tests/gpu_checks.py $g 2>&1 | grep -v "^ok" | tail -12; done
WAVLM_CONV0_FWD_MFMA=0 WAVLM_CONV0_BWD_MFMA=0 timeout 600 python tests/gpu_checks.py conv0_ln 2>&1 | grep -v "^ok" | tail -5
