cd $GRAFT_REPO_ROOT; bash tools/gpu_final.sh a 2>&1 | tail -30
