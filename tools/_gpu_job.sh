cd $GRAFT_REPO_ROOT; bash tools/gpu_final.sh d
