# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; bash tools/gpu_final.sh r6f
timeout 600 python bench.py > gpurun_out/bench_default_r6f.txt 2>gpurun_out/bench_default_r6f.err; tail -1 gpurun_out/bench_default_r6f.txt | cut -c1-300
