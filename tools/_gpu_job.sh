# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu_a.txt 2>&1; tail -5 gpurun_out/pytest_gpu_a.txt
timeout 900 python bench.py 2>gpurun_out/bench_a.err | tail -1 > gpurun_out/bench_a.json; cut -c1-300 gpurun_out/bench_a.json
