# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_h.txt 2>&1; tail -4 gpurun_out/pytest_gpu_h.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_default_h.txt 2>gpurun_out/bench_default_h.err; tail -1 gpurun_out/bench_default_h.txt | cut -c1-300
