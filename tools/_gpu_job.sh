# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_h.txt 2>&1; tail -3 gpurun_out/pytest_gpu_h.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --config extract --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_extract_h.txt; cut -c1-200 gpurun_out/bench_extract_h.txt
