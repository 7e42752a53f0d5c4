#!/bin/bash
# usage (GPU box, through gpurun): bash tools/gpu_final.sh  -> gpurun_out/{pytest_gpu_c.txt, prof_c/, pmc_c/, bench_*_c.txt}
# The round's closing artefacts of one build: the full GPU test suite, the kernel-trace profile, the PMC passes and the four
# bench lines (base with --live-traffic, large, sat_large, extract); copy what is to be judged into profiles/rNN/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_c.txt 2>&1; tail -3 gpurun_out/pytest_gpu_c.txt
timeout 600 bash tools/gpu_prof.sh c
timeout 900 bash tools/gpu_pmc.sh c
timeout 600 python bench.py --steps 20 --warmup 5 --live-traffic > gpurun_out/bench_base_c.txt 2>&1; tail -1 gpurun_out/bench_base_c.txt | cut -c1-220
for c in large sat_large extract; do timeout 400 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/bench_${c}_c.txt 2>&1; tail -1 gpurun_out/bench_${c}_c.txt | cut -c1-220; done
