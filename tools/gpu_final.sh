#!/bin/bash
# usage (GPU box, through gpurun): bash tools/gpu_final.sh <tag>  -> gpurun_out/{pytest_gpu_<tag>.txt, prof_<tag>/, pmc_<tag>/, bench_*_<tag>.txt}
# The round's closing artefacts of one build: the full GPU test suite, the kernel-trace profile, the PMC passes and the four
# bench lines (base with --live-traffic, large, sat_large, extract); copy what is to be judged into profiles/rNN/.
TAG=${1:-x}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_$TAG.txt 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.txt
timeout 600 bash tools/gpu_prof.sh $TAG
timeout 900 bash tools/gpu_pmc.sh $TAG
timeout 600 python bench.py --steps 20 --warmup 5 --live-traffic > gpurun_out/bench_base_$TAG.txt 2>&1; tail -1 gpurun_out/bench_base_$TAG.txt | cut -c1-220
for c in large sat_large extract; do timeout 400 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/bench_${c}_$TAG.txt 2>&1; tail -1 gpurun_out/bench_${c}_$TAG.txt | cut -c1-220; done
