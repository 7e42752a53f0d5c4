#!/bin/bash
# usage (GPU box, through gpurun): bash tools/gpu_final.sh <tag>  -> gpurun_out/{pytest_gpu_<tag>.txt, prof_<tag>/, pmc_<tag>/, bench_*_<tag>.txt}
# The round's closing artefacts of one build: the full GPU test suite, the kernel-trace profile + PMC passes of the headline
# config, the four bench lines (base and large with --live-traffic, sat_large, extract) and the Large kernel-trace summary;
# copy what is to be judged into profiles/rNN/.
TAG=${1:-x}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_$TAG.txt 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.txt
timeout 600 bash tools/gpu_prof.sh $TAG
timeout 900 bash tools/gpu_pmc.sh $TAG                      # base: FETCH / WRITE / SQ passes -> pmc_<tag>/gemm_hbm_traffic_base.json
CFG=large SQ=0 timeout 600 bash tools/gpu_pmc.sh ${TAG}_large   # large: traffic passes only -> pmc_<tag>_large/gemm_hbm_traffic_large.json
timeout 600 python bench.py --steps 20 --warmup 5 --live-traffic --no-secondary > gpurun_out/bench_base_$TAG.txt 2>&1; tail -1 gpurun_out/bench_base_$TAG.txt | cut -c1-260
timeout 600 python bench.py --config large --steps 10 --warmup 3 --live-traffic > gpurun_out/bench_large_$TAG.txt 2>&1; tail -1 gpurun_out/bench_large_$TAG.txt | cut -c1-260
for c in sat_large extract; do timeout 400 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/bench_${c}_$TAG.txt 2>&1; tail -1 gpurun_out/bench_${c}_$TAG.txt | cut -c1-260; done
CFG=large timeout 400 bash tools/gpu_prof_large.sh; cp gpurun_out/prof_large/summary.txt gpurun_out/summary_large_$TAG.txt 2>/dev/null
(for c in large sat_large extract; do tail -1 gpurun_out/bench_${c}_$TAG.txt; done) > gpurun_out/bench_lines_$TAG.jsonl
