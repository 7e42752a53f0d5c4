#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag>   -> gpurun_out/prof_<tag>/{summary.txt,kernel_stats.csv}
# rocprofv3 --kernel-trace --stats of a short bench run, condensed with tools/prof_laststep.py
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python bench.py --steps 3 --warmup 2 --no-settle --no-busy --no-cpu-baseline --no-roofline --no-secondary > $OUT/bench.log 2>&1 || true
TR=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
ST=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
python tools/prof_laststep.py $TR 70 > $OUT/summary.txt 2>&1 || true
python tools/prof_gaps.py $TR 25 > $OUT/gaps.txt 2>&1 || true
[ -n "$ST" ] && cp $ST $OUT/kernel_stats.csv
tail -1 $OUT/bench.log >> $OUT/summary.txt
rm -rf $OUT/raw
