#!/bin/bash
set -e
OUT=$PWD/gpurun_out/prof_large; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python bench.py --config ${CFG:-large} --steps 2 --warmup 1 --no-settle --no-busy --no-cpu-baseline --no-roofline --no-secondary > $OUT/bench.log 2>&1 || true
TR=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python tools/prof_laststep.py $TR 45 > $OUT/summary.txt 2>&1 || true
tail -1 $OUT/bench.log >> $OUT/summary.txt; rm -rf $OUT/raw
