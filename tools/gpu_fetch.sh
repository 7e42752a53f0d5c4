#!/bin/bash
# FETCH_SIZE-only PMC pass (see tools/gpu_pmc.sh): gpurun_out/pmc_<tag>/fetch.txt
TAG=${1:-x}; OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/F -o run -- python bench.py --steps 1 --warmup 1 --no-settle --no-cpu-baseline --no-roofline --no-secondary > $OUT/F.log 2>&1
F=$(find $OUT/F -name "*counter_collection.csv" | head -1); mkdir -p $OUT/f; cp $F $OUT/f/run_counter_collection.csv; cp $F $OUT/w_dummy.csv
mkdir -p $OUT/w; head -1 $F > $OUT/w/run_counter_collection.csv
python tools/pmc_traffic.py $OUT/f $OUT/w 2 > $OUT/fetch.txt 2>&1; rm -rf $OUT/F $OUT/f $OUT/w $OUT/w_dummy.csv
