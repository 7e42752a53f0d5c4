#!/bin/bash
# usage (GPU box, repo root): [CFG=base|large|sat_large] [SQ=0] tools/gpu_pmc.sh <tag>   -> gpurun_out/pmc_<tag>/...
# Separate rocprofv3 PMC passes (never combined with sys/hip traces) over `bench.py --steps 1 --warmup 1`:
#   FETCH_SIZE | WRITE_SIZE | SQ instruction / MFMA-busy counters.  Reduced by tools/pmc_traffic.py and tools/pmc_sq.py.
TAG=${1:-x}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CFG=${CFG:-base}
CMD="python bench.py --config $CFG --steps 1 --warmup 1 --no-settle --no-busy --no-cpu-baseline --no-roofline --no-secondary"
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]*(MFMA|VALU|BUSY|WAVE_CYCLES)[A-Z_0-9]*" | sort -u > $OUT/sq_counters_available.txt
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/$P -o run -- $CMD > $OUT/$P.log 2>&1 || echo "pass $P failed" >> $OUT/errors.txt
done
if [ "${SQ:-1}" = "1" ]; then
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/SQ -o run -- $CMD > $OUT/SQ.log 2>&1 || echo "pass SQ failed" >> $OUT/errors.txt
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/SQ2 -o run -- $CMD > $OUT/SQ2.log 2>&1 || echo "pass SQ2 failed" >> $OUT/errors.txt
fi
F=$(find $OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1)
mkdir -p $OUT/f $OUT/w; [ -n "$F" ] && cp $F $OUT/f/run_counter_collection.csv; [ -n "$W" ] && cp $W $OUT/w/run_counter_collection.csv
python tools/pmc_traffic.py $OUT/f $OUT/w 2 $OUT/gemm_hbm_traffic_$CFG.json > $OUT/pmc_hbm_traffic_$CFG.txt 2>&1 || true
for S in SQ SQ2; do
  C=$(find $OUT/$S -name "*counter_collection.csv" | head -1)
  [ -n "$C" ] && python tools/pmc_sq.py $C 2 > $OUT/pmc_$S.txt 2>&1 || true
done
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/SQ $OUT/SQ2 $OUT/f $OUT/w
