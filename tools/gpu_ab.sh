#!/bin/bash
# usage (GPU box, repo root): tools/gpu_ab.sh <tag> <grep pattern> "<command>" <lib.so> [<lib.so> ...]
# Same-box A/B of library builds: runs <command> under rocprofv3 --kernel-trace --stats once per library (WAVLM_HIP_LIB),
# twice round-robin, and prints the matching rows of each run's kernel stats (name, calls, average ns) into
# gpurun_out/ab_<tag>.txt.  Box-to-box differences of 5-8 % on single kernels make cross-call comparisons unreliable.
TAG=$1; PAT=$2; CMD=$3; shift 3
OUT=$PWD/gpurun_out/ab_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT.txt
for rep in 1 2; do
  for LIB in "$@"; do
    N=$(basename $LIB .so)
    rm -rf $OUT/raw
    WAVLM_HIP_LIB=$PWD/$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- $CMD > $OUT/$N.$rep.log 2>&1
    ST=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
    echo "== $N (rep $rep)" >> $OUT.txt
    [ -n "$ST" ] && python - "$ST" "$PAT" >> $OUT.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print("%-70s %6s calls  avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    tail -3 $OUT/$N.$rep.log >> $OUT.txt
  done
done
rm -rf $OUT
