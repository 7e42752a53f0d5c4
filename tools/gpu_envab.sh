#!/bin/bash
# usage (GPU box, repo root): tools/gpu_envab.sh <tag> <kernel regex> "<command>" NAME:ENV=VAL[,ENV=VAL...] [NAME:... ...]
# Same-box A/B of ENVIRONMENT settings (library switches such as WAVLM_WGRAD_GROUPING): runs <command> under
# rocprofv3 --kernel-trace --stats once per setting, round-robin `REPS` times (default 2), and writes per run the matching
# kernel rows (calls, total ms, average us), their summed time and the command's last output line to gpurun_out/envab_<tag>.txt.
TAG=$1; PAT=$2; CMD=$3; shift 3
OUT=$PWD/gpurun_out/envab_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT.txt
for rep in $(seq 1 ${REPS:-2}); do
  for SPEC in "$@"; do
    N=${SPEC%%:*}; E=${SPEC#*:}
    rm -rf $OUT/raw
    env $(echo "$E" | tr ',' ' ') rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- $CMD > $OUT/$N.$rep.log 2>&1
    ST=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
    echo "== $N (rep $rep)  [$E]" >> $OUT.txt
    [ -n "$ST" ] && python - "$ST" "$PAT" >> $OUT.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        t = float(r["TotalDurationNs"]) / 1e6
        tot += t
        print("%-64s %6s calls  total %9.3f ms  avg %9.1f us" % (r["Name"][:64], r["Calls"], t, float(r["AverageNs"]) / 1e3))
print("matched total %.3f ms (whole run); all kernels %.3f ms" % (tot, sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
PY
    grep -a "^{" $OUT/$N.$rep.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('bench: %.2f ms/step, %.1f %s' % (d['ms_per_step'], d['value'], d['unit']))
except Exception as e: print('no bench line', e)" >> $OUT.txt
  done
done
rm -rf $OUT
cat $OUT.txt
