#!/bin/bash
# usage (GPU box, repo root): tools/host_time8.sh <tag>  -> gpurun_out/host8_<tag>.txt
# Launch-thread time of a training step with 1 and with 8 concurrent Python processes on one host (the 8-GPU node runs one
# process per GPU: 8 launch threads share the host).  The processes share the one GPU of the box, so the measurement uses
# batch 1 (the same ~580 launches per step with next to no GPU work): "host enqueue ms" is then the launch thread's own cost.
TAG=${1:-x}; OUT=$PWD/gpurun_out/host8_$TAG.txt; TMP=$(mktemp -d)
echo "== 1 process" > $OUT
python tools/host_time.py 8 1 2>/dev/null | tail -1 >> $OUT
echo "== 8 concurrent processes" >> $OUT
for i in 0 1 2 3 4 5 6 7; do python tools/host_time.py 8 1 > $TMP/h$i.txt 2>/dev/null & done
wait
for i in 0 1 2 3 4 5 6 7; do tail -1 $TMP/h$i.txt >> $OUT; done
rm -rf $TMP
cat $OUT
