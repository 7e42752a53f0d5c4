cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/pq; rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -- python bench.py --steps 3 --warmup 2 --no-settle --no-busy --no-cpu-baseline --no-roofline > /dev/null 2>&1
TR=$(find /tmp/pq -name "*kernel_trace.csv" | head -1); python tools/prof_sequence.py $TR "at::native" > gpurun_out/r05_native_sequence.txt; wc -l gpurun_out/r05_native_sequence.txt
