cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv0" 2>&1 | tail -3
(timeout 300 python tools/conv0_ln_bench.py; WAVLM_CONV0_BWD_MFMA=0 WAVLM_CONV0_FWD_MFMA=0 timeout 300 python tools/conv0_ln_bench.py) 2>&1 | grep -v amdgpu > gpurun_out/r05_conv0_ln_bench.txt; cat gpurun_out/r05_conv0_ln_bench.txt
rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o run -- python tools/conv0_ln_bench.py > /tmp/pf.log 2>&1
S=$(find /tmp/pf -name "*kernel_stats.csv" | head -1); python - $S <<'PY' > gpurun_out/r05_conv0_ln_prof.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
cat gpurun_out/r05_conv0_ln_prof.txt
