# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tests/gpu_checks.py attention 2>&1 | grep "FAIL\|group" | head
timeout 600 python tests/gpu_checks.py dropout_exact 2>&1 | grep "FAIL\|group" | head
: > gpurun_out/ab_dkv_rsm.txt
for rep in 1 2 3; do for LIB in tools/probe/lib/libwavlm_hip_probenorsm.so unispeech_amd/lib/libwavlm_hip.so; do
  rm -rf /tmp/prof_ab
  WAVLM_HIP_LIB=$PWD/$LIB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/attn_bench.py 0.1 > /tmp/ab.log 2>&1
  ST=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
  echo "== $LIB (rep $rep)" >> gpurun_out/ab_dkv_rsm.txt
  [ -n "$ST" ] && python - "$ST" >> gpurun_out/ab_dkv_rsm.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dkv_kernel" in r["Name"]:
        print("%-66s %6s calls  avg %9.1f us" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  grep "attention bwd" /tmp/ab.log >> gpurun_out/ab_dkv_rsm.txt
done; done
cat gpurun_out/ab_dkv_rsm.txt
