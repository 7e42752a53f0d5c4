# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/attn_dkv64_cmp.py > gpurun_out/dkv64_cmp.txt 2>&1; tail -6 gpurun_out/dkv64_cmp.txt
timeout 600 python tests/gpu_checks.py dropout_exact > gpurun_out/chk_dropout.txt 2>&1; grep "FAIL\|Error" gpurun_out/chk_dropout.txt | head; tail -1 gpurun_out/chk_dropout.txt
: > gpurun_out/ab_dkv64_probes.txt
for LIB in unispeech_amd/lib/libwavlm_hip.so tools/probe/lib/libwavlm_hip_probek1.so tools/probe/lib/libwavlm_hip_probek2.so tools/probe/lib/libwavlm_hip_probek3.so tools/probe/lib/libwavlm_hip_probek4.so tools/probe/lib/libwavlm_hip_probek12.so tools/probe/lib/libwavlm_hip_probek13.so; do
  rm -rf /tmp/prof_ab
  WAVLM_HIP_LIB=$PWD/$LIB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/attn_bench.py 0.1 > /tmp/ab.log 2>&1
  ST=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
  echo "== $LIB" >> gpurun_out/ab_dkv64_probes.txt
  [ -n "$ST" ] && python - "$ST" >> gpurun_out/ab_dkv64_probes.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dkv" in r["Name"] or "dq_kernel<true, true, false>" in r["Name"]:
        print("%-60s %6s calls  avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cat gpurun_out/ab_dkv64_probes.txt
for S in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmc_k; T=$(echo $S | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $S --kernel-trace --output-format csv -d /tmp/pmc_k -o run -- python tools/attn_bench.py 0.1 > /tmp/pmc.log 2>&1
  C=$(find /tmp/pmc_k -name "*counter_collection.csv" | head -1)
  [ -n "$C" ] && python tools/pmc_sq.py $C 1 | grep "kernel\|dkv\|dq_kernel<true, true, false>" > gpurun_out/pmc_dkv64_$T.txt
  cut -c1-220 gpurun_out/pmc_dkv64_$T.txt
done
