# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tests/gpu_checks.py dropout_exact > gpurun_out/chk_dropout.txt 2>&1; grep "FAIL\|Error" gpurun_out/chk_dropout.txt | head; tail -1 gpurun_out/chk_dropout.txt
timeout 600 python tests/gpu_checks.py attention > gpurun_out/chk_attention.txt 2>&1; grep "FAIL\|Error" gpurun_out/chk_attention.txt | head; tail -1 gpurun_out/chk_attention.txt
rm -rf /tmp/prof_ab
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/attn_bench.py 0.1 > gpurun_out/attn_bench_bits.log 2>&1
ST=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && python - "$ST" > gpurun_out/ab_fwdbits.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn" in r["Name"]:
        print("%-66s %6s calls  avg %9.1f us" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat gpurun_out/ab_fwdbits.txt; grep -v "^W2026\|amdgpu.ids" gpurun_out/attn_bench_bits.log | head -20
