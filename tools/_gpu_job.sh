# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/wgrad_fixup_cmp.py > gpurun_out/wgrad_fixup_cmp.txt 2>&1; tail -6 gpurun_out/wgrad_fixup_cmp.txt
timeout 300 python tests/gpu_checks.py gemm_grouped 2>&1 | tail -2
timeout 300 python tests/gpu_checks.py gemm_race 2>&1 | tail -2
(for v in 0 1 0 1; do WAVLM_WGRAD_FIXUP=$v python tools/wgrad_grouped_bench.py; done; for v in 0 1 0 1; do WAVLM_WGRAD_FIXUP=$v python tools/wgrad_grouped_bench.py 31968; done) 2>/dev/null | tee gpurun_out/ab_wgrad_fixup.txt
for v in 0 1 0 1; do WAVLM_WGRAD_FIXUP=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('FIXUP=$v', d['ms_per_step'], d['gpu_busy_ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/ab_wgrad_fixup.txt; done
