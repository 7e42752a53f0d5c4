# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_b.txt 2>&1; tail -4 gpurun_out/pytest_gpu_b.txt
for v in 0 1 0 1; do WAVLM_ATTN_DBITS=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_dbits$v.json; python -c "
import json;d=json.load(open('gpurun_out/bench_dbits$v.json'));print('DBITS=$v', d['ms_per_step'], d['gpu_busy_ms_per_step'], [ (k['name'],k['avg_call_us']) for k in d['roofline']['kernels'][:2]])"; done
