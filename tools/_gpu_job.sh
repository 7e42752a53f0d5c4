# scratch: the command of the builder's latest gpurun call (see tools/gpu_final.sh for the round's closing artefacts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_misc_gpu.py -q -x -k "inference_keeps or extract or convstack or posconv" 2>&1 | tail -3
for c in 1 0 1 0; do WAVLM_EVAL_CACHE=$c python bench.py --config extract --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cache $c', d['ms_per_step'], d['gpu_busy_ms_per_step'], d['host_enqueue_ms_per_step'])"; done
CFG=extract bash tools/gpu_prof_large.sh; cp gpurun_out/prof_large/summary.txt gpurun_out/summary_extract_g.txt; head -14 gpurun_out/summary_extract_g.txt
