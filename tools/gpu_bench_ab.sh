# usage: bash tools/gpu_bench_ab.sh "<ENV_A>" "<ENV_B>" [reps]   -> alternating bench.py runs of one build under two environments
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
A="$1"; B="$2"; R=${3:-2}
python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1   # box warm-up
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s ms_per_step %.2f  gpu_busy %.2f  host %.2f' % ('$E', d['ms_per_step'], d['gpu_busy_ms_per_step'], d['host_enqueue_ms_per_step']))"
  done
done
