# usage (GPU box): bash tools/gpu_reserved.sh [config] -> gpurun_out/reserved_cus_<config>.txt: what leaving n CUs to the RCCL kernels
# costs ONE rank (persistent GEMM grids of 256 - n blocks), alternating runs on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CFG=${1:-base}
O=gpurun_out/reserved_cus_$CFG.txt; : > $O
python bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
for rep in 1 2; do for R in 0 4 6 8; do
python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --reserved-cus $R 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('reserved %d  ms_per_step %.2f  gpu_busy %.2f' % (d['reserved_cus'], d['ms_per_step'], d['gpu_busy_ms_per_step']))" >> $O
done; done
