cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/sk3.txt; : > $O
python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
run() { env $1 python bench.py $2 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-24s %-34s ms_per_step %.2f  gpu_busy %.2f' % ('$1', '$2', d['ms_per_step'], d['gpu_busy_ms_per_step']))" >> $O; }
for rep in 1 2; do
for E in WAVLM_WGRAD_STREAMK=0 WAVLM_WGRAD_STREAMK=1; do
run $E "--config base"
run $E "--config base --reserved-cus 6"
done; done
for E in WAVLM_WGRAD_STREAMK=0 WAVLM_WGRAD_STREAMK=1; do
run $E "--config large --reserved-cus 6"
done
