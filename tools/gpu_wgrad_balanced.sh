# usage (GPU box): bash tools/gpu_wgrad_balanced.sh -> gpurun_out/wgrad_balanced.txt: the grouped weight-gradient launch of a Base block, one-round split
# against the balanced launch (WAVLM_WGRAD_STREAMK=1) over segment-cost / tail-placement settings (profiles/r04/ab_wgrad_balanced_launch.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/wgrad_balanced.txt; : > $O
python tools/wgrad_grouped_bench.py > /dev/null 2>&1
for rep in 1 2; do
WAVLM_WGRAD_STREAMK=0 python tools/wgrad_grouped_bench.py 2>/dev/null | tail -1 >> $O
for SP in 0 1; do for C in 8 20 32 48; do
WAVLM_WGRAD_STREAMK=1 WAVLM_SK_SPREAD=$SP WAVLM_SK_SEG_COST=$C python tools/wgrad_grouped_bench.py 2>/dev/null | tail -1 >> $O
done; done; done
