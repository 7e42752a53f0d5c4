cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/sk2.txt; : > $O
python tools/wgrad_grouped_bench.py > /dev/null 2>&1
for rep in 1 2; do
WAVLM_WGRAD_STREAMK=0 python tools/wgrad_grouped_bench.py 2>/dev/null | tail -1 >> $O
for SP in 0 1; do for C in 8 20 32 48; do
WAVLM_SK_SPREAD=$SP WAVLM_SK_SEG_COST=$C python tools/wgrad_grouped_bench.py 2>/dev/null | tail -1 >> $O
done; done; done
