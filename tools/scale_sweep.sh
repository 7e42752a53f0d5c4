#!/bin/bash
# usage (a node with N >= 2 MI355X): bash tools/scale_sweep.sh [N=8] [steps=20]
#   -> gpurun_out/scale_sweep/<tag>.json, one bench line per setting; read `data_parallel.comm_wait_ms_per_rank` (0 = the
#      gradient all-reduce is fully hidden behind backward) next to `value`.
# Sweeps what the build environment (one GPU per call) could not measure: the CU reservation of the persistent GEMM grids
# (WAVLM_DP_RESERVED_CUS: CUs left to the RCCL kernels while backward runs) and the bucket size (WAVLM_DP_BUCKET_MIB; xGMI is
# point-to-point, a ring all-reduce is bound by one ~153 GB/s link: few large buckets).  Reference seam:
# src/fairseq/distributed/legacy_distributed_data_parallel.py:132-165 (one blocking all-reduce after backward).
N=${1:-8}; STEPS=${2:-20}
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}; mkdir -p gpurun_out/scale_sweep
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 900 python bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline --no-roofline --dp-alt-pass never > gpurun_out/scale_sweep/$tag.log 2>&1
  tail -1 gpurun_out/scale_sweep/$tag.log > gpurun_out/scale_sweep/$tag.json
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/scale_sweep/%s.json" % tag))
    dp = d.get("data_parallel") or {}
    print("%-28s %9.1f audio-s/s  %6.2f ms/step  devices %s  wait ms %s  exposed last bucket %s ms  bus %s GB/s" % (
        tag, d["value"], d["ms_per_step"], dp.get("distinct_devices"), dp.get("comm_wait_ms_per_rank"),
        dp.get("exposed_ms_last_bucket_rank0"), dp.get("allreduce_bus_gb_s_rank0")))
except Exception as e:
    print(tag, "FAILED", e)
PY
}
for cu in 0 4 6 8 16; do run cus${cu}_bucket32 WAVLM_DP_RESERVED_CUS=$cu WAVLM_DP_BUCKET_MIB=32; done
for mb in 16 64; do run cus6_bucket${mb} WAVLM_DP_RESERVED_CUS=6 WAVLM_DP_BUCKET_MIB=$mb; done
# the channel cap is a hypothesis (one channel = one workgroup = one CU; 332 MB per rank and step through 6 channels is an untested
# bandwidth assumption): reserved CUs fixed at 6, RCCL limited to 2 / 4 / 6 / 8 / 12 / 16 channels and left alone ("unset")
for ch in 2 4 6 8 12 16; do run cus6_channels${ch} WAVLM_DP_RESERVED_CUS=6 NCCL_MAX_NCHANNELS=$ch; done
run cus6_channels_unset WAVLM_DP_RESERVED_CUS=6 NCCL_MAX_NCHANNELS=
for n in 1 2 4; do [ $n -lt $N ] && N_SAVE=$N && N=$n && run n${n}_default WAVLM_DP_RESERVED_CUS=6 && N=$N_SAVE; done
