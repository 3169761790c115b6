#!/bin/bash
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
run() { "$@" 2>&1 | grep Average | sed 's/.*Time: //'; }
echo "ifma default: $(run $P 16 1 cpu 5 --warmup 1)  $(run $P 16 1 cpu 5 --warmup 1)"
echo "scalar default: $(MSM_AMD_HOST_NO_IFMA=1 run $P 16 1 cpu 5 --warmup 1)"
MSM_AMD_HOST_TRACE=1 $P 16 1 cpu 1 --warmup 1 2>&1 | grep host_msm | tail -4
MSM_AMD_HOST_TRACE=1 $P 16 1 cpu 1 --warmup 1 --threads 1 2>&1 | grep host_msm | tail -4
MSM_AMD_HOST_NO_IFMA=1 MSM_AMD_HOST_TRACE=1 $P 16 1 cpu 1 --warmup 1 --threads 1 2>&1 | grep host_msm | tail -4
for c in 10 11 12 13; do for g in 1 2 3 4; do
  echo "c=$c groups=$g  $(MSM_AMD_HOST_WINDOW=$c MSM_AMD_HOST_GROUPS=$g run $P 16 1 cpu 5 --warmup 1) $(MSM_AMD_HOST_WINDOW=$c MSM_AMD_HOST_GROUPS=$g run $P 16 1 cpu 5 --warmup 1)"
done; done
for t in 1 2 4 8 16; do echo "threads=$t $(run $P 16 1 cpu 5 --warmup 1 --threads $t)"; done
for l in 12 14 18 20; do echo "log=$l $(run $P $l 1 cpu 3 --warmup 1)"; done
tools/dbg/ifma_bin
