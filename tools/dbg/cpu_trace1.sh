#!/bin/bash
# phase-1 shares on ONE thread (no quota throttling in the picture) and on all
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
for log in 16 20; do
  for t in 1 16; do
    echo "== 2^$log, $t thread(s)"
    MSM_AMD_HOST_TRACE=1 $P $log 1 cpu 3 --threads $t 2>&1 | grep -E "thread 0|buckets|Average" | tail -3
  done
done
