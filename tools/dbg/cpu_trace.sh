#!/bin/bash
# Phase stamps of the product CPU MSM on the box's host cores
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
for log in 16 18 20; do
  echo "== 2^$log"
  MSM_AMD_HOST_TRACE=1 $P $log 1 cpu 4 2>&1 | grep -E "host_msm:|Average" | tail -8
done
