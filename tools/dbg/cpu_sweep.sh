#!/bin/bash
# A/B of the host MSM task shapes on the box's own cores: gpu_profiler 16 1 cpu 5
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
run() { "$@" 2>&1 | grep Average | sed 's/.*Time: //'; }
for rep in 1 2; do
for c in 12 13 14; do for r in 1 2 3 4 6; do
  a=$(MSM_AMD_HOST_WINDOW=$c MSM_AMD_HOST_RANGES=$r run $P 16 1 cpu 5 --warmup 1)
  b=$(MSM_AMD_HOST_BY_POINTS=1 MSM_AMD_HOST_WINDOW=$c MSM_AMD_HOST_RANGES=$r run $P 16 1 cpu 5 --warmup 1)
  echo "c=$c ranges=$r  by_buckets=$a  by_points=$b"
done; done; done
nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|Thread|Core|Socket"; taskset -p $$
