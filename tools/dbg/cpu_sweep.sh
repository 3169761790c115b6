#!/bin/bash
# Window / task-shape sweep of the library's CPU MSM on the box's own cores: gpu_profiler 16 1 cpu 5
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
run() { "$@" 2>&1 | grep Average | sed 's/.*Time: //'; }
echo "default: $(run $P 16 1 cpu 5 --warmup 1)  $(run $P 16 1 cpu 5 --warmup 1)"
for c in 11 12 13 14; do for g in 1 2 3 4; do
  echo "c=$c groups=$g  $(MSM_AMD_HOST_WINDOW=$c MSM_AMD_HOST_GROUPS=$g run $P 16 1 cpu 5 --warmup 1) $(MSM_AMD_HOST_WINDOW=$c MSM_AMD_HOST_GROUPS=$g run $P 16 1 cpu 5 --warmup 1)"
done; done
for t in 1 2 4 8 16 32; do echo "threads=$t $(run $P 16 1 cpu 5 --warmup 1 --threads $t)"; done
MSM_AMD_HOST_TRACE=1 $P 16 1 cpu 1 --warmup 1 2>&1 | grep host_msm | tail -4
for l in 12 14 18 20; do echo "log=$l $(run $P $l 1 cpu 3 --warmup 1)"; done
nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name"
