#!/bin/bash
# batch-size sweep of the vector CPU MSM (additions per shared inversion)
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
one() { "$@" --json 2>/dev/null | grep '^{' | python3 -c "import json,sys; print(' %.2f' % json.loads(sys.stdin.read())['avg_instance_ms'], end='')"; }
for log in 16 18 20; do
  for round in 1 2; do
    echo -n "2^$log"
    for b in 128 256 512 1024; do echo -n "  batch<=$b:"; MSM_AMD_HOST_BATCH=$b one $P $log 1 cpu 8 --warmup 1; done
    echo
  done
done
