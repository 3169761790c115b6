#!/bin/bash
# Product CPU MSM, three runs per size: average ms per MSM after one warm-up call
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
for log in 16 18 20; do
  echo -n "2^$log:"
  for r in 1 2 3; do
    $P $log 1 cpu 8 --warmup 1 --json 2>/dev/null | grep '^{' | python3 -c "import json,sys; print(' %.2f' % json.loads(sys.stdin.read())['avg_instance_ms'], end='')"
  done
  echo " ms"
done
