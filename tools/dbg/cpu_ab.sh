#!/bin/bash
# A/B of the product CPU MSM: tools/dbg/old_bin (a previous build) against the tree's, alternating, same box
NEW=./metal-msm-gpu-acceleration_amd/gpu_profiler
OLD=./tools/dbg/old_bin/gpu_profiler
one() { "$@" --json 2>/dev/null | grep '^{' | python3 -c "import json,sys; print(' %.2f' % json.loads(sys.stdin.read())['avg_instance_ms'], end='')"; }
for log in 16 18 20; do
  for round in 1 2 3; do
    echo -n "2^$log old:"; one $OLD $log 1 cpu 8 --warmup 1
    echo -n "  new:"; one $NEW $log 1 cpu 8 --warmup 1
    echo
  done
done
