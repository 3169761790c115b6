#!/bin/bash
# same binary with and without an environment switch, alternating: tools/dbg/cpu_ab_env.sh VAR
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
one() { "$@" --json 2>/dev/null | grep '^{' | python3 -c "import json,sys; print(' %.2f' % json.loads(sys.stdin.read())['avg_instance_ms'], end='')"; }
for log in 16 18 20; do
  for round in 1 2 3; do
    echo -n "2^$log $1=1:"; env $1=1 $P $log 1 cpu 8 --warmup 1 --json 2>/dev/null | grep '^{' | python3 -c "import json,sys; print(' %.2f' % json.loads(sys.stdin.read())['avg_instance_ms'], end='')"
    echo -n "  default:"; one $P $log 1 cpu 8 --warmup 1
    echo
  done
done
