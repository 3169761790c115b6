#!/bin/bash
# same binary, inversion by exponentiation (MSM_AMD_HOST_INV_FERMAT=1) against the binary GCD, alternating
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
one() { "$@" --json 2>/dev/null | grep '^{' | python3 -c "import json,sys; print(' %.2f' % json.loads(sys.stdin.read())['avg_instance_ms'], end='')"; }
for log in 16 18 20; do
  for round in 1 2 3; do
    echo -n "2^$log fermat:"; MSM_AMD_HOST_INV_FERMAT=1 one $P $log 1 cpu 8 --warmup 1
    echo -n "  bingcd:"; one $P $log 1 cpu 8 --warmup 1
    echo
  done
done
