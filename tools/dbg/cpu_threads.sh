#!/bin/bash
# thread-count sweep of the product CPU MSM (the grant is a 16-CPU quota over 256 visible CPUs)
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
one() { "$@" --json 2>/dev/null | grep '^{' | python3 -c "import json,sys; print(' %.2f' % json.loads(sys.stdin.read())['avg_instance_ms'], end='')"; }
for log in 16 20; do
  echo -n "2^$log:"
  for t in 1 2 4 8 12 14 16 20 24 32; do echo -n "  T=$t"; one $P $log 1 cpu 6 --warmup 1 --threads $t; done
  echo
done
cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -c processor /proc/cpuinfo; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "throttled|nr_periods"
