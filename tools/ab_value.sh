#!/bin/bash
# Same-box alternation of run-time settings, printing only the headline value (for settings that break the kernel timing):
#   tools/ab_value.sh <rounds> "<label>:<ENV=V ...>" ...
rounds=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %8.1f MSM/s  (without prewarm %.1f)' % ('$label', d['value'], d['value_without_prewarm']))"
  done
done
