#!/bin/bash
# Same-box A/B of run-time settings of ONE library build: tools/ab_env.sh <rounds> "<label>:<ENV=V ENV=V ...>" ...
# Alternates the headline bench over the settings and prints MSM/s, the accumulate kernel's ms and the stage spans.
# (tools/ab.sh does the same over different builds.)  Development aid.
rounds=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %8.1f MSM/s  accumulate %.4f ms  stage %s' % ('$label', d['value'], d['roofline']['avg_launch_ms'], d['stage_ms_per_msm']))"
  done
done
