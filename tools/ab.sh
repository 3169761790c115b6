#!/bin/bash
# Same-box A/B of library builds (tools/build_variant.sh): tools/ab.sh <rounds> <name> [<name> ...]
# Alternates the headline bench over build_ab/libmsm_amd_<name>.so and prints MSM/s and the accumulate kernel's ms.
rounds=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    MSM_AMD_LIB=$R/build_ab/libmsm_amd_$v.so python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %8.1f MSM/s  accumulate %.4f ms  stage %s' % ('$v', d['value'], d['roofline']['avg_launch_ms'], d['stage_ms_per_msm']))"
  done
done
