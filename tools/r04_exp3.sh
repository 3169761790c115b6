#!/bin/bash
# round 4, experiment 3: what bounds the accumulate kernel at ~1.05 ms whatever its occupancy?  gathers alone, arithmetic alone
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
export MSM_AMD_LIB=$R/build_ab/libmsm_amd_exp.so
for v in 1 10 11 12 3; do echo "== lone call, variant $v"; MSM_AMD_ACC_VARIANT=$v timeout -k 10 200 python tools/quick_bench.py 20 6 2>&1 | grep "c=17" | tail -2; done | tee $O/r04c_whatif.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/$O/r04c_list_avail.txt 2>&1
echo "list-avail rc=$?"
