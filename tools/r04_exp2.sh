#!/bin/bash
# round 4, experiment 2: the column-form accumulate kernel at three waves per SIMD (168 VGPRs)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
MSM_AMD_ACC_VARIANT=3 timeout -k 10 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_property.py -x -q -m gpu > $O/r04b_w3_tests.txt 2>&1 || { echo w3 tests failed; tail -30 $O/r04b_w3_tests.txt; exit 1; }
tail -3 $O/r04b_w3_tests.txt
for v in 1 3 2; do echo "== lone call, variant $v"; MSM_AMD_ACC_VARIANT=$v timeout -k 10 200 python tools/quick_bench.py 20 6 2>&1 | grep "c=17" | tail -2; done | tee $O/r04b_lone.txt
timeout -k 10 900 tools/ab_env.sh 2 "shipped-2w:MSM_AMD_ACC_VARIANT=1" "w3:MSM_AMD_ACC_VARIANT=3" "w3-lds2:MSM_AMD_ACC_VARIANT=3 MSM_AMD_ACC_LDS=20480" 2>&1 | tee $O/r04b_ab.txt
