#!/bin/bash
# round 4, experiment 1: instruction issue intervals in cycles; the register-lean accumulate kernel alone and in the pipeline
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout -k 10 300 tools/microbench/valu_peak > $O/r04a_valu_peak.txt 2>&1 || { echo valu_peak failed; tail -5 $O/r04a_valu_peak.txt; exit 1; }
echo "== valu_peak done"; cat $O/r04a_valu_peak.txt
MSM_AMD_ACC_VARIANT=2 timeout -k 10 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_property.py -x -q -m gpu > $O/r04a_lean_tests.txt 2>&1 || { echo lean tests failed; tail -30 $O/r04a_lean_tests.txt; exit 1; }
tail -3 $O/r04a_lean_tests.txt
timeout -k 10 900 tools/ab_env.sh 2 "shipped-2w:MSM_AMD_ACC_VARIANT=1" "3w:MSM_AMD_ACC_VARIANT=0" "lean-4w:MSM_AMD_ACC_VARIANT=2" "lean-3w-lds:MSM_AMD_ACC_VARIANT=2 MSM_AMD_ACC_LDS=13312" "lean-2w-lds:MSM_AMD_ACC_VARIANT=2 MSM_AMD_ACC_LDS=20480" 2>&1 | tee $O/r04a_ab.txt
for v in 1 0 2; do echo "== lone call, variant $v"; MSM_AMD_ACC_VARIANT=$v timeout -k 10 200 python tools/quick_bench.py 20 6 2>&1 | grep "c=17" | tail -2; done | tee $O/r04a_lone.txt
