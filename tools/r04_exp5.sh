#!/bin/bash
# round 4: the new GPU tests (config 4 workload over 8 contexts, pipelined multi-ctx, unsized bounded wait, cache verify, 2^24 ark affine)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_multi.py tests/test_gpu_robustness.py tests/test_gpu_bases_cache.py "tests/test_gpu_large.py::test_log24_ark_affine_dlog_identity" -x -q -m gpu -s > $O/r04f_newtests.txt 2>&1
rc=$?
tail -25 $O/r04f_newtests.txt
exit $rc
