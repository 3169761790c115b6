#!/bin/bash
# the driver's GPU tier: python -m pytest tests -m gpu -x -q  (output kept under gpurun_out/)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r04}
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu -s --durations=12 > gpurun_out/${tag}_gputests.txt 2>&1
rc=$?
tail -30 gpurun_out/${tag}_gputests.txt
exit $rc
