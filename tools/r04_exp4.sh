#!/bin/bash
# round 4, experiment 4: instruction-cache counters of the lone accumulate kernel (2 waves shipped / 3 waves / lean 4 waves)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 1 3 2; do
  export MSM_AMD_ACC_VARIANT=$v
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH --output-format csv -d $O/r04d_ic_v$v -o pmc -- $R/metal-msm-gpu-acceleration_amd/gpu_profiler 20 1 gpu_resident 4 > $O/r04d_ic_v$v.out 2> $O/r04d_ic_v$v.err
  echo "variant $v icache rc=$?"
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES --output-format csv -d $O/r04d_sq_v$v -o pmc -- $R/metal-msm-gpu-acceleration_amd/gpu_profiler 20 1 gpu_resident 4 > $O/r04d_sq_v$v.out 2> $O/r04d_sq_v$v.err
  echo "variant $v sq rc=$?"
done
ls -R $O/r04d_ic_v1 | head
