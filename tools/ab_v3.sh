#!/bin/bash
# late round 4: loop shape / wide records A/B on one box.  EXPERIMENTS library: 1 = shipped loop shape, 7 = the shape of
# rounds 1-4, 8 = wide records.    tools/ab_v3.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=${1:-v3b}
O=$R/gpurun_out
export MSM_AMD_LIB=$R/build_ab/libmsm_amd_exp.so
timeout -k 10 300 python tools/ab_acc_kernel.py "1 7 8" 60 | tee $O/${tag}_kernel_alone.txt || exit 1
timeout -k 10 900 tools/ab_env.sh 3 "shipped-1:MSM_AMD_ACC_VARIANT=1" "r4-shape-7:MSM_AMD_ACC_VARIANT=7" "wide-8:MSM_AMD_ACC_VARIANT=8" 2>&1 | tee $O/${tag}_ab.txt || exit 1
cd /tmp && export TMPDIR=/tmp
for v in 1 7 8; do
  export MSM_AMD_ACC_VARIANT=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_vpmc_v$v -o pmc -- python3 $R/tools/quick_bench.py 20 4 > $O/${tag}_vpmc_v$v.out 2> $O/${tag}_vpmc_v$v.err
  python3 - "$O/${tag}_vpmc_v$v" $v <<'PY' | tee -a $O/${tag}_pmc.txt
import csv, collections, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "accumulate_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant", sys.argv[2], {c: round(sum(x) / len(x)) for c, x in acc.items()})
PY
done
