#!/bin/bash
# SQ counters of the accumulate-kernel builds of the experiments library, each alone on the device (lone 2^20 calls):
#   tools/variants_pmc.sh <tag>   ->  gpurun_out/<tag>_variants_pmc.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r04}
O=$R/gpurun_out
export MSM_AMD_LIB=$R/build_ab/libmsm_amd_exp.so
cd /tmp && export TMPDIR=/tmp
for v in 1 3 5 4; do
  export MSM_AMD_ACC_VARIANT=$v
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/${tag}_vpmc_v$v -o pmc -- python3 $R/tools/quick_bench.py 20 4 > $O/${tag}_vpmc_v$v.out 2> $O/${tag}_vpmc_v$v.err
  echo "variant $v rc=$?"
done
python3 - "$O" "$tag" <<'PY'
import csv, collections, json, sys, glob, os
O, tag = sys.argv[1], sys.argv[2]
names = {1: "shipped_2_waves", 3: "column_form_3_waves", 5: "lds_parked_column_form_4_waves", 4: "hand_allocated_5_waves"}
out = {"source": "tools/variants_pmc.sh: rocprofv3 --kernel-trace --pmc ... -- python3 tools/quick_bench.py 20 4 (lone 2^20 calls, the accumulate kernel alone on the device), per launch; build_ab/libmsm_amd_exp.so",
       "kernels": {}}
for v, name in names.items():
    f = glob.glob(os.path.join(O, f"{tag}_vpmc_v{v}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "accumulate_kernel" in k and "redo" not in k:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    d = {c: round(sum(x) / len(x)) for c, x in acc.items()}
    d["avg_launch_us_under_pmc"] = round(sum(dur) / max(1, len(dur)) / 1e3, 1)
    out["kernels"][name] = d
json.dump(out, open(os.path.join(O, f"{tag}_variants_pmc.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
