#!/bin/bash
# Round-4 A/B of the accumulate-kernel builds of the EXPERIMENTS library (build_ab/libmsm_amd_exp.so): parity of each
# variant, the kernel alone (lone 2^20 call), and the headline bench alternating over the variants on one box.
#   tools/ab_variants.sh <tag> [variants, default "1 3 2 5 4"]     (1 = the shipped kernel)
# Variants: 1 shipped (2 waves/SIMD, prefetch) | 3 column form held to 168 VGPRs (3 waves) | 2 register-lean product
# scanning (4 waves) | 5 column form + Y/ZZ/ZZZ parked in LDS (4 waves) | 4 hand-allocated statement (5 waves) + redo pass
# | 10 / 11 / 12 what-if timing kernels (gathers only / arithmetic on an L2-resident slice at 2 / 3 waves).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=${1:-ab}; vars=${2:-"1 3 2 5 4"}
export MSM_AMD_LIB=$R/build_ab/libmsm_amd_exp.so
O=gpurun_out
for v in $vars; do
  [ $v -ge 10 ] && continue
  MSM_AMD_ACC_VARIANT=$v timeout -k 10 300 python -m pytest tests/test_gpu_msm.py -x -q -m gpu > $O/${tag}_parity_v$v.txt 2>&1 || { echo "variant $v: parity FAILED"; tail -20 $O/${tag}_parity_v$v.txt; exit 1; }
  echo "variant $v: $(tail -1 $O/${tag}_parity_v$v.txt)"
done
for v in $vars; do echo "== lone 2^20 call, variant $v"; MSM_AMD_ACC_VARIANT=$v timeout -k 10 200 python tools/quick_bench.py 20 6 2>&1 | grep "c=17" | tail -1; done | tee $O/${tag}_lone.txt
specs=()
for v in $vars; do [ $v -ge 10 ] || specs+=("variant-$v:MSM_AMD_ACC_VARIANT=$v"); done
timeout -k 10 900 tools/ab_env.sh 2 "${specs[@]}" 2>&1 | tee $O/${tag}_ab.txt
