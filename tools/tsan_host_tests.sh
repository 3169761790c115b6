#!/bin/bash
# Host-side ThreadSanitizer run of the threaded HOST code that needs no GPU: the product's CPU MSM (csrc/host_msm.hip:
# window x point-group tasks, shared side accumulators, running sums) through tests/test_host_msm.py.  The device code
# is not instrumented.  Runs in the build container.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
bash "$ROOT/tools/build_variant.sh" tsan "-Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer -Xarch_host -g"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
cd "$ROOT"
MSM_AMD_LIB=$ROOT/build_ab/libmsm_amd_tsan.so LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 \
  python -m pytest tests/test_host_msm.py tests/test_sharding.py -x -q -p no:cacheprovider
MSM_AMD_LIB=$ROOT/build_ab/libmsm_amd_tsan.so LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 python - <<'PY'
import importlib, sys
sys.path.insert(0, ".")
from oracle import bn254_ref as o, c_oracle as co
m = importlib.import_module("metal-msm-gpu-acceleration_amd")
for log_n, threads in ((16, 8), (17, 5)):
    n = 1 << log_n
    pts, sc = co.gen_instance(o.SEED_BASE + 900 + log_n, n)
    assert o.decode_jacobian_mont_le(m.host_msm(sc, pts, n, threads)) == o.decode_jacobian_mont_le(co.msm_best(sc, pts, n))
    print(f"tsan: host_msm 2^{log_n} on {threads} threads ok")
PY
