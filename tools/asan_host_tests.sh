#!/bin/bash
# Host-side AddressSanitizer + UBSan (trapping) run of the CPU test suite: the library's HOST code (C ABI, wire
# conversions, instance files, host bucket method, 64-bit Horner pass, test-op bodies) is instrumented, the device code
# is not (GPU sanitizers are unavailable on this pool).  Runs in the build container, no GPU needed.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
bash "$ROOT/tools/build_variant.sh" asan "-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fsanitize-trap=undefined -Xarch_host -fno-omit-frame-pointer"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
MSM_AMD_LIB=$ROOT/build_ab/libmsm_amd_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
