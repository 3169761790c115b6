#!/bin/bash
# A/B builds of libmsm_amd.so: tools/build_variant.sh <name> "<extra hipcc flags>" ["<flags for k_accumulate.hip only>"]
#   ->  build_ab/libmsm_amd_<name>.so
# (select at run time with MSM_AMD_LIB=$PWD/build_ab/libmsm_amd_<name>.so).  Development aid.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2; accflags=${3:-}
work=$(mktemp -d)
mkdir -p "$work/pkg" "$work/include" "$ROOT/build_ab"
cp -r "$ROOT/metal-msm-gpu-acceleration_amd/csrc" "$work/pkg/csrc"
cp "$ROOT/include/msm_amd.h" "$work/include/"
rm -f "$work"/pkg/csrc/*.o
make -C "$work/pkg/csrc" -j8 ../libmsm_amd.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags" ACC_FLAGS="$accflags" >/dev/null
cp "$work/pkg/libmsm_amd.so" "$ROOT/build_ab/libmsm_amd_$name.so"
rm -rf "$work"
echo "built build_ab/libmsm_amd_$name.so ($flags)"
