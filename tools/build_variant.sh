#!/bin/bash
# A/B builds of libmsm_amd.so: tools/build_variant.sh <name> "<extra hipcc flags>" ["<flags for k_accumulate.hip only>"]
#   ->  build_ab/libmsm_amd_<name>.so
# (select at run time with MSM_AMD_LIB=$PWD/build_ab/libmsm_amd_<name>.so).  `exp` with -DMSM_AMD_EXPERIMENTS is the
# build tests/test_experiments.py loads (made by __graft_entry__.build()).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2; accflags=${3:-}
work=$(mktemp -d)
P=metal-msm-gpu-acceleration_amd
mkdir -p "$work/$P" "$work/include" "$work/tools" "$ROOT/build_ab"
cp -rp "$ROOT/$P/csrc" "$work/$P/csrc"          # the same relative layout as the tree: the Makefile's ../../tools, ../../include
cp "$ROOT/include/msm_amd.h" "$work/include/"
cp -p "$ROOT/tools/gen_accumulate_asm.py" "$ROOT/tools/isa_counts.py" "$work/tools/"
rm -f "$work/$P"/csrc/*.o
make -C "$work/$P/csrc" -j8 ../libmsm_amd.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags" ACC_FLAGS="$accflags" >/dev/null
cp "$work/$P/libmsm_amd.so" "$ROOT/build_ab/libmsm_amd_$name.so"
rm -rf "$work"
echo "built build_ab/libmsm_amd_$name.so ($flags)"
