"""CPU-only tests (no GPU): the C-ABI library loads and exports every declared symbol, and the host-side
logic that needs no device (final_accumulation, window policy, byte accounting, error paths)."""
import ctypes
import os
import random
import re

import pytest

from oracle import bn254_ref as o
from helpers import decode_be32_affine, rand_jac, rand_point

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(msm_pkg):
    L = msm_pkg.lib()
    hdr = open(os.path.join(ROOT, "include", "msm_amd.h")).read()
    declared = set(re.findall(r"\b(msm_amd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(msm_pkg.EXPORTS)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/msm_amd.h but not exported"
    assert L.msm_amd_version().startswith(b"msm_amd")


def test_status_strings(msm_pkg):
    L = msm_pkg.lib()
    for st in range(6):
        assert len(L.msm_amd_strerror(st)) > 0
    assert b"InputError" in L.msm_amd_strerror(msm_pkg.INPUT_ERROR)


def test_window_policy(msm_pkg):
    """Window policy: the reference's 3 below 32 points (msm.rs:137-138); from there the measured optimum per size
    instead of the reference's constant 15 (msm.rs:140) -- results do not depend on the window."""
    L = msm_pkg.lib()
    assert L.msm_amd_auto_window_size(1) == 3 and L.msm_amd_auto_window_size(31) == 3     # msm.rs:137-138
    assert L.msm_amd_auto_window_size(32) == 5
    assert L.msm_amd_auto_window_size(1 << 16) == 15 and L.msm_amd_auto_window_size(1 << 18) == 15
    assert L.msm_amd_auto_window_size(1 << 19) == 16                                                  # u32 digits
    assert L.msm_amd_auto_window_size(1 << 20) == 17 and L.msm_amd_auto_window_size(1 << 24) == 17
    ws = [L.msm_amd_auto_window_size(1 << k) for k in range(5, 25)]
    assert ws == sorted(ws)
    # none of the chosen windows leaves a top digit of fewer than 4 bits (all points of that window in <= 8 buckets)
    for c in set(ws):
        assert 254 - (254 // c) * c >= 4 or 254 % c == 0
    # a lone call (one instance, nothing else in flight) has its own policy: latency, not GPU work per MSM
    lone = [L.msm_amd_auto_window_size_lone(1 << k) for k in range(5, 25)]
    assert L.msm_amd_auto_window_size_lone(31) == 3 and lone == sorted(lone)
    assert L.msm_amd_auto_window_size_lone(1 << 12) == 8 and L.msm_amd_auto_window_size_lone(1 << 16) == 15
    assert L.msm_amd_auto_window_size_lone(1 << 18) == 15 and L.msm_amd_auto_window_size_lone(1 << 19) == 17
    for c in set(lone):
        assert 254 - (254 // c) * c >= 4 or 254 % c == 0


def test_algorithmic_bytes_match_survey(msm_pkg):
    """A(2^20) = 1.424 GB, A3(2^20) = 1.337 GB at c = 15."""
    L = msm_pkg.lib()
    n, W, totB = 1 << 20, 17, 557039
    assert L.msm_amd_algorithmic_bytes(n, 15, 0) == 32 * n + 72 * n * W + 2 * 96 * totB
    assert L.msm_amd_algorithmic_bytes(n, 15, 1) == 72 * n * W + 96 * totB


@pytest.mark.parametrize("W,c", [(1, 3), (2, 5), (17, 15), (5, 7), (85, 3)])
def test_final_accumulation_host(msm_pkg, W, c):
    """final_accumulation.rs:5-40 (host Horner), including window_num == 1 which the reference gets wrong."""
    rng = random.Random(W * 100 + c)
    pj = [rand_jac(rng, rand_point(rng)) for _ in range(W)]
    if W > 2:
        pj[1] = None
    out = msm_pkg.final_accumulation(sum((o.encode_point_be32(p) for p in pj), []), W, c)
    assert decode_be32_affine(out) == o.to_affine(o.final_accumulation(pj, c))


def test_no_device_means_loud_failure(msm_pkg):
    """Without a gfx950 device the product must fail loudly, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(msm_pkg.MsmError) as ei:
        msm_pkg.setup_metal_state()
    assert ei.value.status in (msm_pkg.DEVICE_NOT_FOUND, msm_pkg.LIBRARY_ERROR)
    with pytest.raises(msm_pkg.MsmError):
        msm_pkg.get_global_metal_config()


def test_header_compiles_as_plain_c(tmp_path):
    """include/msm_amd.h is the FFI contract: it must be valid C99 on its own (no C++-isms, no HIP or torch types),
    and a C translation unit using every documented call form must compile against it."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    src.write_text(r'''
#include "msm_amd.h"
static void on_sorted(void* user) { *(int*)user = 1; }
int drive(const void* scalars, const void* points, size_t n, unsigned char out[96]) {
  msm_amd_ctx* ctx = 0;
  msm_amd_timings t;
  int fired = 0;
  int rc = msm_amd_init(-1, &ctx);
  if (rc != MSM_AMD_OK) return rc;
  rc = msm_amd_gpu_msm_h2c(ctx, scalars, points, n, out);
  if (rc == MSM_AMD_OK) rc = msm_amd_gpu_msm_h2c_sync(ctx, scalars, points, n, on_sorted, &fired, out);
  if (rc == MSM_AMD_OK) rc = msm_amd_msm(ctx, MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_H2C_AFFINE, scalars, points, n, out);
  if (rc == MSM_AMD_OK) rc = msm_amd_msm_best(ctx, scalars, points, n, out);
  if (rc == MSM_AMD_OK) rc = msm_amd_last_timings(ctx, &t);
  msm_amd_destroy(ctx);
  return rc + (int)msm_amd_cpu_dispatch_below() * 0 + fired * 0;
}
''')
    for std in ("-std=c99", "-std=c11"):
        subprocess.check_call([gcc, std, "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                               str(src), "-o", str(tmp_path / "use_header.o")])
    # and as C++ (the extern "C" guard)
    gxx = shutil.which("g++")
    if gxx:
        subprocess.check_call([gxx, "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c++",
                               "-c", str(src), "-o", str(tmp_path / "use_header_cpp.o")])


def test_cpu_dispatch_threshold_is_small(msm_pkg):
    """msm_best's size dispatch (msm.rs:440-444) re-tuned for MI355X: the reference's 2^17 would send every realistic
    instance to the CPU; here only sizes below the measured crossover (a handful of points) are computed on the host."""
    t = msm_pkg.lib().msm_amd_cpu_dispatch_below()
    assert 0 <= t <= 1024


def test_build_wrote_the_instruction_counts_of_the_shipped_kernel():
    """bench.py's second roofline reads metal-msm-gpu-acceleration_amd/isa_counts.json, which the build derives from the
    compiler's assembly of k_accumulate.hip (tools/isa_counts.py): 8M + 2S with one shared reduction on 9 x 29-bit
    limbs is 6 x 171 + 2 x 135 + 252 + 1 = 1549 multiplier instructions; the affine start 4M + 2S = 865, plus the 171 of
    the mixed addition's first product when the compiler speculates it above the path split."""
    import json
    path = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "isa_counts.json")
    d = json.load(open(path))
    assert 1500 <= d["multiplier_per_mixed_addition"] <= 1600
    assert d["multiplier_per_affine_start"] in range(840, 1060)
    k = d["kernels"]["low_occupancy_2_waves"]
    assert k["mixed_addition"]["valu"] > k["mixed_addition"]["multiplier"] > k["affine_start"]["multiplier"]
    assert "accumulate_kernelILb1E" in k["symbol"]
