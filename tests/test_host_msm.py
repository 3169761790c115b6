"""The product's CPU MSM (csrc/host_msm.hip: where the reference calls halo2curves::msm::msm_best,
src/bin/gpu_profiler.rs:157-159, src/metal/msm.rs:412) against the oracle -- host code only, no GPU."""
import random
import subprocess
import os
import json

import pytest

from oracle import bn254_ref as o
from oracle import c_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    return o.decode_jacobian_mont_le(a) == o.decode_jacobian_mont_le(b)


@pytest.mark.parametrize("log_n,threads", [(0, 1), (1, 2), (3, 1), (5, 4), (8, 3), (10, 1), (12, 8), (14, 5)])
def test_host_msm_matches_oracle(msm_pkg, log_n, threads):
    n = 1 << log_n
    pts, sc = co.gen_instance(o.SEED_BASE + 900 + log_n, n)
    got = msm_pkg.host_msm(sc, pts, n, threads)
    assert _same(got, co.msm_best(sc, pts, n))
    assert got[64:] in (o.int_to_le_bytes32(o.fq_to_mont(1)), bytes(32))   # normalised: z = R mod p, or 0


def test_host_msm_is_thread_count_independent(msm_pkg):
    n = 3000   # not a power of two: ragged point groups
    pts, sc = co.gen_instance(o.SEED_BASE + 77, n)
    want = msm_pkg.host_msm(sc, pts, n, 1)
    for t in (2, 3, 7, 16, 0):
        assert msm_pkg.host_msm(sc, pts, n, t) == want, t


def test_host_msm_edge_cases(msm_pkg):
    """Zero scalars, identity points, all scalars equal (every point of a window in ONE bucket: the Jacobian side
    accumulators), all points equal (batched-affine doublings), P and -P in one bucket (cancellation)."""
    n = 2048
    rng = random.Random(5)
    pts, sc = co.gen_instance(o.SEED_BASE + 5, n)
    assert msm_pkg.host_msm(bytes(32 * n), pts, n, 4)[64:] == bytes(32)              # all-zero scalars: identity
    assert msm_pkg.host_msm(sc, bytes(64 * n), n, 4)[64:] == bytes(32)               # all-identity points
    sc_eq = sc[:32] * n
    assert _same(msm_pkg.host_msm(sc_eq, pts, n, 4), co.msm_best(sc_eq, pts, n))
    pts_eq = pts[:64] * n
    assert _same(msm_pkg.host_msm(sc, pts_eq, n, 4), co.msm_best(sc, pts_eq, n))
    assert _same(msm_pkg.host_msm(sc_eq, pts_eq, n, 4), co.msm_best(sc_eq, pts_eq, n))
    pb, sb = bytearray(pts), bytearray(sc)
    for i in rng.sample(range(n), 40):
        pb[64 * i:64 * i + 64] = bytes(64)
    for i in rng.sample(range(n), 40):
        sb[32 * i:32 * i + 32] = bytes(32)
    assert _same(msm_pkg.host_msm(bytes(sb), bytes(pb), n, 3), co.msm_best(bytes(sb), bytes(pb), n))
    # P, -P with the same scalar cancel; what is left is k_last * P_last
    P = tuple(o.fq_from_mont(int.from_bytes(pts[i:i + 32], "little")) for i in (0, 32))
    neg = o.encode_affine_h2c((P[0], (o.P - P[1]) % o.P))
    three = pts[:64] + neg + pts[64:128]
    k = sc[:32] + sc[:32] + sc[32:64]
    assert _same(msm_pkg.host_msm(k, three, 3, 1), co.msm_naive(k[64:], three[128:], 1))
    many = (pts[:64] + neg) * 512      # a full batch of cancellations and re-fills in one bucket set
    assert msm_pkg.host_msm(sc[:32] * 1024, many, 1024, 2)[64:] == bytes(32)


def test_host_msm_canonical_scalars_above_r(msm_pkg):
    """CANON_LE scalars are reduced mod r like the device path does (k = r + 5, k = 2^256 - 1)."""
    n = 64
    pts, _ = co.gen_instance(o.SEED_BASE + 6, n)
    ks = [o.R_ORDER + 5, (1 << 256) - 1] + [random.Random(i).randrange(1 << 256) for i in range(n - 2)]
    raw = b"".join(k.to_bytes(32, "little") for k in ks)
    red = b"".join((k % o.R_ORDER).to_bytes(32, "little") for k in ks)
    got = msm_pkg.host_msm(raw, pts, n, 2, scalar_layout=msm_pkg.SCALAR_CANON_LE)
    want = msm_pkg.host_msm(red, pts, n, 2, scalar_layout=msm_pkg.SCALAR_CANON_LE)
    assert got == want
    mont = b"".join(o.encode_scalar_h2c(k % o.R_ORDER) for k in ks)
    assert _same(got, co.msm_best(mont, pts, n))


def test_host_generator_equals_oracle_generator(msm_pkg):
    """msm_amd_generate_instance_host = the device generator's code on the host = the oracle's generator: same bytes."""
    for n, seed in ((1, 3), (1000, o.SEED_BASE), (4096, o.SEED_BASE + 4)):
        assert msm_pkg.generate_instance_host(seed, n, True, 3) == co.gen_instance(seed, n)
    p1, s1 = msm_pkg.generate_instance_host(9, 500, False, 1)
    p2, s2 = co.gen_instance(9, 500, scalars_mont=False)
    assert (p1, s1) == (p2, s2)


def test_host_msm_rejects_bad_arguments(msm_pkg):
    L = msm_pkg.lib()
    import ctypes
    out = ctypes.create_string_buffer(96)
    assert L.msm_amd_host_msm(0, 0, None, b"x" * 64, 1, 1, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_host_msm(0, 0, b"x" * 32, b"x" * 64, 0, 1, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_host_msm(0, msm_pkg.POINT_ARK_PROJECTIVE, b"x" * 32, b"x" * 96, 1, 1, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_host_msm(msm_pkg.SCALAR_CANON_BE32, 0, b"x" * 32, b"x" * 64, 1, 1, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_host_threads() >= 1


def test_gpu_profiler_cpu_mode_needs_no_gpu():
    """BASELINE config 1 shape (`gpu_profiler 16 1 cpu 5`, here 2^10): runs in this GPU-less container, prints the
    reference's two report lines, and its result equals the oracle's."""
    exe = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "gpu_profiler")
    r = subprocess.run([exe, "10", "2", "cpu", "2", "--json", "--threads", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Total Execution Time" in r.stderr and "Average Instance Execution Time" in r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["mode"] == "cpu" and d["host_threads"] == 4
    pts, sc = co.gen_instance(o.SEED_BASE, 1 << 10)
    want = co.msm_best(sc, pts, 1 << 10)
    x, _y = o.decode_jacobian_mont_le(want)
    # result0_x_le_hex is the Montgomery x of the normalised result
    assert bytes.fromhex(d["result0_x_le_hex"]) == o.int_to_le_bytes32(o.fq_to_mont(x))


def test_host_msm_skew_at_a_size_that_uses_the_batched_paths(msm_pkg):
    """2^15 points (batched-affine additions, on AVX-512 IFMA where the host has it): all scalars equal -- every point of
    a window wants ONE bucket, the waiting list does not drain and the Jacobian accumulators take over; all points
    equal -- every addition in a bucket is a doubling; half the points the negatives of the other half."""
    n = 1 << 15
    pts, sc = co.gen_instance(o.SEED_BASE + 1515, n)
    sc_eq = sc[:32] * n
    assert _same(msm_pkg.host_msm(sc_eq, pts, n, 4), co.msm_best(sc_eq, pts, n))
    pts_eq = pts[:64] * n
    assert _same(msm_pkg.host_msm(sc, pts_eq, n, 4), co.msm_best(sc, pts_eq, n))
    half = n // 2
    neg = bytearray(pts[:64 * half])
    for i in range(half):
        y = int.from_bytes(neg[64 * i + 32:64 * i + 64], "little")
        neg[64 * i + 32:64 * i + 64] = ((o.P - y) % o.P).to_bytes(32, "little")
    both = pts[:64 * half] + bytes(neg)
    sc2 = sc[:32 * half] * 2                      # k_i P_i + k_i (-P_i) = O for every i
    assert msm_pkg.host_msm(sc2, both, n, 4)[64:] == bytes(32)
    sc3 = sc[:32 * half] + sc[32 * half:]         # different scalars on the negatives: a generic answer
    assert _same(msm_pkg.host_msm(sc3, both, n, 3), co.msm_best(sc3, both, n))


def test_host_msm_scalar_and_vector_paths_agree(msm_pkg):
    """MSM_AMD_HOST_NO_IFMA=1 forces the MULX / ADX path on a host that has AVX-512 IFMA: same bytes either way."""
    import os
    n = 1 << 14
    pts, sc = co.gen_instance(o.SEED_BASE + 1414, n)
    a = msm_pkg.host_msm(sc, pts, n, 3)
    os.environ["MSM_AMD_HOST_NO_IFMA"] = "1"
    try:
        b = msm_pkg.host_msm(sc, pts, n, 3)
    finally:
        del os.environ["MSM_AMD_HOST_NO_IFMA"]
    assert a == b and _same(a, co.msm_best(sc, pts, n))
