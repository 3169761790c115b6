"""The AVX-512 IFMA field arithmetic of the product's CPU MSM (csrc/host_ifma.cpp: eight field elements per vector,
radix 2^52, Montgomery radix 2^260) against Python big integers -- skipped on a host without IFMA, where the CPU MSM
takes its scalar path."""
import ctypes
import random

import pytest

from oracle import bn254_ref as o

P = o.P
Q_INV = pow(1 << 260, -1, P)


def _run(msm_pkg, op, a, b):
    n = len(a)
    ba = b"".join(x.to_bytes(32, "little") for x in a)
    bb = b"".join(x.to_bytes(32, "little") for x in b)
    out = ctypes.create_string_buffer(32 * n)
    st = msm_pkg.lib().msm_amd_test_op_ifma(op, ba, bb, out, n)
    if st == msm_pkg.FUNCTION_ERROR:
        pytest.skip("this host has no AVX-512 IFMA")
    assert st == msm_pkg.OK
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(n)]


def _vals(rng, n):
    edge = [0, 1, 2, P - 1, P - 2, (1 << 52) - 1, 1 << 52, (1 << 104) + 5, (1 << 208) - 1, (1 << 253), P >> 1, (1 << 260) % P]
    return edge + [rng.randrange(P) for _ in range(n)]


def test_montgomery_product_radix_2_260(msm_pkg):
    rng = random.Random(52)
    a = _vals(rng, 1000)
    b = list(reversed(_vals(rng, 1000)))
    assert _run(msm_pkg, 0, a, b) == [x * y * Q_INV % P for x, y in zip(a, b)]
    assert _run(msm_pkg, 0, a, a) == [x * x * Q_INV % P for x in a]                 # squarings
    seven = [7] * len(a)                                                            # a count that is not a multiple of 8
    assert _run(msm_pkg, 0, a[:13], seven[:13]) == [x * 7 * Q_INV % P for x in a[:13]]


def test_exact_subtraction_and_negation(msm_pkg):
    rng = random.Random(53)
    a = _vals(rng, 1000)
    b = list(reversed(_vals(rng, 1000)))
    assert _run(msm_pkg, 1, a, b) == [(x - y) % P for x, y in zip(a, b)]
    assert _run(msm_pkg, 1, a, a) == [0] * len(a)
    nz = [x for x in a if x]
    assert _run(msm_pkg, 2, nz, nz) == [P - x for x in nz]


def test_domain_conversions(msm_pkg):
    """R = 2^256 (the library's Montgomery domain) <-> Q = 2^260 (the vector code's): x * 2^4 and back."""
    rng = random.Random(54)
    a = _vals(rng, 500)
    up = _run(msm_pkg, 3, a, a)
    assert up == [x * 16 % P for x in a]
    assert _run(msm_pkg, 4, up, up) == a
