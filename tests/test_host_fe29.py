"""Host-CPU execution of the single-op bodies (the same source the GPU test kernel runs): checks the
29-bit-limb internal field representation and its group law against the big-int oracle without a GPU."""
import random

import pytest

from oracle import bn254_ref as o
from helpers import be32_fq, decode_be32_affine, fq_be32, rand_jac, rand_point

EDGE = [0, 1, 2, o.P - 1, o.P - 2, o.MONT_R % o.P, (1 << 253) % o.P, (1 << 29) - 1, 1 << 29, (1 << 232) - 1]


def _vals(rng, n):
    return EDGE + [rng.randrange(o.P) for _ in range(n)]


def _run(msm_pkg, op, a, b):
    cnt = len(a)
    flat = msm_pkg.test_op_host(op, sum((fq_be32(x) for x in a), []), sum((fq_be32(x) for x in b), []), cnt)
    return [be32_fq(flat[8 * i:8 * i + 8]) for i in range(cnt)]


def test_host_matches_device_op_table(msm_pkg):
    """32-bit-limb ops through the host entry point (also exercised on the GPU in test_gpu_unit_ops)."""
    rng = random.Random(1)
    a, b = _vals(rng, 100), list(reversed(_vals(rng, 100)))
    assert _run(msm_pkg, msm_pkg.OP_FP_MUL, a, b) == [x * y % o.P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_FP_ADD, a, b) == [(x + y) % o.P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_FP_SUB, a, b) == [(x - y) % o.P for x, y in zip(a, b)]


def test_fp29_roundtrip_mul_sqr(msm_pkg):
    rng = random.Random(2)
    a, b = _vals(rng, 300), list(reversed(_vals(rng, 300)))
    assert _run(msm_pkg, msm_pkg.OP_FP29_ROUNDTRIP, a, b) == a
    assert _run(msm_pkg, msm_pkg.OP_FP29_MUL, a, b) == [x * y % o.P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_FP29_SQR, a, b) == [x * x % o.P for x in a]


@pytest.mark.parametrize("opname,mult", [("OP_FP29_SUB_K4E30", 1), ("OP_FP29_SUB_K8E30", 1), ("OP_FP29_SUB_K8E31", 3),
                                         ("OP_FP29_SUB_K16E30", 1), ("OP_FP29_SUB_K16E31", 3)])
def test_fp29_lifted_subtraction(msm_pkg, opname, mult):
    rng = random.Random(3)
    a, b = _vals(rng, 200), list(reversed(_vals(rng, 200)))
    assert _run(msm_pkg, getattr(msm_pkg, opname), a, b) == [(x - mult * y) % o.P for x, y in zip(a, b)]


def _ec_cases(rng):
    P = [rand_point(rng) for _ in range(24)]
    cases = [(P[i], P[i + 1]) for i in range(0, 16, 2)]
    cases += [(P[0], P[0]), (P[1], P[1])]                 # equal points -> exact slow path (doubling)
    cases += [(P[2], o.aff_neg(P[2]))]                    # opposite points -> identity
    cases += [(P[3], None), (None, P[4]), (None, None)]
    return cases


def test_ec29_add_and_madd(msm_pkg):
    rng = random.Random(4)
    cases = _ec_cases(rng)
    cnt = len(cases)
    a = sum((o.encode_point_be32(rand_jac(rng, p)) for p, _ in cases), [])
    b = sum((o.encode_point_be32(rand_jac(rng, q)) for _, q in cases), [])
    flat = msm_pkg.test_op_host(msm_pkg.OP_EC29_ADD, a, b, cnt)
    assert [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)] == [o.aff_add(p, q) for p, q in cases]
    b_aff = sum((o.encode_point_be32(o.to_jac(q)) for _, q in cases), [])
    flat = msm_pkg.test_op_host(msm_pkg.OP_EC29_MADD, a, b_aff, cnt)
    assert [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)] == [o.aff_add(p, q) for p, q in cases]


def test_ec29_chained_additions_stay_in_bounds(msm_pkg):
    """64 mixed / 16 full additions chained WITHOUT leaving the lazy internal form: exercises the value and
    limb bounds of bn254_ec29.hip.h across iterations (incl. the accumulator starting as the identity and
    the doubling slow path when the accumulator equals the addend)."""
    rng = random.Random(6)
    P = [rand_point(rng) for _ in range(12)]
    cases = [(P[0], P[1]), (None, P[2]), (P[3], P[3]), (P[4], o.aff_neg(P[4])), (P[5], None), (P[6], P[7]),
             (o.scalar_mul(63, P[8]), o.aff_neg(P[8])), (o.scalar_mul(10, P[9]), o.aff_neg(P[9]))]
    cnt = len(cases)
    a = sum((o.encode_point_be32(rand_jac(rng, p)) for p, _ in cases), [])
    b_aff = sum((o.encode_point_be32(o.to_jac(q)) for _, q in cases), [])
    flat = msm_pkg.test_op_host(msm_pkg.OP_EC29_MADD_CHAIN, a, b_aff, cnt)
    exp = [o.aff_add(p, o.scalar_mul(64, q)) if q is not None else p for p, q in cases]
    assert [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)] == exp
    b = sum((o.encode_point_be32(rand_jac(rng, q)) for _, q in cases), [])
    flat = msm_pkg.test_op_host(msm_pkg.OP_EC29_ADD_CHAIN, a, b, cnt)
    exp = [o.aff_add(p, o.scalar_mul(16, q)) if q is not None else p for p, q in cases]
    assert [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)] == exp


def test_ec29_long_accumulation_keeps_bounds(msm_pkg):
    """Chain 300 mixed additions through the internal representation (each result re-enters as the next
    accumulator after an external round trip is NOT taken: the chain is done in one call per step on
    fresh conversions, so instead check a long random walk of additions against the oracle)."""
    rng = random.Random(5)
    pts = [rand_point(rng) for _ in range(40)]
    acc = None
    accj = None
    for step in range(120):
        q = pts[rng.randrange(40)]
        a = o.encode_point_be32(accj)
        b = o.encode_point_be32(o.to_jac(q))
        out = msm_pkg.test_op_host(msm_pkg.OP_EC29_MADD, a, b, 1)
        acc = o.aff_add(acc, q)
        accj = o.decode_point_be32(out)
        assert decode_be32_affine(out) == acc


def _mmadd_cases(rng):
    P = [rand_point(rng) for _ in range(10)]
    cases = [(P[i], P[i + 1]) for i in range(0, 8, 2)]
    cases += [(P[8], P[8])]                          # equal points: the exact slow path doubles
    cases += [(P[9], o.aff_neg(P[9]))]               # opposite points: identity, then the chain restarts from -b
    cases += [(o.scalar_mul(2, P[0]), o.aff_neg(P[0]))]   # the chain passes through a doubling: (-2P + P) + P ...
    return cases


def _mmadd_expect(p, q):
    # op 26 negates both operands, adds them affine + affine, then adds -q three more times (restarting from -q
    # whenever the accumulator is the identity, as the accumulate kernel does)
    np_, nq = o.aff_neg(p), o.aff_neg(q)
    acc = o.aff_add(np_, nq)
    for _ in range(3):
        acc = nq if acc is None else o.aff_add(acc, nq)
    return acc


def test_ec29_affine_plus_affine_start_of_item(msm_pkg):
    """pti_mmadd (4M + 2S, the second point of every work item in accumulate_kernel) on lazily negated operands,
    followed by mixed additions that consume its lazy result."""
    rng = random.Random(26)
    cases = _mmadd_cases(rng)
    a = sum((o.encode_point_be32(o.to_jac(p)) for p, _ in cases), [])
    b = sum((o.encode_point_be32(o.to_jac(q)) for _, q in cases), [])
    flat = msm_pkg.test_op_host(msm_pkg.OP_EC29_MMADD, a, b, len(cases))
    got = [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(len(cases))]
    assert got == [_mmadd_expect(p, q) for p, q in cases]


def test_host64_field_and_group_ops(msm_pkg):
    """The 4 x 64-bit host arithmetic of the CPU tail (csrc/host_fq64.h: window Horner pass + normalisation of every
    MSM) against the big-int oracle AND, coordinate for coordinate, against the portable 8 x 32-bit host code it
    replaced (same formulas, so the projective coordinates must be identical, not only the point)."""
    rng = random.Random(64)
    a, b = _vals(rng, 400), list(reversed(_vals(rng, 400)))
    assert _run(msm_pkg, msm_pkg.OP_H64_FP_MUL, a, b) == [x * y % o.P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_H64_FP_ADD, a, b) == [(x + y) % o.P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_H64_FP_SUB, a, b) == [(x - y) % o.P for x, y in zip(a, b)]
    cases = _ec_cases(rng)
    cnt = len(cases)
    ja = sum((o.encode_point_be32(rand_jac(rng, p)) for p, _ in cases), [])
    jb = sum((o.encode_point_be32(rand_jac(rng, q)) for _, q in cases), [])
    flat = msm_pkg.test_op_host(msm_pkg.OP_H64_EC_ADD, ja, jb, cnt)
    assert [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)] == [o.aff_add(p, q) for p, q in cases]
    assert flat == msm_pkg.test_op_host(msm_pkg.OP_EC_ADD, ja, jb, cnt)
    flat = msm_pkg.test_op_host(msm_pkg.OP_H64_EC_DBL, ja, jb, cnt)
    assert [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)] == [o.aff_add(p, p) for p, _ in cases]
    assert flat == msm_pkg.test_op_host(msm_pkg.OP_EC_DBL, ja, jb, cnt)
    with pytest.raises(msm_pkg.MsmError):
        msm_pkg.test_op_host(99, ja, jb, cnt)


def test_host64_inversion_by_binary_gcd(msm_pkg):
    """h64::inv (binary GCD with 31-step rounds on 64-bit approximations, csrc/host_fq64.h) against Python's modular
    inverse and against the exponentiation it replaced; 0 -> 0."""
    rng = random.Random(37)
    a = [x for x in _vals(rng, 3000) if x] + [1 << k for k in range(0, 254, 7)] + [o.P - (1 << k) for k in range(0, 250, 11)]
    a += [rng.randrange(1, 1 << k) for k in (8, 31, 32, 33, 62, 63, 64, 65, 96, 127, 128, 129, 190, 200)]
    got = _run(msm_pkg, msm_pkg.OP_H64_FP_INV, a, a)
    assert got == [pow(x, -1, o.P) for x in a]
    assert got == _run(msm_pkg, msm_pkg.OP_H64_FP_INV_FERMAT, a, a)
    assert _run(msm_pkg, msm_pkg.OP_H64_FP_INV, [0], [0]) == [0]


def test_unshipped_multiplication_variants_agree(msm_pkg):
    """One Karatsuba level and the lockstep product-scanning chains (build options measured on the GPU and not shipped,
    HISTORY.md) compute the same field elements as the shipped multiplication -- host twins of ops 32..36."""
    if b"+experiments" not in msm_pkg.lib().msm_amd_version():
        pytest.skip("ops 32..36 exist in -DMSM_AMD_EXPERIMENTS builds only (tests/test_experiments.py runs this test there)")
    rng = random.Random(29)
    a = [rng.randrange(o.P) for _ in range(64)] + [0, 1, o.P - 1]
    b = [rng.randrange(o.P) for _ in range(64)] + [o.P - 1, 0, o.P - 1]
    P = o.P
    assert _run(msm_pkg, msm_pkg.OP_FP29_MUL_KARATSUBA, a, b) == [x * y % P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_FP29_LOCKSTEP_PAIR, a, b) == [(x * y + y * y) % P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_FP29_LOCKSTEP_MIX, a, b) == [(2 * x * y + 2 * x * x + y * y) % P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_FP29_LOCKSTEP_TRIPLE, a, b) == [(x * y + x * x + y * y) % P for x, y in zip(a, b)]
    assert _run(msm_pkg, msm_pkg.OP_FP29_MUL2_KARATSUBA, a, b) == [2 * x * y % P for x, y in zip(a, b)]
