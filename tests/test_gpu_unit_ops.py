"""Single-op kernels vs the big-int oracle.  Mirrors the reference's L0 tests
(src/metal/tests/test_bn254.rs:101-160 uint ops, :239-297 field ops, :373-458 curve ops): same operand
shapes (128-bit uint operands, 32-bit multiplier, shifts 0..255, random field elements, P+P, P+inf ...)."""
import random

import pytest

from oracle import bn254_ref as o
from helpers import be32_fq, decode_be32_affine, fq_be32, rand_jac, rand_point

pytestmark = pytest.mark.gpu
M256 = (1 << 256) - 1


def _flat(vals):
    out = []
    for v in vals:
        out += o.int_to_be32_limbs(v)
    return out


def _ints(flat):
    return [o.be32_limbs_to_int(flat[8 * i:8 * i + 8]) for i in range(len(flat) // 8)]


def test_uint_ops(cfg, msm_pkg):
    rng = random.Random(11)
    n = 256
    a = [rng.getrandbits(128) for _ in range(n)] + [M256, 0, 1 << 255]
    b = [rng.getrandbits(128) for _ in range(n)] + [1, 0, 1 << 255]
    cnt = len(a)
    got = _ints(cfg.test_op(msm_pkg.OP_UINT_ADD, _flat(a), _flat(b), cnt))
    assert got == [(x + y) & M256 for x, y in zip(a, b)]
    hi = [max(x, y) for x, y in zip(a, b)]
    lo = [min(x, y) for x, y in zip(a, b)]
    got = _ints(cfg.test_op(msm_pkg.OP_UINT_SUB, _flat(hi), _flat(lo), cnt))
    assert got == [x - y for x, y in zip(hi, lo)]
    got = _ints(cfg.test_op(msm_pkg.OP_UINT_SUB, _flat(lo), _flat(hi), cnt))   # wrap-around
    assert got == [(x - y) & M256 for x, y in zip(lo, hi)]
    b32 = [rng.getrandbits(32) for _ in range(cnt)]
    got = _ints(cfg.test_op(msm_pkg.OP_UINT_PROD, _flat(a), _flat(b32), cnt))
    assert got == [(x * y) & M256 for x, y in zip(a, b32)]
    sh = [rng.randrange(256) for _ in range(cnt - 4)] + [0, 32, 255, 64]
    wide = [rng.getrandbits(256) for _ in range(cnt)]
    got = _ints(cfg.test_op(msm_pkg.OP_UINT_SHL, _flat(wide), _flat(sh), cnt))
    assert got == [(x << s) & M256 for x, s in zip(wide, sh)]
    got = _ints(cfg.test_op(msm_pkg.OP_UINT_SHR, _flat(wide), _flat(sh), cnt))
    assert got == [x >> s for x, s in zip(wide, sh)]


def test_fp_ops(cfg, msm_pkg):
    rng = random.Random(12)
    edge = [0, 1, o.P - 1, o.MONT_R % o.P, 2, o.P - 2]
    a = [rng.randrange(o.P) for _ in range(500)] + edge + edge
    b = [rng.randrange(o.P) for _ in range(500)] + edge + list(reversed(edge))
    cnt = len(a)
    fa = sum((fq_be32(x) for x in a), [])
    fb = sum((fq_be32(x) for x in b), [])

    def run(op, fb_=fb):
        flat = cfg.test_op(op, fa, fb_, cnt)
        return [be32_fq(flat[8 * i:8 * i + 8]) for i in range(cnt)]

    assert run(msm_pkg.OP_FP_ADD) == [(x + y) % o.P for x, y in zip(a, b)]
    assert run(msm_pkg.OP_FP_SUB) == [(x - y) % o.P for x, y in zip(a, b)]
    assert run(msm_pkg.OP_FP_MUL) == [(x * y) % o.P for x, y in zip(a, b)]
    assert run(msm_pkg.OP_FP_NEG) == [(-x) % o.P for x in a]
    e = [rng.getrandbits(32) for _ in range(cnt - 3)] + [0, 1, 0xFFFFFFFF]
    assert run(msm_pkg.OP_FP_POW, _flat(e)) == [pow(x, k, o.P) for x, k in zip(a, e)]
    # raw residue check: the product of Montgomery residues is the Montgomery residue (bit-exact limbs)
    flat = cfg.test_op(msm_pkg.OP_FP_MUL, fa, fb, cnt)
    assert _ints(flat) == [o.mont_mul_p(o.fq_to_mont(x), o.fq_to_mont(y)) for x, y in zip(a, b)]


def test_ec_ops(cfg, msm_pkg):
    rng = random.Random(13)
    P = [rand_point(rng) for _ in range(24)]
    cases = []   # (p, q) affine or None
    for i in range(0, 16, 2):
        cases.append((P[i], P[i + 1]))                    # generic
    cases += [(P[0], P[0]), (P[1], P[1])]                 # add_with_self -> doubling (test_bn254.rs add_with_self)
    cases += [(P[2], o.aff_neg(P[2]))]                    # P + (-P) = identity
    cases += [(P[3], None), (None, P[4]), (None, None)]   # infinity rhs / lhs / both (test_bn254.rs:430-458)
    cnt = len(cases)
    a = sum((o.encode_point_be32(rand_jac(rng, p)) for p, _ in cases), [])
    b = sum((o.encode_point_be32(rand_jac(rng, q)) for _, q in cases), [])
    flat = cfg.test_op(msm_pkg.OP_EC_ADD, a, b, cnt)
    got = [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)]
    assert got == [o.aff_add(p, q) for p, q in cases]
    # mixed add: rhs affine (z = one) or identity
    b_aff = sum((o.encode_point_be32(o.to_jac(q)) for _, q in cases), [])
    flat = cfg.test_op(msm_pkg.OP_EC_MADD, a, b_aff, cnt)
    got = [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)]
    assert got == [o.aff_add(p, q) for p, q in cases]
    flat = cfg.test_op(msm_pkg.OP_EC_DBL, a, a, cnt)
    got = [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)]
    assert got == [o.aff_add(p, p) for p, _ in cases]
    # scalar multiplication incl. 0, 1, r-1, r (test_bn254.rs `mul`)
    ks = [rng.randrange(o.R_ORDER) for _ in range(cnt - 4)] + [0, 1, o.R_ORDER - 1, o.R_ORDER]
    kb = sum((o.int_to_be32_limbs(k) for k in ks), [])
    flat = cfg.test_op(msm_pkg.OP_EC_MUL, a, kb, cnt)
    got = [decode_be32_affine(flat[24 * i:24 * i + 24]) for i in range(cnt)]
    assert got == [o.scalar_mul(k, p) if p is not None else None for k, (p, _) in zip(ks, cases)]


def test_ec29_device_matches_host_twin_and_oracle(cfg, msm_pkg):
    """The internal-representation point ops (incl. the affine + affine start of a work item, op 26) on the GPU:
    same answers as the host build of the same code and as the oracle."""
    rng = random.Random(26)
    P = [rand_point(rng) for _ in range(12)]
    cases = [(P[i], P[i + 1]) for i in range(0, 8, 2)] + [(P[8], P[8]), (P[9], o.aff_neg(P[9])),
                                                          (o.scalar_mul(2, P[0]), o.aff_neg(P[0]))]
    cnt = len(cases)
    a = sum((o.encode_point_be32(o.to_jac(p)) for p, _ in cases), [])
    b = sum((o.encode_point_be32(o.to_jac(q)) for _, q in cases), [])
    for op in (msm_pkg.OP_EC29_MMADD, msm_pkg.OP_EC29_MADD, msm_pkg.OP_EC29_ADD, msm_pkg.OP_EC29_MADD_CHAIN):
        dev = cfg.test_op(op, a, b, cnt)
        assert list(dev) == list(msm_pkg.test_op_host(op, a, b, cnt)), op

    def expect(p, q):
        np_, nq = o.aff_neg(p), o.aff_neg(q)
        acc = o.aff_add(np_, nq)
        for _ in range(3):
            acc = nq if acc is None else o.aff_add(acc, nq)
        return acc

    dev = cfg.test_op(msm_pkg.OP_EC29_MMADD, a, b, cnt)
    assert [decode_be32_affine(dev[24 * i:24 * i + 24]) for i in range(cnt)] == [expect(p, q) for p, q in cases]


def test_unshipped_multiplication_variants_on_the_device(cfg, msm_pkg):
    """Ops 32..36: one Karatsuba level and the lockstep product-scanning chains (inline-assembly multiply-adds on the
    device), build options that were measured and not shipped (HISTORY.md) -- same values as the shipped
    multiplication, and the host-only ops 27..31 are refused by the device entry point."""
    if b"+experiments" not in msm_pkg.lib().msm_amd_version():
        pytest.skip("ops 32..36 exist in -DMSM_AMD_EXPERIMENTS builds only (tests/test_experiments.py runs this test there)")
    rng = random.Random(2936)
    a = [rng.randrange(o.P) for _ in range(300)] + [0, 1, o.P - 1]
    b = [rng.randrange(o.P) for _ in range(300)] + [o.P - 1, 0, o.P - 1]
    cnt = len(a)
    fa, fb = sum((fq_be32(x) for x in a), []), sum((fq_be32(x) for x in b), [])

    def run(op):
        flat = cfg.test_op(op, fa, fb, cnt)
        return [be32_fq(flat[8 * i:8 * i + 8]) for i in range(cnt)]

    P = o.P
    assert run(msm_pkg.OP_FP29_MUL_KARATSUBA) == [x * y % P for x, y in zip(a, b)]
    assert run(msm_pkg.OP_FP29_LOCKSTEP_PAIR) == [(x * y + y * y) % P for x, y in zip(a, b)]
    assert run(msm_pkg.OP_FP29_LOCKSTEP_MIX) == [(2 * x * y + 2 * x * x + y * y) % P for x, y in zip(a, b)]
    assert run(msm_pkg.OP_FP29_LOCKSTEP_TRIPLE) == [(x * y + x * x + y * y) % P for x, y in zip(a, b)]
    assert run(msm_pkg.OP_FP29_MUL2_KARATSUBA) == [2 * x * y % P for x, y in zip(a, b)]
    with pytest.raises(msm_pkg.MsmError):
        cfg.test_op(msm_pkg.OP_H64_FP_MUL, fa, fb, cnt)
