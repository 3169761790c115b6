"""CPU tests of the oracle itself (no GPU): known answers, Python big-int vs C restatement, the
deterministic generator, and the committed golden fixtures."""
import json
import os
import random
import pytest

from oracle import bn254_ref as o
from oracle import c_oracle as co
from helpers import h2c_instance_bytes, rand_jac, rand_point, small_instance

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_known_answers():
    assert o.is_on_curve(o.GEN)
    # 2G on alt_bn128 (the EIP-196 test vector), independent of any code in this repository
    assert o.scalar_mul(2, o.GEN) == (
        1368015179489954701390400359078579693043519447331113978918064868415326638035,
        9918110051302171585080402603319702774565515993150576347155970296011118125764)
    # 3G = G + 2G (a generic addition, not a doubling): the alt_bn128 ecAdd/ecMul test vector
    g3 = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
          19321533766552368860946552437480515441416830039777911637913418824951667761761)
    assert o.aff_add(o.GEN, o.scalar_mul(2, o.GEN)) == g3 and o.scalar_mul(3, o.GEN) == g3
    assert o.to_affine(o.scalar_mul_jac(o.R_ORDER, o.to_jac(o.GEN))) is None          # r * G = O
    assert o.aff_add(o.GEN, o.aff_neg(o.GEN)) is None
    # constants hard-coded in the reference shader (fp_bn254.h.metal:25-46), big-endian limb order
    assert o.int_to_be32_limbs(o.P) == o.REF_N_LIMBS_BE
    assert (o.MONT_R * o.MONT_R) % o.P == 0x06D89F71CAB8351F47AB1EFF0A417FF6B5E71911D44501FBF32CFC5B538AFA89
    assert o.MONT_R - o.P == (1 << 256) - o.P


def test_c_field_and_group_ops_match_python():
    rng = random.Random(3)
    for _ in range(200):
        a, b = rng.randrange(o.P), rng.randrange(o.P)
        am, bm = o.fq_to_mont(a), o.fq_to_mont(b)
        ab, bb = o.int_to_le_bytes32(am), o.int_to_le_bytes32(bm)
        assert int.from_bytes(co.fq_mul(ab, bb), "little") == o.mont_mul_p(am, bm)
        assert int.from_bytes(co.fq_add(ab, bb), "little") == (am + bm) % o.P
        assert int.from_bytes(co.fq_sub(ab, bb), "little") == (am - bm) % o.P
    for _ in range(50):
        k = rng.randrange(o.R_ORDER)
        assert int.from_bytes(co.fr_from_mont(o.encode_scalar_h2c(k)), "little") == k
    pts = [rand_point(rng) for _ in range(12)]
    cases = [(pts[0], pts[1]), (pts[2], pts[2]), (pts[3], o.aff_neg(pts[3])), (pts[4], None), (None, pts[5]),
             (None, None), (pts[6], pts[7])]
    for p, q in cases:
        a = o.encode_projective_ark(rand_jac(rng, p))
        b = o.encode_projective_ark(rand_jac(rng, q))
        assert o.decode_jacobian_mont_le(co.jac_add(a, b)) == o.aff_add(p, q)
        assert o.decode_jacobian_mont_le(co.jac_double(a)) == o.aff_add(p, p)


def test_generator_c_equals_python():
    n = 40
    pb, sb = co.gen_instance(o.SEED_BASE + 1, n, True, threads=3)
    pts, sc = o.gen_instance(o.SEED_BASE + 1, n)
    assert all(o.is_on_curve(p) for p in pts)
    assert pb == b"".join(o.encode_affine_h2c(p) for p in pts)
    assert sb == b"".join(o.encode_scalar_h2c(k) for k in sc)
    pb2, sb2 = co.gen_instance(o.SEED_BASE + 1, n, False, threads=1)
    assert pb2 == pb and sb2 == b"".join(o.int_to_le_bytes32(k) for k in sc)


def test_msm_variants_agree_small():
    for n in (1, 2, 5, 31, 32, 100):
        pts, sc = small_instance(n, n)
        if n >= 5:
            sc[1] = 0
            pts[2] = None
            pts[3] = pts[4]
            sc[3] = sc[4]
        sb, pb = h2c_instance_bytes(pts, sc)
        expect = o.msm_naive(sc, pts)
        assert o.decode_jacobian_mont_le(co.msm_naive(sb, pb, n)) == expect
        assert o.decode_jacobian_mont_le(co.msm_reference_pipeline(sb, pb, n, 0)) == expect
        assert o.decode_jacobian_mont_le(co.msm_reference_pipeline(sb, pb, n, 6)) == expect
        for th in (1, 3):
            assert o.decode_jacobian_mont_le(co.msm_best(sb, pb, n, th)) == expect
        assert o.msm_pippenger(sc, pts) == expect


def test_msm_medium_c_variants_agree():
    n = 1 << 12
    pb, sb = co.gen_instance(77, n)
    a = co.msm_best(sb, pb, n, 4)
    b = co.msm_reference_pipeline(sb, pb, n, 0)
    c = co.msm_best(sb, pb, n, 1)
    assert a == b == c


def test_dlog_identity():
    """MSM(k, (a0 + i d) G) == (sum k_i (a0 + i d)) G -- validates an MSM without any MSM code."""
    n = 300
    rng = random.Random(8)
    a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
    sc = [rng.randrange(o.R_ORDER) for _ in range(n)]
    sb = b"".join(o.encode_scalar_h2c(k) for k in sc)
    pb, exp = co.dlog_instance(a0, d, sb, n, threads=3)
    e = sum(k * (a0 + i * d) for i, k in enumerate(sc)) % o.R_ORDER
    assert o.decode_jacobian_mont_le(exp) == o.scalar_mul(e, o.GEN)
    # the generated points really are (a0 + i d) G
    for i in (0, 1, 2, 255, 256, 257, n - 1):
        x = o.fq_from_mont(int.from_bytes(pb[64 * i:64 * i + 32], "little"))
        y = o.fq_from_mont(int.from_bytes(pb[64 * i + 32:64 * i + 64], "little"))
        assert (x, y) == o.scalar_mul((a0 + i * d) % o.R_ORDER, o.GEN)
    assert co.msm_best(sb, pb, n, 2) == exp


def test_stage_mirrors_consistent():
    """The per-stage mirrors compose to the whole MSM (msm.rs:189-217 order)."""
    pts, sc = small_instance(5, 40)
    c, _starts, W, bl = o.window_params(40)
    pairs = o.prepare_buckets_indices(sc, c, W)
    assert len(pairs) == 40 * W
    # the reference's "breaking scalar" 2^14 + 1 with window 14 (prepare_buckets_indices.rs:132-137)
    pr = o.prepare_buckets_indices([(1 << 14) + 1], 14, 19)
    assert pr[0] == (0, 0) and pr[1] == (1 * ((1 << 14) - 1) + 1 - 1, 0) and pr[2] == (0xFFFFFFFF, 0xFFFFFFFF)
    srt = o.sort_buckets_indices(pairs)
    assert sorted(srt) == sorted(pairs) and all(a[0] <= b[0] for a, b in zip(srt, srt[1:]))
    buckets = o.bucket_wise_accumulation(srt, [o.to_jac(p) for p in pts], W * bl)
    res = o.sum_reduction(W, buckets)
    assert o.to_affine(o.final_accumulation(res, c)) == o.msm_naive(sc, pts)


def test_golden_fixtures_match_oracle():
    """tests/golden/*.json were produced by tests/golden/make_golden.py from this oracle; re-derive."""
    with open(os.path.join(GOLDEN, "msm_small.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        pts, sc = o.gen_instance(case["seed"], case["n"])
        exp = o.msm_naive(sc, pts)
        assert [hex(exp[0]), hex(exp[1])] == case["result_affine"]
    with open(os.path.join(GOLDEN, "field_ops.json")) as f:
        g = json.load(f)
    for v in g["mul"]:
        a, b = int(v["a"], 16), int(v["b"], 16)
        assert hex(a * b % o.P) == v["ab"] and hex(o.mont_mul_p(o.fq_to_mont(a), o.fq_to_mont(b))) == v["mont_ab"]


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 32, 33, 200, 1500])
def test_window_parallel_batched_affine_restatement(n):
    """oracle_msm_best (halo2curves 0.7 msm_best shape: window tasks, Booth digits, affine buckets with batched
    additions, Jacobian side buckets for collisions) against the definition (double-and-add) and against the second,
    independent C implementation (oracle_msm_chunked), for every task layout: the halo2curves shape (groups = 1),
    automatic groups, more groups than useful, one thread."""
    from oracle import c_oracle as co
    pts, sc = co.gen_instance(o.SEED_BASE + 3000 + n, n)
    want = o.decode_jacobian_mont_le(co.msm_naive(sc, pts, n)) if n <= 200 else None
    ref = o.decode_jacobian_mont_le(co.msm_chunked(sc, pts, n, 3))
    if want is not None:
        assert ref == want
    for threads, groups in ((1, 1), (4, 1), (4, 0), (8, 5)):
        got, info = co.msm_best_ex(sc, pts, n, threads, groups)
        assert o.decode_jacobian_mont_le(got) == ref, (threads, groups, info)
        assert info["threads_used"] >= 1 and info["windows"] >= 1


def test_batched_affine_corner_cases():
    """Equal points in one bucket (doubling inside a batch), P and -P (bucket becomes empty), identity bases, zero
    scalars, and many hits on one bucket within a batch of 64 (the Jacobian side bucket)."""
    from oracle import c_oracle as co
    rng = random.Random(77)
    n = 300
    P = [o.scalar_mul(rng.randrange(1, o.R_ORDER), o.GEN) for _ in range(6)]
    pts = [P[i % 6] for i in range(n)]
    sc = [rng.randrange(o.R_ORDER) for _ in range(n)]
    for i in range(0, 60):
        sc[i] = 12345                      # the same digit pattern: one bucket per window gets 60 hits in a row
    pts[70] = o.aff_neg(pts[64])
    sc[70] = sc[64]                        # P + (-P) in the same bucket
    pts[80] = None                         # identity base
    sc[90] = 0
    sb = b"".join(o.encode_scalar_h2c(k) for k in sc)
    pb = b"".join(o.encode_affine_h2c(p) for p in pts)
    want = o.msm_naive(sc, pts)
    for threads, groups in ((1, 1), (4, 0), (3, 2)):
        got, _ = co.msm_best_ex(sb, pb, n, threads, groups)
        assert o.decode_jacobian_mont_le(got) == want
    assert o.decode_jacobian_mont_le(co.msm_chunked(sb, pb, n, 2)) == want
