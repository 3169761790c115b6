"""Per-stage parity through the C ABI, mirroring the reference's stage tests:
prepare_buckets_indices.rs:121-219, sort_buckets.rs:91-182, bucket_wise_accumulation.rs:227-601,
sum_reduction.rs:261-356, final_accumulation.rs.  Oracles are the CPU mirrors in oracle/bn254_ref.py."""
import json
import os
import random

import pytest

from oracle import bn254_ref as o
from helpers import decode_be32_affine, rand_jac, rand_point

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _scalars_be32(sc):
    return sum((o.encode_scalar_be32(k) for k in sc), [])


def test_prepare_buckets_indices_breaking_scalar(cfg):
    """prepare_buckets_indices.rs:121-170: window 14, scalar 2^14 + 1 first."""
    rng = random.Random(1)
    sc = [(1 << 14) + 1] + [rng.randrange(o.R_ORDER) for _ in range(4)]
    c, W = 14, len(range(0, 254, 14))
    got = cfg.prepare_buckets_indices(_scalars_be32(sc), len(sc), c, W)
    assert got == o.prepare_buckets_indices(sc, c, W)          # exact order, stronger than the set compare


def test_prepare_buckets_indices_golden(cfg):
    with open(os.path.join(GOLDEN, "digits.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        sc = [int(s, 16) for s in case["scalars"]]
        got = cfg.prepare_buckets_indices(_scalars_be32(sc), len(sc), case["window_size"], case["num_windows"])
        assert got == [tuple(p) for p in case["pairs"]]


@pytest.mark.parametrize("window_size,log_n", [(2, 3), (5, 6), (13, 9), (15, 12), (16, 10), (24, 8), (21, 11)])
def test_prepare_buckets_indices_large(cfg, window_size, log_n):
    """prepare_buckets_indices.rs:174-219 (proptest: window 2..24, n 2^3..2^15)."""
    rng = random.Random(window_size * 100 + log_n)
    n = 1 << log_n
    sc = [rng.randrange(o.R_ORDER) for _ in range(n)]
    W = len(range(0, 254, window_size))
    got = cfg.prepare_buckets_indices(_scalars_be32(sc), n, window_size, W)
    assert got == o.prepare_buckets_indices(sc, window_size, W)


def _check_sorted(pairs, got):
    assert sorted(got) == sorted(pairs)                       # nothing lost (multiset)
    assert all(a[0] <= b[0] for a, b in zip(got, got[1:]))    # non-decreasing keys


def test_sort_buckets_indices_small(cfg):
    """sort_buckets.rs:91-126 with the reference's literal pair list."""
    with open(os.path.join(GOLDEN, "reference_index_lists.json")) as f:
        flat = json.load(f)["sort_buckets_small"]
    pairs = [(flat[2 * i], flat[2 * i + 1]) for i in range(len(flat) // 2)]
    got = cfg.sort_buckets_indices(pairs)
    _check_sorted(pairs, got)
    assert got == sorted(pairs, key=lambda p: p[0])           # and stable, like par_sort_by_key


@pytest.mark.parametrize("length", [32 + 1, 256 * 7 + 5, 4096, 4097, 3 * 4096 + 17, 50000])
def test_sort_buckets_indices_large(cfg, length):
    """sort_buckets.rs:130-182: keys < 16*length, values < length; plus sentinels."""
    rng = random.Random(length)
    pairs = [(rng.randrange(length * 16), rng.randrange(length)) for _ in range(length)]
    for i in range(0, length, 97):
        pairs[i] = (0xFFFFFFFF, 0xFFFFFFFF)
    got = cfg.sort_buckets_indices(pairs)
    _check_sorted(pairs, got)
    assert got == sorted(pairs, key=lambda p: p[0])


def _run_accumulation(cfg, pairs, points_aff, rng):
    pairs = sorted(pairs, key=lambda p: p[0])
    pj = [rand_jac(rng, p) for p in points_aff]
    total = max(a for a, _ in pairs) + 1
    got = cfg.bucket_wise_accumulation(pairs, sum((o.encode_point_be32(p) for p in pj), []), len(pj), total)
    exp = o.bucket_wise_accumulation(pairs, pj, total)
    assert [decode_be32_affine(g) for g in got] == [o.to_affine(e) for e in exp]
    # untouched buckets stay all-zero (z = 0), as in Metal's zero-filled buffers
    for b in range(total):
        if exp[b] is None and all(k != b for k, _ in pairs):
            assert got[b] == [0] * 24


def test_bucket_wise_accumulation_reference_lists(cfg):
    """bucket_wise_accumulation.rs:227-546: the reference's 16 index lists (gaps, duplicates -> P+P,
    single bucket, boundary-spanning buckets, three 'Failing Instance' regressions)."""
    with open(os.path.join(GOLDEN, "reference_index_lists.json")) as f:
        cases = json.load(f)["bucket_wise_accumulation"]
    assert len(cases) == 16
    rng = random.Random(99)
    for case in cases:
        pairs = [tuple(p) for p in case["buckets_indices"]]
        npts = max(i for _, i in pairs) + 1
        pts = [rand_point(rng) for _ in range(npts)]
        _run_accumulation(cfg, pairs, pts, rng)


@pytest.mark.parametrize("log_size,num_buckets", [(3, 2), (5, 7), (7, 31), (9, 16), (9, 2)])
def test_bucket_wise_accumulation_large_instance(cfg, log_size, num_buckets):
    """bucket_wise_accumulation.rs:551-601 (proptest: n 2^3..2^9, 2..31 buckets)."""
    rng = random.Random(log_size * 50 + num_buckets)
    n = 1 << log_size
    pts = [rand_point(rng) for _ in range(min(n, 64))]
    pts = [pts[i % len(pts)] for i in range(n)]              # repeats exercise the doubling path too
    pairs = [(rng.randrange(num_buckets), i) for i in range(n)]
    _run_accumulation(cfg, pairs, pts, rng)


def test_bucket_wise_accumulation_sentinels_skipped(cfg):
    rng = random.Random(5)
    pts = [rand_point(rng) for _ in range(6)]
    pairs = [(0, 0), (0, 1), (3, 2), (0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF)]
    pj = [rand_jac(rng, p) for p in pts]
    got = cfg.bucket_wise_accumulation(pairs, sum((o.encode_point_be32(p) for p in pj), []), len(pj), 5)
    exp = o.bucket_wise_accumulation(pairs, pj, 5)
    assert [decode_be32_affine(g) for g in got] == [o.to_affine(e) for e in exp]


@pytest.mark.parametrize("window_num,buckets_size", [(1, 3), (1, 7), (3, 8), (2, 2), (19, 3 * 16), (5, 7 * 64),
                                                       (2, 3584), (1, 32767)])
def test_sum_reduction(cfg, window_num, buckets_size):
    """sum_reduction.rs:261-356: res[j] = sum_b (b+1) * B[j*len + b] for W 1..19, bucket counts 2..3584
    (and the production size 2^15 - 1)."""
    rng = random.Random(window_num * 10000 + buckets_size)
    base = [rand_point(rng) for _ in range(32)]
    mat = []
    for i in range(window_num * buckets_size):
        mat.append(None if i % 11 == 5 else rand_jac(rng, base[rng.randrange(32)]))
    got = cfg.sum_reduction(sum((o.encode_point_be32(p) for p in mat), []), buckets_size, window_num)
    exp = o.sum_reduction(window_num, mat)
    assert [decode_be32_affine(g) for g in got] == [o.to_affine(e) for e in exp]


def test_stage_chain_equals_msm(cfg, msm_pkg):
    """The five stage entry points chained as exec_metal_commands does (msm.rs:189-217)."""
    rng = random.Random(77)
    n = 200
    pts = [rand_point(rng) for _ in range(n)]
    sc = [rng.randrange(o.R_ORDER) for _ in range(n)]
    c, _st, W, bl = o.window_params(n, 7)
    pairs = cfg.prepare_buckets_indices(_scalars_be32(sc), n, c, W)
    pairs = cfg.sort_buckets_indices(pairs)
    pj = [rand_jac(rng, p) for p in pts]
    buckets = cfg.bucket_wise_accumulation(pairs, sum((o.encode_point_be32(p) for p in pj), []), n, W * bl)
    res = cfg.sum_reduction(sum(buckets, []), bl, W)
    out = msm_pkg.final_accumulation(sum(res, []), W, c)
    assert decode_be32_affine(out) == o.msm_naive(sc, pts)


@pytest.mark.gpu
@pytest.mark.parametrize("n,key_bits", [(1, 32), (257, 32), (70001, 20), (17 << 12, 24), (300000, 32)])
def test_sort_pairs_device_in_place(cfg, n, key_bits):
    """Device-resident form of sort_buckets_indices (sort_buckets.rs:15-34 / the reference's sort benchmark):
    ascending keys, same multiset of pairs (the sort post-conditions of sort_buckets.rs:111-125), stable."""
    import numpy as np
    rng = np.random.default_rng(42 + n)
    pairs = np.empty((n, 2), dtype=np.uint32)
    pairs[:, 0] = rng.integers(0, 1 << key_bits, size=n, dtype=np.uint64).astype(np.uint32)
    if key_bits == 32 and n > 4:
        pairs[::7, 0] = 0xFFFFFFFF        # the zero-digit sentinels sort last
    pairs[:, 1] = np.arange(n, dtype=np.uint32)
    d = cfg.alloc(8 * n)
    cfg.to_device(d, pairs.tobytes())
    ms = cfg.sort_pairs_device(d, n, key_bits)
    got = np.frombuffer(cfg.to_host(d, 8 * n), dtype=np.uint32).reshape(n, 2)
    cfg.free(d)
    assert ms >= 0
    order = np.argsort(pairs[:, 0], kind="stable")
    assert np.array_equal(got, pairs[order])
