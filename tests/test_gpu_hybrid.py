"""Hybrid front-end (src/metal/msm.rs:366-507): msm_best with filter_zeros, gpu_with_cpu with the reference's
split policy.  Mirrors test_gpu_cpu_correctness_medium_sample_h2c / test_best_msm_correctness_medium_sample_h2c
(msm.rs:618-689) at sizes the CPU oracle finishes in seconds."""
import random

import pytest

from oracle import bn254_ref as o
from oracle import c_oracle as co
from helpers import h2c_instance_bytes, small_instance

pytestmark = pytest.mark.gpu


def test_reference_split_policy(msm_pkg):
    L = msm_pkg.lib()
    assert L.msm_amd_reference_split(3000) == 1000                    # n/3 below 2^18 (msm.rs:377-378)
    assert L.msm_amd_reference_split(1 << 18) == 1 << 17              # n/2 below 2^20
    assert L.msm_amd_reference_split(1 << 20) == (1 << 20) * 2 // 3   # 2n/3 from 2^20


@pytest.mark.parametrize("zero_frac", [0.0, 0.29, 0.31, 0.9, 1.0])
def test_msm_best_filter_zeros(cfg, msm_pkg, zero_frac):
    """filter_zeros threshold 30 % (msm.rs:470): below it the inputs pass unchanged, above it zero scalars and
    their points are compacted away; the result is the same MSM either way."""
    rng = random.Random(int(zero_frac * 100))
    n = 3000
    pts, sc = small_instance(11, 200)
    pts = [pts[i % 200] for i in range(n)]
    sc = [rng.randrange(1, o.R_ORDER) for _ in range(n)]
    nz = int(round(zero_frac * n))
    for i in rng.sample(range(n), nz):
        sc[i] = 0
    sb, pb = h2c_instance_bytes(pts, sc)
    out = msm_pkg.msm_best(sb, pb, cfg)
    expect = co.msm_best(sb, pb, n, 2)
    assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(expect)
    if zero_frac == 1.0:
        assert o.decode_jacobian_mont_le(out) is None


@pytest.mark.parametrize("n,split", [(1000, None), (1000, 0), (1000, 1000), (4097, 1234)])
def test_gpu_with_cpu(cfg, msm_pkg, n, split):
    pb, sb = co.gen_instance(o.SEED_BASE + 50, n)
    out = msm_pkg.gpu_with_cpu(sb, pb, cfg, split_at=split, cpu_threads=2)
    assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(sb, pb, n, 2))
    z = int.from_bytes(out[64:96], "little")
    assert z in (0, o.MONT_R % o.P)


def test_after_sort_callback_fires_before_accumulation_ends(cfg, msm_pkg):
    """gpu_msm_h2c_sync's condvar hook (msm.rs:237-241, 306-312): the callback runs exactly once, after the sort of
    THIS MSM and -- at a size where accumulation takes over a millisecond -- while accumulation is still pending."""
    from oracle import c_oracle as co
    n = 1 << 20
    pts, sc = co.gen_instance(0xB2540000 + 77, n)
    ref = msm_pkg.gpu_msm_h2c(sc, pts, cfg)
    for _ in range(4):                            # steady state: every one of the ctx's four workspaces has been grown
        assert msm_pkg.gpu_msm_h2c_sync(sc, pts, None, cfg) == ref   # to this size by the SAME (unsplit) path -- growing
    fired = []                                    # one frees and allocates device memory, which synchronises the device in
                                                  # the middle of the enqueue; gpu_msm_h2c splits 2^20 host points into ranges
    out = msm_pkg.gpu_msm_h2c_sync(sc, pts, lambda: fired.append(1), cfg)
    assert fired == [1]
    t = cfg.timings()
    assert t.reserved2[1] == 1.0 and t.reserved2[2] > 0.0, list(t.reserved2)   # device clock: accumulation ended AFTER the mark
    assert out == ref
    assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(sc, pts, n))
    # no callback: same result, state reads "no callback ran"
    out2 = msm_pkg.gpu_msm_h2c_sync(sc, pts, None, cfg)
    assert out2 == out and cfg.timings().reserved2[1] == -1.0


@pytest.mark.parametrize("n", [1, 2, 7, 15, 16, 17, 40])
def test_msm_best_size_dispatch_agrees_on_both_sides(cfg, msm_pkg, n):
    """Sizes below msm_amd_cpu_dispatch_below() take the host bucket method, the rest the GPU: same answers."""
    pts, sc = small_instance(300 + n, n)
    sc[0] = 0
    sb, pb = h2c_instance_bytes(pts, sc)
    assert o.decode_jacobian_mont_le(msm_pkg.msm_best(sb, pb, cfg)) == o.msm_naive(sc, pts)
