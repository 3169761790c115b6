"""Property tests of the whole path through the C ABI -- the counterpart of the reference's proptest-driven
end-to-end tests (src/metal/msm.rs:581-689: random instances against a CPU MSM, affine equality).  hypothesis
draws the size, the window, the scalar distribution and the point pool; the C oracle (restated msm_best, and
the reference pipeline for small n) is the checker.  Few examples per property: every example is a GPU call."""
import random

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import bn254_ref as o
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

POOL_SEED = o.SEED_BASE + 4000
POOL_N = 512
_pool = {}


def pool():
    if not _pool:
        pb, _ = co.gen_instance(POOL_SEED, POOL_N, True, threads=2)
        _pool["p"] = [pb[64 * i:64 * i + 64] for i in range(POOL_N)]
    return _pool["p"]


SCALAR_KINDS = ("uniform", "zeros30", "zeros90", "small16", "equal", "edge")


def make_scalars(kind, n, rng):
    r = o.R_ORDER
    if kind == "uniform":
        ks = [rng.randrange(r) for _ in range(n)]
    elif kind == "zeros30":
        ks = [0 if rng.random() < 0.3 else rng.randrange(r) for _ in range(n)]
    elif kind == "zeros90":
        ks = [0 if rng.random() < 0.9 else rng.randrange(r) for _ in range(n)]
    elif kind == "small16":
        ks = [rng.randrange(1 << 16) for _ in range(n)]
    elif kind == "equal":
        ks = [rng.randrange(r)] * n
    else:
        edge = [0, 1, r - 1, r - 2, (1 << 253) - 1, 1 << 253, (1 << 14) + 1, (1 << 15) - 1, int("1" * 253, 2)]
        ks = [rng.choice(edge) for _ in range(n)]
    return b"".join(o.encode_scalar_h2c(k) for k in ks)


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n=st.integers(1, 700), window=st.sampled_from([0, 0, 3, 4, 6, 9, 12, 15]), kind=st.sampled_from(SCALAR_KINDS),
       dup=st.booleans(), with_identity=st.booleans(), seed=st.integers(0, 2**32 - 1))
def test_msm_matches_cpu_msm(cfg, n, window, kind, dup, with_identity, seed):
    rng = random.Random(seed)
    pts = pool()
    if dup:                                            # few distinct points: P + P and P + (-P)-free collisions
        pick = [pts[rng.randrange(4)] for _ in range(n)]
    else:
        pick = [pts[rng.randrange(POOL_N)] for _ in range(n)]
    if with_identity and n > 2:
        pick[rng.randrange(n)] = bytes(64)             # halo2curves identity (0, 0)
    pb = b"".join(pick)
    sb = make_scalars(kind, n, rng)
    cfg.set_window_size(window)
    try:
        got = cfg.msm(sb, pb, n)
    finally:
        cfg.set_window_size(0)
    want = co.msm_best(sb, pb, n, 2)
    assert o.decode_jacobian_mont_le(got) == o.decode_jacobian_mont_le(want)
    z = int.from_bytes(got[64:96], "little")
    assert z in (0, o.MONT_R % o.P)                    # normalised output


@settings(max_examples=15, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n=st.integers(2, 300), split=st.floats(0.0, 1.0), kind=st.sampled_from(SCALAR_KINDS),
       seed=st.integers(0, 2**32 - 1))
def test_hybrid_entry_points_match_cpu_msm(cfg, msm_pkg, n, split, kind, seed):
    """msm_best (device filter_zeros) and gpu_with_cpu (any split point) against the same oracle."""
    rng = random.Random(seed)
    pb = b"".join(pool()[rng.randrange(POOL_N)] for _ in range(n))
    sb = make_scalars(kind, n, rng)
    want = o.decode_jacobian_mont_le(co.msm_best(sb, pb, n, 2))
    assert o.decode_jacobian_mont_le(msm_pkg.msm_best(sb, pb, cfg)) == want
    assert o.decode_jacobian_mont_le(msm_pkg.gpu_with_cpu(sb, pb, cfg, split_at=int(split * n), cpu_threads=2)) == want
