"""Opt-in bases cache of the host-slice entry points (msm_amd_set_bases_cache): the reference re-uploads and
re-converts the bases on every call (msm.rs:152-153) although its callers pass the same slice every time
(benches/msm_benchmark.rs:116-121)."""
import ctypes

import pytest

from oracle import bn254_ref as o
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def _same(a, b):
    return o.decode_jacobian_mont_le(a) == o.decode_jacobian_mont_le(b)


@pytest.fixture()
def ccfg(msm_pkg):
    c = msm_pkg.setup_metal_state()
    yield c
    c.close()


def test_cache_hits_give_the_same_points(ccfg, msm_pkg):
    n = 1 << 14
    pts, sc = co.gen_instance(o.SEED_BASE + 314, n)
    sc2 = co.gen_instance(o.SEED_BASE + 315, n)[1]
    want, want2 = co.msm_best(sc, pts, n), co.msm_best(sc2, pts, n)
    off = ccfg.msm(sc, pts, n)
    assert _same(off, want)
    ccfg.set_bases_cache(256 << 20)
    assert ccfg.msm(sc, pts, n) == off                       # miss: fills the entry
    assert ccfg.bases_cache_stats()["misses"] == 1
    for _ in range(3):
        assert ccfg.msm(sc, pts, n) == off                   # hits
        assert _same(ccfg.msm(sc2, pts, n), want2)           # other scalars, same bases
    st = ccfg.bases_cache_stats()
    assert st["hits"] == 6 and st["misses"] == 1 and st["entries"] == 1 and st["bytes"] == 64 * n
    # batch: five instances over the same bases (the criterion bench's shape) -> one entry, hits inside the batch
    outs = ccfg.msm_batch([sc, sc2, sc, sc2, sc], [pts] * 5, [n] * 5)
    assert [_same(x, w) for x, w in zip(outs, [want, want2, want, want2, want])] == [True] * 5
    assert ccfg.bases_cache_stats()["entries"] == 1
    # msm_best goes through the cache too, with and without the zero filter
    assert _same(msm_pkg.msm_best(sc, pts, ccfg), want)
    zsc = bytearray(sc)
    for i in range(0, n, 2):
        zsc[32 * i:32 * i + 32] = bytes(32)
    assert _same(msm_pkg.msm_best(bytes(zsc), pts, ccfg), co.msm_best(bytes(zsc), pts, n))
    ccfg.set_bases_cache(0)
    assert ccfg.bases_cache_stats()["entries"] == 0 and ccfg.msm(sc, pts, n) == off


def test_a_mutated_base_is_detected(ccfg, msm_pkg):
    """Same address, changed content: phase 0 (records 0, S, 2S, ... with S = n / 1024) is re-hashed on every call, the
    other phases in rotation -- a change in phase 0 is caught at once, any other within S calls."""
    n = 1 << 12                                              # S = 4 phases
    pts, sc = co.gen_instance(o.SEED_BASE + 271, n)
    other = co.gen_instance(o.SEED_BASE + 272, 8)[0]
    buf = ctypes.create_string_buffer(pts, len(pts))         # one address for the life of the test
    addr = ctypes.cast(buf, ctypes.c_void_p)
    L = msm_pkg.lib()

    def run():
        out = ctypes.create_string_buffer(96)
        ccfg._check(L.msm_amd_gpu_msm_h2c(ccfg.h, sc, addr, n, out))
        return out.raw

    ccfg.set_bases_cache(64 << 20)
    base = run()
    assert _same(base, co.msm_best(sc, pts, n)) and run() == base
    # (1) record 8 = phase 0 (8 mod 4 == 0): detected by the very next call
    ctypes.memmove(ctypes.addressof(buf) + 64 * 8, other[:64], 64)
    got = run()
    assert _same(got, co.msm_best(sc, buf.raw, n)) and got != base
    assert ccfg.bases_cache_stats()["invalidations"] == 1
    # (2) record 5 = phase 1: detected within S = 4 calls, never later
    base2 = run()
    ctypes.memmove(ctypes.addressof(buf) + 64 * 5, other[64:128], 64)
    want = co.msm_best(sc, buf.raw, n)
    results = [run() for _ in range(5)]
    assert _same(results[-1], want)
    first_ok = next(i for i, r in enumerate(results) if _same(r, want))
    assert first_ok <= 4 and all(_same(r, want) for r in results[first_ok:])
    assert all(r == base2 for r in results[:first_ok])       # until detected: the cached bases, consistently
    assert ccfg.bases_cache_stats()["invalidations"] == 2


def test_budget_and_eviction(ccfg, msm_pkg):
    n = 1 << 12
    insts = [co.gen_instance(o.SEED_BASE + 500 + j, n) for j in range(3)]
    want = [co.msm_best(sc, pts, n) for pts, sc in insts]
    ccfg.set_bases_cache(2 * 64 * n)                         # room for two arrays
    for j in (0, 1, 2, 0, 1, 2):
        assert _same(ccfg.msm(insts[j][1], insts[j][0], n), want[j])
    st = ccfg.bases_cache_stats()
    assert st["entries"] == 2 and st["bytes"] == 2 * 64 * n and st["misses"] == 6 and st["hits"] == 0   # LRU thrash, still right
    for j in (1, 2, 1, 2):
        assert _same(ccfg.msm(insts[j][1], insts[j][0], n), want[j])
    assert ccfg.bases_cache_stats()["hits"] == 4
    # a batch of three arrays with room for two: the third runs uncached, nothing in use is evicted
    outs = ccfg.msm_batch([s for _p, s in insts], [p for p, _s in insts], [n] * 3)
    assert all(_same(x, w) for x, w in zip(outs, want))
    # an array larger than the whole budget is simply not cached
    big = co.gen_instance(o.SEED_BASE + 600, 4 * n)
    assert _same(ccfg.msm(big[1], big[0], 4 * n), co.msm_best(big[1], big[0], 4 * n))
    assert ccfg.bases_cache_stats()["entries"] == 2
    # other layouts are cached under their own key (ark projective: 96-byte records)
    pts, sc = insts[0]
    proj = b"".join(pts[64 * i:64 * i + 64] + o.int_to_le_bytes32(o.fq_to_mont(1)) for i in range(n))
    for _ in range(2):
        assert _same(msm_pkg.metal_msm(proj, sc, ccfg), want[0])
