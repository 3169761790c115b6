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


def test_full_verification_catches_any_mutated_record_on_the_first_call(ccfg, msm_pkg):
    """msm_amd_set_bases_cache_verify(full): EVERY record of the caller's array is re-hashed on every hit, so a single
    changed record outside phase 0 -- which the sampled check may miss for up to n / 1024 calls -- is caught by the
    very next call.  msm_amd_bases_cache_invalidate is the explicit form of the same."""
    import time
    n = 1 << 16                                              # 64 phases: sampling could take 64 calls
    pts, sc = co.gen_instance(o.SEED_BASE + 281, n)
    other = co.gen_instance(o.SEED_BASE + 282, 4)[0]
    buf = ctypes.create_string_buffer(pts, len(pts))
    addr = ctypes.cast(buf, ctypes.c_void_p)
    L = msm_pkg.lib()

    def run():
        out = ctypes.create_string_buffer(96)
        ccfg._check(L.msm_amd_gpu_msm_h2c(ccfg.h, sc, addr, n, out))
        return out.raw

    ccfg.set_bases_cache(64 << 20)
    ccfg.set_bases_cache_verify(True)
    base = run()
    assert _same(base, co.msm_best(sc, pts, n)) and run() == base and ccfg.bases_cache_stats()["hits"] == 1
    for k, rec in enumerate((37, n - 1, 12345)):             # none of them in phase 0 (multiples of 64)
        ctypes.memmove(ctypes.addressof(buf) + 64 * rec, other[64 * k:64 * k + 64], 64)
        got = run()                                          # the FIRST call after the change
        assert _same(got, co.msm_best(sc, buf.raw, n)), rec
        assert ccfg.bases_cache_stats()["invalidations"] == k + 1
        assert run() == got                                  # refilled, hit again
    # explicit invalidation: sampled mode, a change the sampling would not see at once, the caller says so
    ccfg.set_bases_cache_verify(False)
    before = run()
    ctypes.memmove(ctypes.addressof(buf) + 64 * 41, other[192:256], 64)
    L.msm_amd_bases_cache_invalidate(ccfg.h, addr)
    after = run()
    assert _same(after, co.msm_best(sc, buf.raw, n)) and after != before
    # what the full check costs (printed with -s; profiles/r04_bases_cache_verify.txt keeps a run)
    n2 = 1 << 20
    pts2, sc2 = co.gen_instance(o.SEED_BASE + 283, n2)
    for full in (False, True):
        ccfg.set_bases_cache_verify(full)
        ccfg.msm(sc2, pts2, n2)
        t0 = time.perf_counter()
        for _ in range(5):
            ccfg.msm(sc2, pts2, n2)
        print("bases cache hit, 2^20 points, verify=%s: %.3f ms per blocking call" %
              ("full" if full else "sampled", (time.perf_counter() - t0) / 5 * 1e3))
