"""Round-3 robustness items: (a) an upload staged by a single-instance entry point is ordered before the front end
whatever stream that ends up on (page-locked scalars + point-range split), (b) every host wait for the GPU is bounded
and a stalled device surfaces as MSM_AMD_PIPELINE_ERROR instead of a hang (the reference's gpu_msm_h2c_sync is one
blocking call that always returns, msm.rs:237-349), (c) teardown."""
import ctypes
import os
import time

import pytest

from oracle import bn254_ref as o
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def _with_env(key, value, fn):
    old = os.environ.get(key)
    os.environ[key] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = old


def _same(a, b):
    return o.decode_jacobian_mont_le(a) == o.decode_jacobian_mont_le(b)


def _pinned_copy(cfg, data: bytes):
    """A page-locked host buffer holding `data` (a ctypes array registered with the ctx); caller unregisters."""
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    addr = ctypes.addressof(buf)
    cfg._check(__import__("importlib").import_module("metal-msm-gpu-acceleration_amd").lib().msm_amd_host_register(
        cfg.h, ctypes.c_void_p(addr), len(data)))
    return buf, addr


def test_prepared_msm_with_page_locked_scalars_under_forced_split(cfg, msm_pkg):
    """msm_prepared uploads the scalars asynchronously (DMA, the host does not block) and the split path runs its
    front ends on the front stream: the upload must be ordered before them.  MSM_AMD_SPLIT=4 vs unsplit vs oracle."""
    n = 1 << 18
    pts, sc = co.gen_instance(o.SEED_BASE + 1818, n)
    want = co.msm_best(sc, pts, n)
    d_prep = cfg.bases_upload(pts, n)
    buf, addr = _pinned_copy(cfg, sc)
    L = msm_pkg.lib()
    try:
        out = ctypes.create_string_buffer(96)
        for parts in ("4", "1", "8", "2"):
            for _rep in range(3):
                # scramble the scratch the previous call left behind: a front end that ran ahead of the upload would
                # read the OTHER scalars and produce another point
                other = co.gen_instance(o.SEED_BASE + 99, 4096)[1]
                assert cfg.msm_prepared(other * (n // 4096), d_prep, n) != b""
                st = _with_env("MSM_AMD_SPLIT", parts, lambda: L.msm_amd_msm_prepared(
                    cfg.h, msm_pkg.SCALAR_MONT_LE, ctypes.c_void_p(addr), ctypes.c_void_p(d_prep), n, out))
                assert st == 0, cfg and L.msm_amd_last_error(cfg.h)
                assert _same(out.raw, want), parts
                assert cfg.timings().reserved == int(parts)
        # msm_tables takes the same staged-upload route (tables are never split, the stream choice still applies)
        tb = cfg.tables_build(pts, n)
        try:
            st = L.msm_amd_msm_tables(cfg.h, ctypes.c_void_p(tb), msm_pkg.SCALAR_MONT_LE, ctypes.c_void_p(addr), out)
            assert st == 0 and _same(out.raw, want)
        finally:
            cfg.tables_free(tb)
    finally:
        L.msm_amd_host_unregister(cfg.h, ctypes.c_void_p(addr))
        cfg.free(d_prep)
        del buf


def test_prepared_msm_page_locked_at_2p23_splits_by_itself(cfg, msm_pkg):
    """2^23 points: run_batch_device splits into four ranges on its own; page-locked scalars; equals the unsplit call."""
    n = 1 << 23
    dp, ds = cfg.generate_instance(o.SEED_BASE + 2323, n, True)
    L = msm_pkg.lib()
    try:
        sc = cfg.to_host(ds, 32 * n)
        d_prep = cfg.bases_prepare_device(dp, n)
        buf, addr = _pinned_copy(cfg, sc)
        try:
            out = ctypes.create_string_buffer(96)
            assert L.msm_amd_msm_prepared(cfg.h, msm_pkg.SCALAR_MONT_LE, ctypes.c_void_p(addr), ctypes.c_void_p(d_prep), n, out) == 0
            assert cfg.timings().reserved == 4
            want = _with_env("MSM_AMD_SPLIT", "1", lambda: cfg.msm_batch_device([ds], [dp], [n])[0])
            assert out.raw == want
        finally:
            L.msm_amd_host_unregister(cfg.h, ctypes.c_void_p(addr))
            cfg.free(d_prep)
            del buf
    finally:
        cfg.free(dp)
        cfg.free(ds)


def test_bounded_wait_returns_pipeline_error_and_the_ctx_recovers(msm_pkg):
    """A kernel holds the main stream (msm_amd_test_hold: what a stalled device looks like to the host); a blocking
    MSM with a 150 ms wait bound must come back with PIPELINE_ERROR naming the event and instance, later calls fail
    the same way while the device is busy, and once it is free the ctx works again."""
    cfg = msm_pkg.setup_metal_state()
    try:
        n = 1 << 12
        pts, sc = co.gen_instance(o.SEED_BASE + 12, n)
        want = cfg.msm(sc, pts, n)
        assert _same(want, co.msm_best(sc, pts, n))
        dp, ds = cfg.alloc(64 * n), cfg.alloc(32 * n)
        cfg.to_device(dp, pts)
        cfg.to_device(ds, sc)
        # nothing is sized in advance (round 4): a call that has to grow a workspace or a page-locked slot behind the
        # hold kernel waits for the ctx's streams WITH the bound and fails, instead of blocking inside hipMalloc
        cfg.set_wait_timeout_ms(150)
        hold0 = cfg.test_hold(4000)
        t0 = time.perf_counter()
        with pytest.raises(msm_pkg.MsmError) as e0:
            cfg.msm_batch_device([ds] * 8, [dp] * 8, [n] * 8)   # first-ever call of this shape: every buffer must grow
        assert e0.value.status == msm_pkg.PIPELINE_ERROR and "cannot grow" in str(e0.value)
        assert 0.1 < time.perf_counter() - t0 < 2.0
        cfg.test_release(hold0)
        cfg.set_wait_timeout_ms(60000)
        cfg.synchronize()
        assert cfg.msm_batch_device([ds] * 8, [dp] * 8, [n] * 8) == [want] * 8   # the ctx recovered; now it is sized
        for _ in range(4):                                       # ... and so is every workspace for a lone call's geometry
            assert cfg.msm_batch_device([ds], [dp], [n])[0] == want
        cfg.set_wait_timeout_ms(150)
        hold = cfg.test_hold(4000)
        t0 = time.perf_counter()
        with pytest.raises(msm_pkg.MsmError) as e:
            cfg.msm_batch_device([ds], [dp], [n])
        waited = time.perf_counter() - t0
        assert e.value.status == msm_pkg.PIPELINE_ERROR
        assert "timed out after 150 ms" in str(e.value) and "'reduce' of instance 0" in str(e.value)
        assert 0.1 < waited < 2.0
        with pytest.raises(msm_pkg.MsmError) as e2:            # still held: nothing new is enqueued behind a stall
            cfg.msm_batch_device([ds], [dp], [n])
        assert "still busy" in str(e2.value)
        with pytest.raises(msm_pkg.MsmError):
            cfg.synchronize()                                   # bounded as well
        cfg.test_release(hold)
        cfg.set_wait_timeout_ms(60000)
        cfg.synchronize()
        assert cfg.msm_batch_device([ds], [dp], [n])[0] == want   # the abandoned batch was released, results are right
        # submit / wait API: after a timeout the ticket stays valid and can be waited for again
        cfg.set_wait_timeout_ms(150)
        hold = cfg.test_hold(4000)
        h = cfg.submit_batch_device([ds, ds], [dp, dp], [n, n])
        with pytest.raises(msm_pkg.MsmError) as e3:
            cfg.wait_batch(h)
        assert "the ticket stays valid" in str(e3.value)
        cfg.test_release(hold)
        cfg.set_wait_timeout_ms(60000)
        assert cfg.wait_batch(h) == [want, want]
        cfg.free(dp)
        cfg.free(ds)
    finally:
        cfg.close()


def test_hold_kernel_ends_by_itself(msm_pkg):
    """The test aid cannot outlive its own bound: unreleased, it exits after max_ms and destroy returns."""
    cfg = msm_pkg.setup_metal_state()
    cfg.test_hold(300)
    t0 = time.perf_counter()
    cfg.synchronize()
    assert time.perf_counter() - t0 < 5.0
    cfg.close()


def test_create_use_destroy_many_contexts(msm_pkg):
    """Teardown order (device-wide sync, events, memory, streams): contexts come and go without leaking state."""
    n = 1 << 10
    pts, sc = co.gen_instance(o.SEED_BASE + 1, n)
    want = co.msm_best(sc, pts, n)
    for _ in range(8):
        c = msm_pkg.setup_metal_state()
        h = None
        try:
            assert _same(c.msm(sc, pts, n), want)
            dp, ds = c.generate_instance(o.SEED_BASE + 2, 1 << 14, True)
            h = c.submit_batch_device([ds] * 3, [dp] * 3, [1 << 14] * 3)    # destroyed with work in flight
        finally:
            c.close()


def test_staged_upload_of_pageable_slices_gives_the_same_points(msm_pkg):
    """MSM_AMD_STAGED_UPLOAD=1: pageable host slices go through the library's own page-locked ring (helper threads +
    DMA per 4 MiB chunk) instead of the runtime's staged copy -- the default on HIP runtimes older than 7.2, whose
    pageable copies serialise with kernels.  Same results, batch and single call, odd sizes included."""
    sizes = [(1 << 19) + 12345, 1 << 18, (1 << 20) - 7]
    inst = [co.gen_instance(o.SEED_BASE + 7700 + j, n) for j, n in enumerate(sizes)]
    want = [co.msm_best(sc, pts, n) for (pts, sc), n in zip(inst, sizes)]
    for staged in ("1", "0"):
        cfg = _with_env("MSM_AMD_STAGED_UPLOAD", staged, lambda: msm_pkg.setup_metal_state())
        try:
            outs = _with_env("MSM_AMD_STAGED_UPLOAD", staged,
                             lambda: cfg.msm_batch([s for _p, s in inst], [p for p, _s in inst], sizes))
            assert all(_same(a, b) for a, b in zip(outs, want)), staged
            one = _with_env("MSM_AMD_STAGED_UPLOAD", staged, lambda: msm_pkg.gpu_msm_h2c(inst[2][1], inst[2][0], cfg))
            assert _same(one, want[2])
            cfg.set_bases_cache(1 << 30)
            for _ in range(2):
                outs = cfg.msm_batch([s for _p, s in inst], [p for p, _s in inst], sizes)
                assert all(_same(a, b) for a, b in zip(outs, want)), ("cache", staged)
        finally:
            cfg.close()
