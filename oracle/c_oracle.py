"""ctypes loader for oracle/libmsm_oracle.so (CPU oracle / CPU baseline -- test infrastructure only)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_int, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsm_oracle.so")
_LIB = None
# what oracle_msm_best does (printed in bench.py's cpu_baseline)
MSM_BEST_ALGORITHM = ("window-parallel Pippenger (halo2curves 0.7 msm_best shape): c = ceil(ln n), Booth digits, "
                      "affine buckets with batched additions (64 per shared inversion), Jacobian side buckets for "
                      "collisions, summation by parts; points additionally split into groups so that all threads have "
                      "a window task")


def build(force=False):
    src = os.path.join(_HERE, "msm_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        L.oracle_gen_instance.argtypes = [c_uint64, c_size_t, c_int, c_void_p, c_void_p, c_int]
        L.oracle_gen_instance.restype = None
        for name in ("oracle_msm_best", "oracle_msm_chunked"):
            getattr(L, name).argtypes = [c_char_p, c_char_p, c_size_t, c_int, c_void_p]
        L.oracle_msm_best_ex.argtypes = [c_char_p, c_char_p, c_size_t, c_int, c_int, c_void_p, c_void_p]
        L.oracle_msm_reference_pipeline.argtypes = [c_char_p, c_char_p, c_size_t, c_uint32, c_void_p]
        L.oracle_msm_naive.argtypes = [c_char_p, c_char_p, c_size_t, c_void_p]
        L.oracle_dlog_instance.argtypes = [c_char_p, c_char_p, c_char_p, c_size_t, c_int, c_void_p, c_void_p]
        for name in ("oracle_fq_mul", "oracle_fq_add", "oracle_fq_sub", "oracle_jac_add"):
            getattr(L, name).argtypes = [c_char_p, c_char_p, c_void_p]
            getattr(L, name).restype = None
        L.oracle_fr_from_mont.argtypes = [c_char_p, c_void_p]
        L.oracle_fr_from_mont.restype = None
        L.oracle_jac_double.argtypes = [c_char_p, c_void_p]
        L.oracle_jac_double.restype = None
        _LIB = L
    return _LIB


def default_threads():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container that sees
    256 logical CPUs but owns 16 of them must not start 256 compute threads)."""
    n = max(1, len(os.sched_getaffinity(0)))
    quota = None
    try:                                    # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def gen_instance(seed, n, scalars_mont=True, threads=None):
    """(points bytes 64n, scalars bytes 32n) in the h2c memory layout."""
    pts = ctypes.create_string_buffer(64 * n)
    sc = ctypes.create_string_buffer(32 * n)
    lib().oracle_gen_instance(seed, n, 1 if scalars_mont else 0, pts, sc, threads or default_threads())
    return pts.raw, sc.raw


def msm_best(scalars: bytes, points: bytes, n: int, threads=None) -> bytes:
    out = ctypes.create_string_buffer(96)
    rc = lib().oracle_msm_best(scalars, points, n, threads or default_threads(), out)
    assert rc == 0
    return out.raw


def msm_best_ex(scalars: bytes, points: bytes, n: int, threads=None, groups=0):
    """(result, info) with info = {window, windows, groups, threads_used}; groups = 1 is the halo2curves shape
    exactly (one task per window), 0 = as many point groups as keep all threads busy."""
    out = ctypes.create_string_buffer(96)
    info = (c_uint32 * 4)()
    rc = lib().oracle_msm_best_ex(scalars, points, n, threads or default_threads(), groups, out, info)
    assert rc == 0
    return out.raw, {"window": info[0], "windows": info[1], "groups": info[2], "threads_used": info[3]}


def msm_chunked(scalars: bytes, points: bytes, n: int, threads=None) -> bytes:
    """Round-1 baseline (points split over threads, serial Jacobian bucket method per slice): a second checker."""
    out = ctypes.create_string_buffer(96)
    rc = lib().oracle_msm_chunked(scalars, points, n, threads or default_threads(), out)
    assert rc == 0
    return out.raw


def msm_reference_pipeline(scalars: bytes, points: bytes, n: int, window_size=0) -> bytes:
    out = ctypes.create_string_buffer(96)
    rc = lib().oracle_msm_reference_pipeline(scalars, points, n, window_size, out)
    assert rc == 0
    return out.raw


def msm_naive(scalars: bytes, points: bytes, n: int) -> bytes:
    out = ctypes.create_string_buffer(96)
    lib().oracle_msm_naive(scalars, points, n, out)
    return out.raw


def dlog_instance(a0: int, d: int, scalars: bytes, n: int, threads=None):
    """Points P_i = (a0 + i d) G (64n bytes) and the expected MSM result (96 bytes)."""
    pts = ctypes.create_string_buffer(64 * n)
    exp = ctypes.create_string_buffer(96)
    rc = lib().oracle_dlog_instance(a0.to_bytes(32, "little"), d.to_bytes(32, "little"), scalars, n,
                                    threads or default_threads(), pts, exp)
    assert rc == 0
    return pts.raw, exp.raw


def _bin2(fn, a: bytes, b: bytes, size):
    out = ctypes.create_string_buffer(size)
    fn(a, b, out)
    return out.raw


def fq_mul(a, b):
    return _bin2(lib().oracle_fq_mul, a, b, 32)


def fq_add(a, b):
    return _bin2(lib().oracle_fq_add, a, b, 32)


def fq_sub(a, b):
    return _bin2(lib().oracle_fq_sub, a, b, 32)


def jac_add(a, b):
    return _bin2(lib().oracle_jac_add, a, b, 96)


def jac_double(a):
    out = ctypes.create_string_buffer(96)
    lib().oracle_jac_double(a, out)
    return out.raw


def fr_from_mont(a):
    out = ctypes.create_string_buffer(32)
    lib().oracle_fr_from_mont(a, out)
    return out.raw
