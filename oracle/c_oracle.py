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
MSM_BEST_ALGORITHM = "chunk-parallel Pippenger, serial Jacobian buckets, signed digits"


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
        for name in ("oracle_msm_best",):
            getattr(L, name).argtypes = [c_char_p, c_char_p, c_size_t, c_int, c_void_p]
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
    return max(1, len(os.sched_getaffinity(0)))


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
