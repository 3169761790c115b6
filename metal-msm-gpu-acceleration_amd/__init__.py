"""Python binding of libmsm_amd.so for tests and bench.py (ctypes over the C ABI of include/msm_amd.h).

The product is the C/HIP library; this module only loads it and mirrors the reference's entry-point
names (`mopro_msm::metal::msm::{gpu_msm_h2c, metal_msm, setup_metal_state, ...}`, src/metal/msm.rs) so
that tests read like the reference's own.  There is NO CPU fallback: if the library is missing or no
gfx950 device is present the calls raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSM_AMD_LIB points development A/B runs at another build of the same library; the default is the in-tree one
LIB_PATH = os.environ.get("MSM_AMD_LIB") or os.path.join(_HERE, "libmsm_amd.so")

(OK, DEVICE_NOT_FOUND, LIBRARY_ERROR, FUNCTION_ERROR, PIPELINE_ERROR, INPUT_ERROR, FILE_OPEN_ERROR,
 DESERIALIZATION_ERROR, INVALID_DATA) = range(9)
SCALAR_MONT_LE, SCALAR_CANON_LE, SCALAR_CANON_BE32 = 0, 1, 2
POINT_H2C_AFFINE, POINT_ARK_PROJECTIVE, POINT_ARK_AFFINE, POINT_JAC_BE32, POINT_PREPARED, POINT_TABLES = 0, 1, 2, 3, 4, 5
POINT_BYTES = {POINT_H2C_AFFINE: 64, POINT_ARK_PROJECTIVE: 96, POINT_ARK_AFFINE: 72, POINT_JAC_BE32: 96}
(OP_UINT_ADD, OP_UINT_SUB, OP_UINT_PROD, OP_UINT_SHL, OP_UINT_SHR, OP_FP_ADD, OP_FP_SUB, OP_FP_MUL, OP_FP_NEG,
 OP_FP_POW, OP_EC_ADD, OP_EC_MUL, OP_EC_MADD, OP_EC_DBL, OP_FP29_MUL, OP_FP29_SQR, OP_FP29_SUB_K4E30,
 OP_FP29_SUB_K8E30, OP_FP29_SUB_K8E31, OP_FP29_SUB_K16E30, OP_FP29_SUB_K16E31, OP_FP29_ROUNDTRIP, OP_EC29_MADD,
 OP_EC29_ADD, OP_EC29_MADD_CHAIN, OP_EC29_ADD_CHAIN, OP_EC29_MMADD, OP_H64_FP_MUL, OP_H64_FP_ADD, OP_H64_FP_SUB,
 OP_H64_EC_ADD, OP_H64_EC_DBL, OP_FP29_MUL_KARATSUBA, OP_FP29_LOCKSTEP_PAIR, OP_FP29_LOCKSTEP_MIX,
 OP_FP29_LOCKSTEP_TRIPLE, OP_FP29_MUL2_KARATSUBA, OP_H64_FP_INV, OP_H64_FP_INV_FERMAT) = range(39)


def op_is_point(op):
    return 10 <= op <= 13 or 22 <= op <= 26 or op in (OP_H64_EC_ADD, OP_H64_EC_DBL)

# every symbol include/msm_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "msm_amd_init", "msm_amd_init_reusable", "msm_amd_get_global", "msm_amd_destroy", "msm_amd_strerror",
    "msm_amd_last_error", "msm_amd_set_window_size", "msm_amd_auto_window_size", "msm_amd_auto_window_size_lone", "msm_amd_gpu_msm_h2c",
    "msm_amd_gpu_msm_h2c_sync", "msm_amd_cpu_dispatch_below", "msm_amd_host_register", "msm_amd_host_unregister",
    "msm_amd_metal_msm_ark", "msm_amd_msm", "msm_amd_msm_batch", "msm_amd_msm_best", "msm_amd_gpu_with_cpu",
    "msm_amd_reference_split", "msm_amd_msm_device",
    "msm_amd_msm_batch_device", "msm_amd_submit_batch_device", "msm_amd_wait_batch", "msm_amd_device_alloc", "msm_amd_device_free", "msm_amd_copy_to_device",
    "msm_amd_copy_to_host", "msm_amd_stream", "msm_amd_synchronize", "msm_amd_generate_instance",
    "msm_amd_prepare_buckets_indices", "msm_amd_sort_buckets_indices", "msm_amd_bucket_wise_accumulation",
    "msm_amd_sum_reduction", "msm_amd_final_accumulation", "msm_amd_test_op", "msm_amd_test_op_host",
    "msm_amd_last_timings",
    "msm_amd_algorithmic_bytes", "msm_amd_version",
    "msm_amd_instances_save", "msm_amd_instances_open", "msm_amd_instances_count", "msm_amd_instances_size",
    "msm_amd_instances_read", "msm_amd_instances_close", "msm_amd_instances_default_path", "msm_amd_to_wire",
    "msm_amd_from_wire", "msm_amd_sort_pairs_device", "msm_amd_bases_upload", "msm_amd_bases_prepare_device",
    "msm_amd_msm_prepared", "msm_amd_sum_points", "msm_amd_tables_build", "msm_amd_tables_build_device",
    "msm_amd_tables_info", "msm_amd_tables_free", "msm_amd_msm_tables",
    "msm_amd_set_wait_timeout_ms", "msm_amd_set_bases_cache", "msm_amd_bases_cache_stats", "msm_amd_test_hold",
    "msm_amd_test_release", "msm_amd_msm_batch_multi", "msm_amd_msm_batch_multi_device", "msm_amd_msm_range_multi", "msm_amd_shard_range",
    "msm_amd_submit_batch_multi_device", "msm_amd_wait_batch_multi", "msm_amd_bases_cache_invalidate",
    "msm_amd_set_bases_cache_verify",
    "msm_amd_scalar_bytes", "msm_amd_point_bytes", "msm_amd_shard_owner",
    "msm_amd_shard_count", "msm_amd_ctx_device", "msm_amd_pin_thread_to_device", "msm_amd_gather_init",
    "msm_amd_gather_size", "msm_amd_gather_all", "msm_amd_gather_last_error", "msm_amd_gather_destroy",
    "msm_amd_host_msm", "msm_amd_tuned_split", "msm_amd_host_threads", "msm_amd_generate_instance_host", "msm_amd_test_op_ifma",
]


AFTER_SORT_FN = ctypes.CFUNCTYPE(None, c_void_p)   # msm_amd_after_sort_fn


class Timings(ctypes.Structure):
    _fields_ = [("convert_ms", c_float), ("digits_ms", c_float), ("sort_ms", c_float), ("accumulate_ms", c_float),
                ("reduce_ms", c_float), ("final_ms", c_float), ("total_gpu_ms", c_float), ("n", c_uint32),
                ("window_size", c_uint32), ("num_windows", c_uint32), ("reserved", c_uint32),
                ("accumulate_kernel_ms", c_float), ("reserved2", c_float * 3)]


class MsmError(RuntimeError):
    """Counterpart of MetalError (src/metal/abstraction/errors.rs:4-19)."""

    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"msm_amd status {status}: {_lib().msm_amd_strerror(status).decode()} {detail}")


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(LIB_PATH)
        L.msm_amd_strerror.restype = c_char_p
        L.msm_amd_strerror.argtypes = [c_int]
        L.msm_amd_last_error.restype = c_char_p
        L.msm_amd_last_error.argtypes = [c_void_p]
        L.msm_amd_version.restype = c_char_p
        L.msm_amd_init.argtypes = [c_int, POINTER(c_void_p)]
        L.msm_amd_bases_upload.argtypes = [c_void_p, c_int, c_void_p, c_size_t, POINTER(c_void_p)]
        L.msm_amd_bases_prepare_device.argtypes = [c_void_p, c_int, c_void_p, c_size_t, c_void_p]
        L.msm_amd_msm_prepared.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
        L.msm_amd_tables_build.argtypes = [c_void_p, c_int, c_void_p, c_size_t, c_uint32, POINTER(c_void_p)]
        L.msm_amd_tables_build_device.argtypes = [c_void_p, c_int, c_void_p, c_size_t, c_uint32, POINTER(c_void_p)]
        L.msm_amd_tables_info.argtypes = [c_void_p, c_void_p, POINTER(c_size_t), POINTER(c_uint32), POINTER(c_uint32),
                                          POINTER(c_size_t)]
        L.msm_amd_tables_free.argtypes = [c_void_p, c_void_p]
        L.msm_amd_msm_tables.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p]
        L.msm_amd_sum_points.argtypes = [c_void_p, c_size_t, c_void_p]
        L.msm_amd_sort_pairs_device.argtypes = [c_void_p, c_void_p, c_size_t, c_uint32, POINTER(c_float)]
        L.msm_amd_instances_save.argtypes = [c_char_p, c_size_t, POINTER(c_size_t), POINTER(c_void_p),
                                             POINTER(c_void_p)]
        L.msm_amd_instances_open.argtypes = [c_char_p, POINTER(c_void_p)]
        L.msm_amd_instances_count.argtypes = [c_void_p]
        L.msm_amd_instances_count.restype = c_size_t
        L.msm_amd_instances_size.argtypes = [c_void_p, c_size_t]
        L.msm_amd_instances_size.restype = c_size_t
        L.msm_amd_instances_read.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p]
        L.msm_amd_instances_close.argtypes = [c_void_p]
        L.msm_amd_instances_close.restype = None
        L.msm_amd_instances_default_path.argtypes = [c_char_p, c_uint32, c_uint32, c_char_p, c_size_t]
        L.msm_amd_instances_default_path.restype = c_size_t
        L.msm_amd_to_wire.argtypes = [c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]
        L.msm_amd_from_wire.argtypes = [c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]
        L.msm_amd_init_reusable.argtypes = [POINTER(c_void_p)]
        L.msm_amd_get_global.argtypes = [POINTER(c_void_p)]
        L.msm_amd_destroy.argtypes = [c_void_p]
        L.msm_amd_destroy.restype = None
        L.msm_amd_set_window_size.argtypes = [c_void_p, c_uint32]
        L.msm_amd_auto_window_size.argtypes = [c_size_t]
        L.msm_amd_auto_window_size.restype = c_uint32
        L.msm_amd_auto_window_size_lone.argtypes = [c_size_t]
        L.msm_amd_auto_window_size_lone.restype = c_uint32
        L.msm_amd_gpu_msm_h2c.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.msm_amd_metal_msm_ark.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.msm_amd_gpu_msm_h2c_sync.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, AFTER_SORT_FN, c_void_p, c_void_p]
        L.msm_amd_host_register.argtypes = [c_void_p, c_void_p, c_size_t]
        L.msm_amd_host_unregister.argtypes = [c_void_p, c_void_p]
        L.msm_amd_cpu_dispatch_below.argtypes = []
        L.msm_amd_cpu_dispatch_below.restype = c_size_t
        L.msm_amd_msm.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
        L.msm_amd_msm_batch.argtypes = [c_void_p, c_int, c_int, c_size_t, POINTER(c_void_p), POINTER(c_void_p),
                                        POINTER(c_size_t), c_void_p]
        L.msm_amd_msm_best.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.msm_amd_gpu_with_cpu.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_void_p]
        L.msm_amd_reference_split.argtypes = [c_size_t]
        L.msm_amd_reference_split.restype = c_size_t
        L.msm_amd_msm_device.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
        L.msm_amd_msm_batch_device.argtypes = [c_void_p, c_int, c_int, c_size_t, POINTER(c_void_p),
                                               POINTER(c_void_p), POINTER(c_size_t), c_void_p]
        L.msm_amd_submit_batch_device.argtypes = [c_void_p, c_int, c_int, c_size_t, POINTER(c_void_p),
                                                  POINTER(c_void_p), POINTER(c_size_t), c_void_p, POINTER(c_int)]
        L.msm_amd_wait_batch.argtypes = [c_void_p, c_int]
        L.msm_amd_device_alloc.argtypes = [c_void_p, c_size_t, POINTER(c_void_p)]
        L.msm_amd_device_free.argtypes = [c_void_p, c_void_p]
        L.msm_amd_copy_to_device.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
        L.msm_amd_copy_to_host.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
        L.msm_amd_stream.argtypes = [c_void_p]
        L.msm_amd_stream.restype = c_void_p
        L.msm_amd_synchronize.argtypes = [c_void_p]
        L.msm_amd_generate_instance.argtypes = [c_void_p, c_uint64, c_size_t, c_int, c_void_p, c_void_p]
        L.msm_amd_prepare_buckets_indices.argtypes = [c_void_p, c_void_p, c_size_t, c_uint32, c_uint32, c_void_p]
        L.msm_amd_sort_buckets_indices.argtypes = [c_void_p, c_void_p, c_size_t]
        L.msm_amd_bucket_wise_accumulation.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_uint32,
                                                       c_void_p]
        L.msm_amd_sum_reduction.argtypes = [c_void_p, c_void_p, c_uint32, c_uint32, c_void_p]
        L.msm_amd_final_accumulation.argtypes = [c_void_p, c_uint32, c_uint32, c_void_p]
        L.msm_amd_test_op.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t]
        L.msm_amd_test_op_host.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_size_t]
        L.msm_amd_last_timings.argtypes = [c_void_p, POINTER(Timings)]
        L.msm_amd_algorithmic_bytes.argtypes = [c_size_t, c_uint32, c_int]
        L.msm_amd_algorithmic_bytes.restype = c_uint64
        L.msm_amd_set_wait_timeout_ms.argtypes = [c_void_p, c_uint32]
        L.msm_amd_set_bases_cache.argtypes = [c_void_p, c_size_t]
        L.msm_amd_bases_cache_stats.argtypes = [c_void_p, POINTER(c_uint64)]
        L.msm_amd_bases_cache_invalidate.argtypes = [c_void_p, c_void_p]
        L.msm_amd_set_bases_cache_verify.argtypes = [c_void_p, c_int]
        L.msm_amd_test_hold.argtypes = [c_void_p, c_uint32, POINTER(c_void_p)]
        L.msm_amd_test_release.argtypes = [c_void_p, c_void_p]
        L.msm_amd_msm_batch_multi.argtypes = [POINTER(c_void_p), c_size_t, c_int, c_int, c_size_t, POINTER(c_void_p),
                                              POINTER(c_void_p), POINTER(c_size_t), c_void_p]
        L.msm_amd_msm_batch_multi_device.argtypes = L.msm_amd_msm_batch_multi.argtypes
        L.msm_amd_submit_batch_multi_device.argtypes = L.msm_amd_msm_batch_multi.argtypes + [POINTER(c_void_p)]
        L.msm_amd_wait_batch_multi.argtypes = [c_void_p]
        L.msm_amd_msm_range_multi.argtypes = [POINTER(c_void_p), c_size_t, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                              c_void_p]
        L.msm_amd_shard_range.argtypes = [c_size_t, c_size_t, c_size_t, POINTER(c_size_t), POINTER(c_size_t)]
        L.msm_amd_shard_range.restype = None
        L.msm_amd_scalar_bytes.argtypes = [c_int]
        L.msm_amd_scalar_bytes.restype = c_size_t
        L.msm_amd_point_bytes.argtypes = [c_int]
        L.msm_amd_point_bytes.restype = c_size_t
        L.msm_amd_shard_owner.argtypes = [c_size_t, c_size_t]
        L.msm_amd_shard_owner.restype = c_size_t
        L.msm_amd_shard_count.argtypes = [c_size_t, c_size_t, c_size_t]
        L.msm_amd_shard_count.restype = c_size_t
        L.msm_amd_ctx_device.argtypes = [c_void_p]
        L.msm_amd_pin_thread_to_device.argtypes = [c_int]
        L.msm_amd_gather_init.argtypes = [POINTER(c_int), c_int, POINTER(c_void_p)]
        L.msm_amd_gather_size.argtypes = [c_void_p]
        L.msm_amd_gather_all.argtypes = [c_void_p, POINTER(c_void_p), c_size_t, POINTER(c_void_p)]
        L.msm_amd_gather_last_error.argtypes = [c_void_p]
        L.msm_amd_gather_last_error.restype = c_char_p
        L.msm_amd_gather_destroy.argtypes = [c_void_p]
        L.msm_amd_gather_destroy.restype = None
        L.msm_amd_host_msm.argtypes = [c_int, c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        L.msm_amd_generate_instance_host.argtypes = [c_uint64, c_size_t, c_int, c_void_p, c_void_p, c_int]
        L.msm_amd_test_op_ifma.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_size_t]
        L.msm_amd_tuned_split.argtypes = [c_size_t]
        L.msm_amd_tuned_split.restype = c_size_t
        _LIB = L
    return _LIB


def lib():
    return _lib()


def _u32buf(seq):
    arr = (c_uint32 * len(seq))(*seq)
    return arr


class MsmConfig:
    """MetalMsmConfig (msm.rs:59-63): device + stream + kernels.  `setup_metal_state()` makes one."""

    def __init__(self, device=-1, _handle=None, _owned=True):
        self._owned = _owned
        if _handle is not None:
            self.h = _handle
            return
        h = c_void_p()
        st = _lib().msm_amd_init(device, ctypes.byref(h))
        if st != OK:
            raise MsmError(st)
        self.h = h

    def _check(self, st):
        if st != OK:
            raise MsmError(st, _lib().msm_amd_last_error(self.h).decode())

    def close(self):
        if self.h and self._owned:
            _lib().msm_amd_destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- whole MSM -------------------------------------------------------------------------
    def set_window_size(self, c):
        self._check(_lib().msm_amd_set_window_size(self.h, c))

    def msm(self, scalars: bytes, points: bytes, n: int, scalar_layout=SCALAR_MONT_LE,
            point_layout=POINT_H2C_AFFINE) -> bytes:
        out = ctypes.create_string_buffer(96)
        self._check(_lib().msm_amd_msm(self.h, scalar_layout, point_layout, scalars, points, n, out))
        return out.raw

    def msm_batch(self, scalars_list, points_list, ns, scalar_layout=SCALAR_MONT_LE,
                  point_layout=POINT_H2C_AFFINE):
        k = len(ns)
        sp = (c_void_p * k)(*[ctypes.cast(ctypes.c_char_p(s), c_void_p) for s in scalars_list])
        if point_layout in (POINT_PREPARED, POINT_TABLES):    # device pointers / table handles, not host bytes
            pp = (c_void_p * k)(*points_list)
        else:
            pp = (c_void_p * k)(*[ctypes.cast(ctypes.c_char_p(p), c_void_p) for p in points_list])
        nn = (c_size_t * k)(*ns)
        out = ctypes.create_string_buffer(96 * k)
        self._check(_lib().msm_amd_msm_batch(self.h, scalar_layout, point_layout, k, sp, pp, nn, out))
        return [out.raw[96 * i:96 * i + 96] for i in range(k)]

    def msm_batch_device(self, d_scalars, d_points, ns, scalar_layout=SCALAR_MONT_LE,
                         point_layout=POINT_H2C_AFFINE):
        k = len(ns)
        sp = (c_void_p * k)(*d_scalars)
        pp = (c_void_p * k)(*d_points)
        nn = (c_size_t * k)(*ns)
        out = ctypes.create_string_buffer(96 * k)
        self._check(_lib().msm_amd_msm_batch_device(self.h, scalar_layout, point_layout, k, sp, pp, nn, out))
        return [out.raw[96 * i:96 * i + 96] for i in range(k)]

    def submit_batch_device(self, d_scalars, d_points, ns, scalar_layout=SCALAR_MONT_LE,
                            point_layout=POINT_H2C_AFFINE):
        """Enqueue a batch; returns a handle for wait_batch (pipelined form of msm_batch_device)."""
        k = len(ns)
        sp = (c_void_p * k)(*d_scalars)
        pp = (c_void_p * k)(*d_points)
        nn = (c_size_t * k)(*ns)
        out = ctypes.create_string_buffer(96 * k)
        ticket = c_int(-1)
        self._check(_lib().msm_amd_submit_batch_device(self.h, scalar_layout, point_layout, k, sp, pp, nn, out,
                                                       ctypes.byref(ticket)))
        return (ticket.value, out, k)

    def wait_batch(self, handle):
        ticket, out, k = handle
        self._check(_lib().msm_amd_wait_batch(self.h, ticket))
        return [out.raw[96 * i:96 * i + 96] for i in range(k)]

    def set_wait_timeout_ms(self, ms: int):
        """Upper bound of every host wait for the GPU (0 = none); a wait that reaches it raises PIPELINE_ERROR."""
        self._check(_lib().msm_amd_set_wait_timeout_ms(self.h, ms))

    def set_bases_cache(self, max_bytes: int):
        """Opt-in cache of converted bases for the host-slice entry points (0 = off)."""
        self._check(_lib().msm_amd_set_bases_cache(self.h, max_bytes))

    def bases_cache_invalidate(self, data=None):
        """The caller's word that the points array `data` (a bytes object handed over earlier; None: every array)
        changed in place: its cache entries are dropped."""
        ptr = None if data is None else ctypes.cast(ctypes.c_char_p(data), c_void_p)
        self._check(_lib().msm_amd_bases_cache_invalidate(self.h, ptr))

    def set_bases_cache_verify(self, full: bool):
        self._check(_lib().msm_amd_set_bases_cache_verify(self.h, 1 if full else 0))

    def bases_cache_stats(self):
        st = (c_uint64 * 5)()
        self._check(_lib().msm_amd_bases_cache_stats(self.h, st))
        return {"hits": st[0], "misses": st[1], "invalidations": st[2], "bytes": st[3], "entries": st[4]}

    def test_hold(self, max_ms: int):
        h = c_void_p()
        self._check(_lib().msm_amd_test_hold(self.h, max_ms, ctypes.byref(h)))
        return h

    def test_release(self, handle):
        self._check(_lib().msm_amd_test_release(self.h, handle))

    def device(self) -> int:
        return _lib().msm_amd_ctx_device(self.h)

    def host_register(self, data: bytes):
        """Page-lock the memory of a bytes object (keep it alive until host_unregister): DMA uploads."""
        addr = ctypes.cast(ctypes.c_char_p(data), c_void_p)
        self._check(_lib().msm_amd_host_register(self.h, addr, len(data)))

    def host_unregister(self, data: bytes):
        self._check(_lib().msm_amd_host_unregister(self.h, ctypes.cast(ctypes.c_char_p(data), c_void_p)))

    # ---- device memory ---------------------------------------------------------------------
    def alloc(self, nbytes) -> int:
        p = c_void_p()
        self._check(_lib().msm_amd_device_alloc(self.h, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, dptr):
        self._check(_lib().msm_amd_device_free(self.h, c_void_p(dptr)))

    def to_device(self, dptr, data: bytes):
        self._check(_lib().msm_amd_copy_to_device(self.h, c_void_p(dptr), data, len(data)))

    def to_host(self, dptr, nbytes) -> bytes:
        out = ctypes.create_string_buffer(nbytes)
        self._check(_lib().msm_amd_copy_to_host(self.h, out, c_void_p(dptr), nbytes))
        return out.raw

    def generate_instance(self, seed, n, scalars_mont=True):
        """Device-resident synthetic instance: returns (d_points, d_scalars)."""
        dp = self.alloc(64 * n)
        ds = self.alloc(32 * n)
        self._check(_lib().msm_amd_generate_instance(self.h, seed, n, 1 if scalars_mont else 0, c_void_p(dp),
                                                     c_void_p(ds)))
        return dp, ds

    def stream(self) -> int:
        return _lib().msm_amd_stream(self.h)

    def synchronize(self):
        self._check(_lib().msm_amd_synchronize(self.h))

    def timings(self) -> Timings:
        t = Timings()
        self._check(_lib().msm_amd_last_timings(self.h, ctypes.byref(t)))
        return t

    # ---- stages (reference wire layout: lists of u32) ----------------------------------------
    def prepare_buckets_indices(self, scalars_be32, n, window_size, num_windows):
        out = (c_uint32 * (n * num_windows * 2))()
        self._check(_lib().msm_amd_prepare_buckets_indices(self.h, _u32buf(scalars_be32), n, window_size,
                                                           num_windows, out))
        return [(out[2 * i], out[2 * i + 1]) for i in range(n * num_windows)]

    def sort_buckets_indices(self, pairs):
        flat = [v for pr in pairs for v in pr]
        buf = _u32buf(flat)
        self._check(_lib().msm_amd_sort_buckets_indices(self.h, buf, len(pairs)))
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(len(pairs))]

    # ---- persistent bases (the reference re-uploads them per call, msm.rs:152-153) -----------
    def bases_upload(self, points: bytes, n: int, point_layout=POINT_H2C_AFFINE) -> int:
        """Convert once, keep resident; returns a device pointer to pass with POINT_PREPARED (free with .free)."""
        p = c_void_p()
        self._check(_lib().msm_amd_bases_upload(self.h, point_layout, points, n, ctypes.byref(p)))
        return p.value

    def bases_prepare_device(self, d_points, n, point_layout=POINT_H2C_AFFINE) -> int:
        out = self.alloc(64 * n)
        self._check(_lib().msm_amd_bases_prepare_device(self.h, point_layout, c_void_p(d_points), n, c_void_p(out)))
        return out

    def msm_prepared(self, scalars: bytes, d_prepared, n, scalar_layout=SCALAR_MONT_LE) -> bytes:
        out = ctypes.create_string_buffer(96)
        self._check(_lib().msm_amd_msm_prepared(self.h, scalar_layout, scalars, c_void_p(d_prepared), n, out))
        return out.raw

    # ---- precomputed window tables for fixed bases (beyond the reference) --------
    def tables_build(self, points: bytes, n: int, point_layout=POINT_H2C_AFFINE, window_size=0) -> int:
        """Returns a table handle: pass it as the points pointer with POINT_TABLES, free with tables_free."""
        h = c_void_p()
        self._check(_lib().msm_amd_tables_build(self.h, point_layout, points, n, window_size, ctypes.byref(h)))
        return h.value

    def tables_build_device(self, d_points, n, point_layout=POINT_H2C_AFFINE, window_size=0) -> int:
        h = c_void_p()
        self._check(_lib().msm_amd_tables_build_device(self.h, point_layout, c_void_p(d_points), n, window_size,
                                                       ctypes.byref(h)))
        return h.value

    def tables_info(self, tables):
        n, nbytes, c, W = c_size_t(), c_size_t(), c_uint32(), c_uint32()
        self._check(_lib().msm_amd_tables_info(self.h, c_void_p(tables), ctypes.byref(n), ctypes.byref(c),
                                               ctypes.byref(W), ctypes.byref(nbytes)))
        return {"n": n.value, "window_size": c.value, "num_windows": W.value, "device_bytes": nbytes.value}

    def tables_free(self, tables):
        self._check(_lib().msm_amd_tables_free(self.h, c_void_p(tables)))

    def msm_tables(self, scalars: bytes, tables, scalar_layout=SCALAR_MONT_LE) -> bytes:
        out = ctypes.create_string_buffer(96)
        self._check(_lib().msm_amd_msm_tables(self.h, c_void_p(tables), scalar_layout, scalars, out))
        return out.raw

    def sort_pairs_device(self, d_pairs, n_pairs, key_bits=32) -> float:
        """In-place device sort of (key, value) u32 pairs; returns the device time in ms."""
        ms = c_float()
        self._check(_lib().msm_amd_sort_pairs_device(self.h, c_void_p(d_pairs), n_pairs, key_bits, ctypes.byref(ms)))
        return ms.value

    def bucket_wise_accumulation(self, sorted_pairs, points_be32, n_points, total_buckets):
        flat = [v for pr in sorted_pairs for v in pr]
        out = (c_uint32 * (total_buckets * 24))()
        self._check(_lib().msm_amd_bucket_wise_accumulation(self.h, _u32buf(flat) if flat else None,
                                                            len(sorted_pairs), _u32buf(points_be32), n_points,
                                                            total_buckets, out))
        return [list(out[24 * i:24 * i + 24]) for i in range(total_buckets)]

    def sum_reduction(self, buckets_be32, buckets_size, num_windows):
        out = (c_uint32 * (num_windows * 24))()
        self._check(_lib().msm_amd_sum_reduction(self.h, _u32buf(buckets_be32), buckets_size, num_windows, out))
        return [list(out[24 * i:24 * i + 24]) for i in range(num_windows)]

    def test_op(self, op, a, b, count):
        per = 24 if op_is_point(op) else 8
        out = (c_uint32 * (count * per))()
        self._check(_lib().msm_amd_test_op(self.h, op, _u32buf(a), _u32buf(b), out, count))
        return list(out)


def msm_batch_multi(configs, scalars_list, points_list, ns, scalar_layout=SCALAR_MONT_LE,
                    point_layout=POINT_H2C_AFFINE, device=False):
    """The instance loop sharded over several configs (gpu_profiler.rs:101-106): instance j -> configs[j mod G], one
    host thread per config inside the library.  device=True: lists of device pointers (instance j on config j mod G's
    GPU)."""
    k, g = len(ns), len(configs)
    cc = (c_void_p * g)(*[c.h for c in configs])
    if device:
        sp = (c_void_p * k)(*scalars_list)
        pp = (c_void_p * k)(*points_list)
    else:
        sp = (c_void_p * k)(*[ctypes.cast(ctypes.c_char_p(s), c_void_p) for s in scalars_list])
        pp = (c_void_p * k)(*[ctypes.cast(ctypes.c_char_p(p), c_void_p) for p in points_list])
    nn = (c_size_t * k)(*ns)
    out = ctypes.create_string_buffer(96 * k)
    fn = _lib().msm_amd_msm_batch_multi_device if device else _lib().msm_amd_msm_batch_multi
    st = fn(cc, g, scalar_layout, point_layout, k, sp, pp, nn, out)
    if st != OK:
        detail = "; ".join(_lib().msm_amd_last_error(c.h).decode() for c in configs)
        raise MsmError(st, detail)
    return [out.raw[96 * i:96 * i + 96] for i in range(k)]


def submit_batch_multi_device(configs, d_scalars, d_points, ns, scalar_layout=SCALAR_MONT_LE,
                              point_layout=POINT_H2C_AFFINE):
    """Pipelined form of msm_batch_multi(..., device=True): enqueue every config's share, return a handle for
    wait_batch_multi (msm_amd_submit_batch_multi_device)."""
    k, g = len(ns), len(configs)
    cc = (c_void_p * g)(*[c.h for c in configs])
    sp = (c_void_p * k)(*d_scalars)
    pp = (c_void_p * k)(*d_points)
    nn = (c_size_t * k)(*ns)
    out = ctypes.create_string_buffer(96 * k)
    ticket = c_void_p()
    st = _lib().msm_amd_submit_batch_multi_device(cc, g, scalar_layout, point_layout, k, sp, pp, nn, out,
                                                  ctypes.byref(ticket))
    if st != OK:
        raise MsmError(st, "; ".join(_lib().msm_amd_last_error(c.h).decode() for c in configs))
    return (ticket, out, k, configs)


def wait_batch_multi(handle):
    ticket, out, k, configs = handle
    st = _lib().msm_amd_wait_batch_multi(ticket)
    if st != OK:
        raise MsmError(st, "; ".join(_lib().msm_amd_last_error(c.h).decode() for c in configs))
    return [out.raw[96 * i:96 * i + 96] for i in range(k)]


def msm_range_multi(configs, scalars: bytes, points: bytes, n: int, scalar_layout=SCALAR_MONT_LE,
                    point_layout=POINT_H2C_AFFINE) -> bytes:
    """ONE instance split by point range over several configs (msm_amd_msm_range_multi): config g uploads and runs the
    points of msm_amd_shard_range(n, G, g), the partial results are added."""
    g = len(configs)
    cc = (c_void_p * g)(*[c.h for c in configs])
    out = ctypes.create_string_buffer(96)
    st = _lib().msm_amd_msm_range_multi(cc, g, scalar_layout, point_layout, scalars, points, n, out)
    if st != OK:
        raise MsmError(st, "; ".join(_lib().msm_amd_last_error(c.h).decode() for c in configs))
    return out.raw


def shard_range(n: int, n_ctx: int, k: int):
    b, e = c_size_t(0), c_size_t(0)
    _lib().msm_amd_shard_range(n, n_ctx, k, ctypes.byref(b), ctypes.byref(e))
    return b.value, e.value


def shard_owner(instance: int, n_ctx: int) -> int:
    return _lib().msm_amd_shard_owner(instance, n_ctx)


def shard_count(n_inst: int, n_ctx: int, k: int) -> int:
    return _lib().msm_amd_shard_count(n_inst, n_ctx, k)


class RcclGather:
    """RCCL all-gather of per-rank result blocks from C++ (msm_amd_gather_*): one communicator per listed device."""

    def __init__(self, devices):
        self.g = c_void_p()
        arr = (c_int * len(devices))(*devices)
        st = _lib().msm_amd_gather_init(arr, len(devices), ctypes.byref(self.g))
        if st != OK:
            raise MsmError(st, "msm_amd_gather_init")
        self.n = len(devices)

    def all_gather(self, blocks):
        per = len(blocks[0])
        send = (c_void_p * self.n)(*[ctypes.cast(ctypes.c_char_p(b), c_void_p) for b in blocks])
        outs = [ctypes.create_string_buffer(per * self.n) for _ in range(self.n)]
        recv = (c_void_p * self.n)(*[ctypes.cast(o, c_void_p) for o in outs])
        st = _lib().msm_amd_gather_all(self.g, send, per, recv)
        if st != OK:
            raise MsmError(st, _lib().msm_amd_gather_last_error(self.g).decode())
        return [o.raw for o in outs]

    def close(self):
        if self.g:
            _lib().msm_amd_gather_destroy(self.g)
        self.g = None


def host_msm(scalars: bytes, points: bytes, n: int, threads=0, scalar_layout=SCALAR_MONT_LE,
             point_layout=POINT_H2C_AFFINE) -> bytes:
    """The product's CPU MSM (where the reference calls halo2curves::msm::msm_best: gpu_profiler.rs:157-159,
    msm.rs:412) -- host code of the library, no GPU and no ctx needed."""
    out = ctypes.create_string_buffer(96)
    st = _lib().msm_amd_host_msm(scalar_layout, point_layout, scalars, points, n, threads, out)
    if st != OK:
        raise MsmError(st)
    return out.raw


def generate_instance_host(seed, n, scalars_mont=True, threads=0):
    """(points, scalars) of the deterministic synthetic instance, generated on the host by the library."""
    pts = ctypes.create_string_buffer(64 * n)
    sc = ctypes.create_string_buffer(32 * n)
    st = _lib().msm_amd_generate_instance_host(seed, n, 1 if scalars_mont else 0, pts, sc, threads)
    if st != OK:
        raise MsmError(st)
    return pts.raw, sc.raw


def test_op_host(op, a, b, count):
    """Same single-op bodies as MsmConfig.test_op, executed on the host CPU by the library (no GPU)."""
    per = 24 if op_is_point(op) else 8
    out = (c_uint32 * (count * per))()
    st = _lib().msm_amd_test_op_host(op, _u32buf(a), _u32buf(b), out, count)
    if st != OK:
        raise MsmError(st)
    return list(out)


def sum_points(results) -> bytes:
    """Host sum of 96-byte results (the final addition of gpu_with_cpu, msm.rs:418-419)."""
    blob = b"".join(results)
    if len(blob) != 96 * len(results):
        raise ValueError("every result must be 96 bytes")
    out = ctypes.create_string_buffer(96)
    st = _lib().msm_amd_sum_points(blob, len(results), out)
    if st != OK:
        raise MsmError(st)
    return out.raw


def final_accumulation(res_be32, num_windows, window_size):
    """final_accumulation (final_accumulation.rs:5-40) -- host code in the library, no GPU needed."""
    out = (c_uint32 * 24)()
    st = _lib().msm_amd_final_accumulation(_u32buf(res_be32), num_windows, window_size, out)
    if st != OK:
        raise MsmError(st)
    return list(out)


# ---- names of the reference API (src/metal/msm.rs) ----------------------------------------------------
def setup_metal_state(device=-1) -> MsmConfig:
    """setup_metal_state (msm.rs:77-94)."""
    return MsmConfig(device)


def setup_metal_state_reusable() -> MsmConfig:
    """setup_metal_state_reusable (msm.rs:96-109): process-global cached config."""
    h = c_void_p()
    st = _lib().msm_amd_init_reusable(ctypes.byref(h))
    if st != OK:
        raise MsmError(st)
    return MsmConfig(_handle=h, _owned=False)


def get_global_metal_config() -> MsmConfig:
    """get_global_metal_config (msm.rs:114-119)."""
    h = c_void_p()
    st = _lib().msm_amd_get_global(ctypes.byref(h))
    if st != OK:
        raise MsmError(st, "MetalMsmConfig must be initialized before use.")
    return MsmConfig(_handle=h, _owned=False)


def gpu_msm_h2c(scalars: bytes, points: bytes, config: MsmConfig | None = None) -> bytes:
    """gpu_msm_h2c (msm.rs:352-364): bn256::Fr scalars (32 B each) x bn256::G1Affine points (64 B each)."""
    n = min(len(scalars) // 32, len(points) // 64)
    cfg = config or setup_metal_state_reusable()
    out = ctypes.create_string_buffer(96)
    cfg._check(_lib().msm_amd_gpu_msm_h2c(cfg.h, scalars, points, n, out))
    return out.raw


def gpu_msm_h2c_sync(scalars: bytes, points: bytes, after_sort, config: MsmConfig | None = None) -> bytes:
    """gpu_msm_h2c_sync (msm.rs:237-349): `after_sort()` is called once the GPU has finished sorting this MSM's
    bucket indices -- where the reference notifies its (Mutex<bool>, Condvar) pair (msm.rs:306-312)."""
    n = min(len(scalars) // 32, len(points) // 64)
    cfg = config or setup_metal_state_reusable()
    out = ctypes.create_string_buffer(96)
    cb = AFTER_SORT_FN((lambda _user: after_sort()) if after_sort is not None else 0)
    cfg._check(_lib().msm_amd_gpu_msm_h2c_sync(cfg.h, scalars, points, n, cb, None, out))
    return out.raw


def msm_best(scalars: bytes, points: bytes, config: MsmConfig | None = None) -> bytes:
    """msm_best (msm.rs:424-445): zero-scalar filtering + MSM."""
    n = min(len(scalars) // 32, len(points) // 64)
    cfg = config or setup_metal_state_reusable()
    out = ctypes.create_string_buffer(96)
    cfg._check(_lib().msm_amd_msm_best(cfg.h, scalars, points, n, out))
    return out.raw


def gpu_with_cpu(scalars: bytes, points: bytes, config: MsmConfig | None = None, split_at=None, cpu_threads=0) -> bytes:
    """gpu_with_cpu (msm.rs:366-421); split_at defaults to the reference's policy."""
    n = min(len(scalars) // 32, len(points) // 64)
    cfg = config or setup_metal_state_reusable()
    if split_at is None:
        split_at = _lib().msm_amd_reference_split(n)
    out = ctypes.create_string_buffer(96)
    cfg._check(_lib().msm_amd_gpu_with_cpu(cfg.h, scalars, points, n, split_at, cpu_threads, out))
    return out.raw


def metal_msm(points: bytes, scalars: bytes, config: MsmConfig) -> bytes:
    """metal_msm (msm.rs:220-234): ark G1Projective points (96 B each) x ark Fr scalars."""
    n = min(len(scalars) // 32, len(points) // 96)
    out = ctypes.create_string_buffer(96)
    config._check(_lib().msm_amd_metal_msm_ark(config.h, points, scalars, n, out))
    return out.raw
