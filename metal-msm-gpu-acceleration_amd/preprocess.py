"""Instance-file harness: host-side mirror of src/utils/preprocess.rs of the reference.

Same names, same file format, same file naming (`msm_{log}x{n}.bin` under
`~/.msm_gpu_acceleration/msm_vecs`), so that cache files interchange with the Rust crate.  The byte
work (bincode 1.3 framing, limb reordering, Montgomery <-> canonical scalars) is done by the C ABI
(`msm_amd_instances_*`, `msm_amd_to_wire`, csrc/instance_file.hip); new instances are generated on
the GPU by the library's deterministic generator.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_size_t, c_void_p

from . import (FILE_OPEN_ERROR, DESERIALIZATION_ERROR, INVALID_DATA, OK, POINT_H2C_AFFINE, POINT_JAC_BE32,
               SCALAR_CANON_BE32, SCALAR_CANON_LE, MsmConfig, _lib)


class HarnessError(RuntimeError):
    """HarnessError (preprocess.rs:11-21)."""


class FileOpenError(HarnessError, OSError):
    pass


class DeserializationError(HarnessError):
    pass


class InvalidData(HarnessError):
    pass


_ERRORS = {FILE_OPEN_ERROR: FileOpenError, DESERIALIZATION_ERROR: DeserializationError, INVALID_DATA: InvalidData}


def _check(st, what):
    if st != OK:
        raise _ERRORS.get(st, HarnessError)(f"{what}: {_lib().msm_amd_strerror(st).decode()}")


class MsmInstance:
    """MsmInstance (preprocess.rs:24-28) in wire form: `points` = n x 96 B (24 x u32 per point, x|y|z most
    significant limb first, Montgomery), `scalars` = n x 32 B (8 x u32, most significant limb first, canonical).
    Feed them to `MsmConfig.msm(..., scalar_layout=SCALAR_CANON_BE32, point_layout=POINT_JAC_BE32)`."""

    def __init__(self, points: bytes, scalars: bytes):
        if len(points) % 96 or len(scalars) % 32 or len(points) // 96 != len(scalars) // 32:
            raise InvalidData("points and scalars differ in length")   # the assert of preprocess.rs:78
        self.points = points
        self.scalars = scalars

    def __len__(self):
        return len(self.scalars) // 32


def to_wire(scalars: bytes, points: bytes, scalar_layout, point_layout, point_bytes) -> MsmInstance:
    """ToLimbs (limbs_conversion.rs:87-137, 282-327) for a whole instance."""
    n = len(scalars) // 32
    if len(points) // point_bytes != n:
        raise InvalidData("points and scalars differ in length")
    so = ctypes.create_string_buffer(32 * n)
    po = ctypes.create_string_buffer(96 * n)
    _check(_lib().msm_amd_to_wire(scalar_layout, point_layout, scalars, points, n, so, po), "to_wire")
    return MsmInstance(po.raw, so.raw)


def save_msm_instances(path, data):
    """save_msm_instances (preprocess.rs:84-97)."""
    k = len(data)
    pts = (c_void_p * k)(*[ctypes.cast(ctypes.c_char_p(d.points), c_void_p) for d in data])
    scs = (c_void_p * k)(*[ctypes.cast(ctypes.c_char_p(d.scalars), c_void_p) for d in data])
    ns = (c_size_t * k)(*[len(d) for d in data])
    _check(_lib().msm_amd_instances_save(os.fspath(path).encode(), k, ns, pts, scs), f"save {path}")


def load_msm_instances(path):
    """load_msm_instances (preprocess.rs:99-111)."""
    h = c_void_p()
    _check(_lib().msm_amd_instances_open(os.fspath(path).encode(), ctypes.byref(h)), f"open {path}")
    try:
        out = []
        for j in range(_lib().msm_amd_instances_count(h)):
            n = _lib().msm_amd_instances_size(h, j)
            po = ctypes.create_string_buffer(96 * n)
            so = ctypes.create_string_buffer(32 * n)
            _check(_lib().msm_amd_instances_read(h, j, po, so), f"read {path}")
            out.append(MsmInstance(po.raw, so.raw))
        return out
    finally:
        _lib().msm_amd_instances_close(h)


def default_msm_vec_repo():
    """default_msm_vec_repo (preprocess.rs:204-212)."""
    return os.path.dirname(instance_path(0, 0, None))


def instance_path(log_instance_size, num_instances, dir=None):
    buf = ctypes.create_string_buffer(4096)
    m = _lib().msm_amd_instances_default_path(None if dir is None else os.fspath(dir).encode(), log_instance_size,
                                              num_instances, buf, len(buf))
    if m == 0:
        raise InvalidData("path too long")
    return buf.value.decode()


def generate_msm_instances(instance_size, num_instances, seed, config: MsmConfig):
    """generate_msm_instances (preprocess.rs:113-138): uniform random points and scalars.  The reference
    draws from the caller's RNG on the CPU; here instance j comes from the library's device generator with
    seed + j (the same streams bench.py and gpu_profiler use)."""
    out = []
    for j in range(num_instances):
        d_pts, d_sc = config.generate_instance(seed + j, instance_size, scalars_mont=False)
        try:
            out.append(to_wire(config.to_host(d_sc, 32 * instance_size), config.to_host(d_pts, 64 * instance_size),
                               SCALAR_CANON_LE, POINT_H2C_AFFINE, 64))
        finally:
            config.free(d_pts)
            config.free(d_sc)
    return out


def get_or_create_msm_instances(log_instance_size, num_instances, seed, dir=None, config: MsmConfig | None = None):
    """get_or_create_msm_instances (preprocess.rs:143-202): load `msm_{log}x{n}.bin` when it exists and matches
    (instance count, first instance's size), else generate, save and return."""
    path = instance_path(log_instance_size, num_instances, dir)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if os.path.exists(path):
        msm_list = load_msm_instances(path)
        if msm_list and len(msm_list) == num_instances and len(msm_list[0]) == 1 << log_instance_size:
            return msm_list
        raise InvalidData(f"File mismatch: has instance_size={len(msm_list[0]) if msm_list else 0} and "
                          f"num_instances={len(msm_list)}, need {log_instance_size} & {num_instances}")
    if config is None:
        from . import setup_metal_state_reusable
        config = setup_metal_state_reusable()
    msm_list = generate_msm_instances(1 << log_instance_size, num_instances, seed, config)
    save_msm_instances(path, msm_list)
    return msm_list


def run_instance(config: MsmConfig, inst: MsmInstance) -> bytes:
    """One MSM of a wire-form instance on the GPU (what gpu_profiler does with a loaded instance)."""
    return config.msm(inst.scalars, inst.points, len(inst), scalar_layout=SCALAR_CANON_BE32,
                      point_layout=POINT_JAC_BE32)
