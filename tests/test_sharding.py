"""Sharding arithmetic of msm_amd_msm_batch_multi (instance j -> ctx j mod G; the loop of
src/bin/gpu_profiler.rs:101-106 spread over contexts) -- host logic, no GPU."""
import ctypes

import pytest


@pytest.mark.parametrize("n_inst,n_ctx", [(1, 1), (5, 1), (5, 2), (40, 8), (7, 8), (8, 8), (41, 8), (3, 5)])
def test_shard_owner_and_counts_partition_the_instances(msm_pkg, n_inst, n_ctx):
    owners = [msm_pkg.shard_owner(j, n_ctx) for j in range(n_inst)]
    assert owners == [j % n_ctx for j in range(n_inst)]
    counts = [msm_pkg.shard_count(n_inst, n_ctx, k) for k in range(n_ctx)]
    assert counts == [owners.count(k) for k in range(n_ctx)]
    assert sum(counts) == n_inst
    assert max(counts) - min(counts) <= 1                       # balanced: config 4 = 40 instances -> 5 per GPU at 8
    assert msm_pkg.shard_count(n_inst, n_ctx, n_ctx) == 0       # out of range
    # the i-th instance of ctx k is global instance k + i * G (what the gather relies on)
    for k in range(n_ctx):
        mine = [j for j in range(n_inst) if owners[j] == k]
        assert mine == [k + i * n_ctx for i in range(counts[k])]


def test_config4_shape(msm_pkg):
    assert [msm_pkg.shard_count(40, 8, k) for k in range(8)] == [5] * 8
    assert [msm_pkg.shard_count(40, g, 0) for g in (1, 2, 4, 8)] == [40, 20, 10, 5]


def test_batch_multi_rejects_bad_arguments_without_a_gpu(msm_pkg):
    L = msm_pkg.lib()
    out = ctypes.create_string_buffer(96)
    one = (ctypes.c_void_p * 1)(None)
    n = (ctypes.c_size_t * 1)(4)
    assert L.msm_amd_msm_batch_multi(None, 1, 0, 0, 1, one, one, n, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_msm_batch_multi(one, 1, 0, 0, 1, one, one, n, out) == msm_pkg.INPUT_ERROR     # null ctx in the list
    assert L.msm_amd_msm_batch_multi(one, 0, 0, 0, 1, one, one, n, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_gather_init(None, 1, ctypes.byref(ctypes.c_void_p())) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_ctx_device(None) == -1
    assert L.msm_amd_pin_thread_to_device(0) in (0, 1)          # never an error, GPU or not


@pytest.mark.parametrize("n,n_ctx", [(1, 1), (10, 3), (1 << 24, 8), (7, 8), (100003, 2), (8, 8), (0, 4)])
def test_point_ranges_partition_one_instance(msm_pkg, n, n_ctx):
    """msm_amd_shard_range (the split of msm_amd_msm_range_multi) = multi_gpu.point_range, the Python launcher's."""
    import importlib
    mg = importlib.import_module(msm_pkg.__name__ + ".multi_gpu")
    ranges = [msm_pkg.shard_range(n, n_ctx, k) for k in range(n_ctx)]
    assert ranges == [mg.point_range(k, n_ctx, n) for k in range(n_ctx)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    assert all(ranges[k][1] == ranges[k + 1][0] for k in range(n_ctx - 1))
    sizes = [e - b for b, e in ranges]
    assert max(sizes) - min(sizes) <= 1
    assert msm_pkg.shard_range(n, n_ctx, n_ctx) == (0, 0)        # out of range


def test_layout_sizes_and_range_multi_argument_checks(msm_pkg):
    L = msm_pkg.lib()
    assert [L.msm_amd_scalar_bytes(k) for k in (msm_pkg.SCALAR_MONT_LE, msm_pkg.SCALAR_CANON_LE, msm_pkg.SCALAR_CANON_BE32, 9)] == [32, 32, 32, 0]
    assert [L.msm_amd_point_bytes(k) for k in (msm_pkg.POINT_H2C_AFFINE, msm_pkg.POINT_ARK_PROJECTIVE, msm_pkg.POINT_ARK_AFFINE,
                                               msm_pkg.POINT_JAC_BE32, 99)] == [64, 96, 72, 96, 0]
    out = ctypes.create_string_buffer(96)
    buf = ctypes.create_string_buffer(256)
    one = (ctypes.c_void_p * 1)(None)
    assert L.msm_amd_msm_range_multi(None, 1, 0, 0, buf, buf, 2, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_msm_range_multi(one, 1, 0, 0, buf, buf, 0, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_msm_range_multi(one, 1, 0, msm_pkg.POINT_PREPARED, buf, buf, 2, out) == msm_pkg.INPUT_ERROR
    assert L.msm_amd_msm_range_multi(one, 1, 0, 0, buf, buf, 2, out) == msm_pkg.INPUT_ERROR        # null ctx in the list
