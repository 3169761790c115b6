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
