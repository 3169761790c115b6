"""Several contexts below the C ABI: msm_amd_msm_batch_multi shards the reference's instance
loop (gpu_profiler.rs:101-106) as instance j -> ctx j mod G with one host thread per ctx.  One GPU is available to
the tests, so the contexts share it: that still proves the library has no hidden globals between contexts and that
two of them run concurrently from two threads.  The RCCL gather runs with the ranks this box has (one)."""
import json
import os
import subprocess
import threading

import pytest

from oracle import bn254_ref as o
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "gpu_profiler")


def _same(a, b):
    return o.decode_jacobian_mont_le(a) == o.decode_jacobian_mont_le(b)


def test_two_contexts_on_one_device_through_batch_multi(cfg, msm_pkg):
    sizes = [1 << 12, 3000, 1 << 14, 17, 1 << 13]          # 5 instances over 2 contexts: 3 + 2
    inst = [co.gen_instance(o.SEED_BASE + 40 + j, n) for j, n in enumerate(sizes)]
    want = [co.msm_best(sc, pts, n) for (pts, sc), n in zip(inst, sizes)]
    second = msm_pkg.setup_metal_state(cfg.device())
    try:
        got = msm_pkg.msm_batch_multi([cfg, second], [sc for _p, sc in inst], [p for p, _s in inst], sizes)
        assert all(_same(g, w) for g, w in zip(got, want))
        assert got == cfg.msm_batch([sc for _p, sc in inst], [p for p, _s in inst], sizes)   # same bytes as one ctx
        # device-resident variant: instance j's buffers allocated through ITS ctx
        ctxs = [cfg, second]
        dps, dss = [], []
        for j, ((pts, sc), n) in enumerate(zip(inst, sizes)):
            c = ctxs[msm_pkg.shard_owner(j, 2)]
            dp, ds = c.alloc(64 * n), c.alloc(32 * n)
            c.to_device(dp, pts)
            c.to_device(ds, sc)
            dps.append(dp)
            dss.append(ds)
        try:
            got_dev = msm_pkg.msm_batch_multi(ctxs, dss, dps, sizes, device=True)
            assert got_dev == got
        finally:
            for j, (dp, ds) in enumerate(zip(dps, dss)):
                ctxs[msm_pkg.shard_owner(j, 2)].free(dp)
                ctxs[msm_pkg.shard_owner(j, 2)].free(ds)
        # more contexts than instances, and a ctx listed twice
        assert msm_pkg.msm_batch_multi([cfg, second], [inst[0][1]], [inst[0][0]], [sizes[0]]) == [got[0]]
        with pytest.raises(msm_pkg.MsmError):
            msm_pkg.msm_batch_multi([cfg, cfg], [sc for _p, sc in inst], [p for p, _s in inst], sizes)
    finally:
        second.close()


def test_one_instance_split_by_point_range_over_two_contexts(cfg, msm_pkg):
    """msm_amd_msm_range_multi: the 'single huge instance' row of SURVEY.md section 8e below the C ABI."""
    second = msm_pkg.setup_metal_state(cfg.device())
    try:
        for n, layout in ((100003, msm_pkg.POINT_H2C_AFFINE), (1, msm_pkg.POINT_H2C_AFFINE), (1 << 16, msm_pkg.POINT_ARK_AFFINE)):
            pts, sc = co.gen_instance(o.SEED_BASE + 70 + (n & 7), n)
            want = co.msm_best(sc, pts, n)
            if layout == msm_pkg.POINT_ARK_AFFINE:       # 72-byte records: x, y, infinity flag + padding
                import numpy as np
                rec = np.zeros((n, 72), dtype=np.uint8)
                rec[:, :64] = np.frombuffer(pts, dtype=np.uint8).reshape(n, 64)
                pts = rec.tobytes()
            got = msm_pkg.msm_range_multi([cfg, second], sc, pts, n, point_layout=layout)
            assert _same(got, want)
            assert _same(msm_pkg.msm_range_multi([cfg], sc, pts, n, point_layout=layout), want)
    finally:
        second.close()


def test_two_contexts_driven_from_two_python_threads(cfg, msm_pkg):
    """The header's promise: calls on one ctx are serialised, different ctxs run concurrently."""
    n = 1 << 15
    inst = [co.gen_instance(o.SEED_BASE + 60 + j, n) for j in range(2)]
    want = [co.msm_best(sc, pts, n) for pts, sc in inst]
    second = msm_pkg.setup_metal_state(cfg.device())
    results = [[None] * 6, [None] * 6]
    errors = []

    def drive(k, c):
        try:
            pts, sc = inst[k]
            for r in range(6):
                results[k][r] = c.msm(sc, pts, n)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    try:
        ts = [threading.Thread(target=drive, args=(k, c)) for k, c in enumerate((cfg, second))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        for k in range(2):
            assert all(_same(r, want[k]) for r in results[k])
    finally:
        second.close()


def test_rccl_gather_from_cxx_with_the_ranks_this_box_has(cfg, msm_pkg):
    """ncclCommInitAll + ncclAllGather through msm_amd_gather_*: one rank here (world size 1 still goes through
    RCCL); the N-rank run is the driver's, on an 8-GPU node."""
    g = msm_pkg.RcclGather([cfg.device()])
    try:
        block = bytes(range(96)) * 5                      # 5 results of one rank (config 3: 5 instances per GPU)
        assert g.all_gather([block]) == [block]
        bigger = bytes(reversed(range(96))) * 7           # the device buffers grow
        assert g.all_gather([bigger]) == [bigger]
    finally:
        g.close()
    with pytest.raises(msm_pkg.MsmError):                 # RCCL wants one rank per GPU
        msm_pkg.RcclGather([cfg.device(), cfg.device()])


def test_pipelined_multi_context_form_equals_the_blocking_one(cfg, msm_pkg):
    """msm_amd_submit_batch_multi_device / msm_amd_wait_batch_multi (two batches in flight over two contexts of one
    device) return the bytes of msm_amd_msm_batch_multi_device; the benchmark loop this keeps fed is
    benches/msm_benchmark.rs:29-34."""
    sizes = [1 << 13, 5000, 1 << 12, 33, 1 << 14, 9000, 77]
    inst = [co.gen_instance(o.SEED_BASE + 80 + j, n) for j, n in enumerate(sizes)]
    second = msm_pkg.setup_metal_state(cfg.device())
    ctxs = [cfg, second]
    dps, dss = [], []
    try:
        for j, ((pts, sc), n) in enumerate(zip(inst, sizes)):
            c = ctxs[msm_pkg.shard_owner(j, 2)]
            dp, ds = c.alloc(64 * n), c.alloc(32 * n)
            c.to_device(dp, pts)
            c.to_device(ds, sc)
            dps.append(dp)
            dss.append(ds)
        blocking = msm_pkg.msm_batch_multi(ctxs, dss, dps, sizes, device=True)
        assert all(_same(g, co.msm_best(sc, pts, n)) for g, ((pts, sc), n) in zip(blocking, zip(inst, sizes)))
        h0 = msm_pkg.submit_batch_multi_device(ctxs, dss, dps, sizes)
        h1 = msm_pkg.submit_batch_multi_device(ctxs, dss[:3], dps[:3], sizes[:3])     # both in flight before any wait
        assert msm_pkg.wait_batch_multi(h0) == blocking
        assert msm_pkg.wait_batch_multi(h1) == blocking[:3]
        one = msm_pkg.submit_batch_multi_device(ctxs, dss[:1], dps[:1], sizes[:1])   # fewer instances than contexts
        assert msm_pkg.wait_batch_multi(one) == blocking[:1]
        with pytest.raises(msm_pkg.MsmError):
            msm_pkg.submit_batch_multi_device([cfg, cfg], dss, dps, sizes)
    finally:
        for j, (dp, ds) in enumerate(zip(dps, dss)):
            ctxs[msm_pkg.shard_owner(j, 2)].free(dp)
            ctxs[msm_pkg.shard_owner(j, 2)].free(ds)
        second.close()


def test_config4_workload_40_instances_of_2p20_over_eight_contexts(cfg, msm_pkg):
    """BASELINE config 4's WORKLOAD through the G = 8 code on the one GPU this box has: 40 instances x 2^20 points
    (one set of dlog-structured bases P_i = (a0 + i d) G shared by all, 40 scalar sets) sharded as instance j ->
    context j mod 8 over EIGHT contexts of one device -- blocking (msm_amd_msm_batch_multi_device) and pipelined (two
    batches of 40 in flight) -- then the RCCL gather of the 5 x 96-byte blocks with the ranks the box has.  Every one
    of the 40 results is checked against (sum k_i (a0 + i d)) G from big integers (tests/helpers.dlog_expected).
    What this does NOT measure is scaling: eight contexts share one GPU.  The loop being sharded:
    gpu_profiler.rs:101-106, benches/msm_benchmark.rs:29-34."""
    from helpers import dlog_expected
    import random
    n, inst, G = 1 << 20, 40, 8
    rng = random.Random(404)
    a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
    ctxs = [cfg] + [msm_pkg.setup_metal_state(cfg.device()) for _ in range(G - 1)]
    d_pts, d_sc, want = None, [], []
    try:
        for j in range(inst):
            dp, ds = cfg.generate_instance(o.SEED_BASE + 4000 + j, n, True)   # scalars from the device generator
            sb = cfg.to_host(ds, 32 * n)
            if j == 0:
                pb, _ = co.dlog_instance(a0, d, sb, n)
                cfg.to_device(dp, pb)
                d_pts = dp
            else:
                cfg.free(dp)
            d_sc.append(ds)
            want.append(dlog_expected(a0, d, sb, n))
        ns = [n] * inst
        got = msm_pkg.msm_batch_multi(ctxs, d_sc, [d_pts] * inst, ns, device=True)
        for j in range(inst):
            assert o.decode_jacobian_mont_le(got[j]) == want[j], j
        h0 = msm_pkg.submit_batch_multi_device(ctxs, d_sc, [d_pts] * inst, ns)
        h1 = msm_pkg.submit_batch_multi_device(ctxs, d_sc[::-1], [d_pts] * inst, ns)   # a second batch behind it
        assert msm_pkg.wait_batch_multi(h0) == got
        assert msm_pkg.wait_batch_multi(h1) == got[::-1]
        # the gather of config 4: every rank contributes its ceil(40 / G) x 96 B; this box has one rank
        g = msm_pkg.RcclGather([cfg.device()])
        try:
            mine = b"".join(got[j] for j in range(inst) if msm_pkg.shard_owner(j, G) == 0)
            assert len(mine) == 5 * 96 and g.all_gather([mine]) == [mine]
        finally:
            g.close()
    finally:
        for ds in d_sc:
            cfg.free(ds)
        if d_pts is not None:
            cfg.free(d_pts)
        for c in ctxs[1:]:
            c.close()


def _run(*args):
    r = subprocess.run([EXE, *args, "--json"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_gpu_profiler_gpus_flag_matches_single_context():
    """`gpu_profiler 16 6 gpu_resident 2 --gpus 1` (sharded path, RCCL gather of one rank) and two contexts on one
    device give the results of the plain single-context run."""
    single, _ = _run("16", "6", "gpu_resident", "2")
    sharded, err = _run("16", "6", "gpu_resident", "2", "--devices", "0")
    assert sharded["results_fnv1a64"] == single["results_fnv1a64"]
    assert sharded["rccl_gather"] is True and "every rank holds all 6 results: yes" in err
    shared, err2 = _run("16", "6", "gpu_resident", "2", "--devices", "0,0")
    assert shared["results_fnv1a64"] == single["results_fnv1a64"] and shared["gpus"] == 2
    assert "RCCL gather skipped" in err2
    host, _ = _run("14", "5", "gpu", "1", "--devices", "0,0,0")
    host1, _ = _run("14", "5", "gpu", "1")
    assert host["results_fnv1a64"] == host1["results_fnv1a64"]
    ranged, _ = _run("14", "5", "gpu", "1", "--devices", "0,0,0", "--range-split")     # every instance over 3 contexts
    assert ranged["results_fnv1a64"] == host1["results_fnv1a64"]


def test_gpu_profiler_config4_over_eight_contexts_of_one_device():
    """`gpu_profiler 20 40 gpu_resident 1 --devices 0,0,0,0,0,0,0,0`: config 4's argv through the sharded CLI path
    (eight contexts, pipelined submit / wait across retries) gives the results of the single-context run."""
    single, _ = _run("20", "40", "gpu_resident", "1")
    eight, err = _run("20", "40", "gpu_resident", "2", "--devices", "0,0,0,0,0,0,0,0")
    assert eight["results_fnv1a64"] == single["results_fnv1a64"] and eight["gpus"] == 8
    assert "RCCL gather skipped" in err
