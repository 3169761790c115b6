"""N > 1 path on CPU: two gloo ranks shard instances and all-gather the 96-byte results
(the GPU compute is replaced by the CPU oracle here; the sharding/gather code is the product's)."""
import importlib
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_rank, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
    from oracle import c_oracle as co
    local = []
    for g in mg.instance_ids(rank, world, per_rank):
        pts, sc = co.gen_instance(mg.instance_seed(g), n, True, threads=1)
        local.append(co.msm_best(sc, pts, n, 1))
    allr = mg.all_gather_results(local, dist)
    g = mg.ResultGatherer(dist, None, per_rank)        # the preallocated form bench.py uses
    g.gather(local)
    assert g.fetch() == allr
    q.put((rank, allr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, per_rank, n = 2, 3, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
    from oracle import c_oracle as co
    expect = []
    for g in range(world * per_rank):
        pts, sc = co.gen_instance(mg.instance_seed(g), n, True, threads=1)
        expect.append(co.msm_best(sc, pts, n, 1))
    assert got[0] == expect and got[1] == expect          # every rank sees all results, in global order
    # each instance is owned by exactly one rank
    owners = [mg.instance_ids(r, world, per_rank) for r in range(world)]
    assert sorted(sum(owners, [])) == list(range(world * per_rank))


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
    res = [bytes([i]) * 96 for i in range(4)]
    assert mg.all_gather_results(res, None) == res
    with pytest.raises(ValueError):
        mg.all_gather_results([b"short"], None)
    with pytest.raises(ValueError):
        mg.instance_ids(2, 2, 1)


def _range_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("metal-msm-gpu-acceleration_amd")
    mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
    from oracle import c_oracle as co
    pts, sc = co.gen_instance(mg.instance_seed(900), n, True, threads=1)
    # the per-range MSM is the oracle's here (no GPU); split, gather and the final addition are the product's
    local = lambda b, e: co.msm_best(sc[32 * b:32 * e], pts[64 * b:64 * e], e - b, 1)
    q.put((rank, mg.sharded_msm(local, pkg.sum_points, rank, world, n, dist)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_point_range_split_of_one_instance():
    """SURVEY §8e "single huge instance": rank g owns points [g n/G, (g+1) n/G); the G partial results are
    all-gathered and added by msm_amd_sum_points on every rank."""
    world, n = 2, 301
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
    from oracle import bn254_ref as o
    from oracle import c_oracle as co
    pts, sc = co.gen_instance(mg.instance_seed(900), n, True, threads=1)
    whole = co.msm_best(sc, pts, n, 1)
    assert got[0] == got[1]
    assert o.decode_jacobian_mont_le(got[0]) == o.decode_jacobian_mont_le(whole)
    z = int.from_bytes(got[0][64:96], "little")
    assert z == o.MONT_R % o.P                                     # normalised like every other result
    # ranges tile [0, n) for any world size, including more ranks than points
    for w, m in ((1, 5), (3, 10), (8, 5), (4, 0)):
        r = [mg.point_range(k, w, m) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == m and all(r[k][1] == r[k + 1][0] for k in range(w - 1))


def test_sum_points_host():
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("metal-msm-gpu-acceleration_amd")
    from oracle import bn254_ref as o
    import random
    rng = random.Random(3)
    pts = [o.scalar_mul(rng.randrange(1, o.R_ORDER), o.GEN) for _ in range(4)]
    enc = [o.encode_projective_ark(o.to_jac(p)) for p in pts] + [bytes(96), o.encode_projective_ark(None)]
    acc = None
    for p in pts:
        acc = o.jac_add(acc, o.to_jac(p))
    assert o.decode_jacobian_mont_le(pkg.sum_points(enc)) == o.to_affine(acc)
    assert o.decode_jacobian_mont_le(pkg.sum_points([])) is None
    assert o.decode_jacobian_mont_le(pkg.sum_points([enc[0], o.encode_projective_ark(o.to_jac(o.aff_neg(pts[0])))])) is None
    assert o.decode_jacobian_mont_le(pkg.sum_points([enc[1], enc[1]])) == o.to_affine(o.jac_double(o.to_jac(pts[1])))
