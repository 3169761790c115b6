"""Headline benchmark: BN254 G1 MSM/s at log_size=20 (5 instances per GPU), see BASELINE.json.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path (metal::msm::gpu_msm_h2c pipeline) over one batch of 5 synthetic
2^20-point instances per GPU, inputs already resident in HBM (device generator).  Instances shard across
ranks with no data-path collective; the per-instance results (96 B each) are all-gathered over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-size", type=int, default=20)
    ap.add_argument("--instances", type=int, default=5, help="instances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window", type=int, default=0, help="force the window size (0 = the library's automatic choice)")
    ap.add_argument("--precomputed-tables", action="store_true",
                    help="NOT the headline: window tables 2^(c w) P precomputed once per set of bases (SURVEY §8f N4), "
                         "one bucket set, c = log2(n) - 1")
    ap.add_argument("--table-window", type=int, default=0, help="window bits of --precomputed-tables (0 = automatic)")
    ap.add_argument("--persistent-bases", action="store_true",
                    help="NOT the headline: bases converted once and kept resident (SURVEY §8f N4); the default "
                         "re-converts them inside every MSM like the reference does (msm.rs:152-153)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    m = importlib.import_module("metal-msm-gpu-acceleration_amd")
    mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
    cfg = m.setup_metal_state(local_rank)          # fails loudly without a gfx950 device
    if args.window:
        cfg.set_window_size(args.window)
    n = 1 << args.log_size
    inst = args.instances
    d_pts, d_sc = [], []
    for g in mg.instance_ids(rank, world, inst):   # rank r owns global instances r*inst .. r*inst+inst-1
        dp, ds = cfg.generate_instance(mg.instance_seed(g), n, True)
        d_pts.append(dp)
        d_sc.append(ds)
    ns = [n] * inst
    point_layout = m.POINT_H2C_AFFINE
    tables = []
    if args.precomputed_tables:
        tables = [cfg.tables_build_device(dp, n, window_size=args.table_window) for dp in d_pts]
        for dp in d_pts:
            cfg.free(dp)
        d_pts = tables
        point_layout = m.POINT_TABLES
    elif args.persistent_bases:
        raw, d_pts = d_pts, [cfg.bases_prepare_device(dp, n) for dp in d_pts]
        for dp in raw:
            cfg.free(dp)
        point_layout = m.POINT_PREPARED

    gatherer = mg.ResultGatherer(dist, dev, inst) if dist is not None else None

    def finish(handle):
        outs = cfg.wait_batch(handle)                    # host Horner pass of the batch
        if gatherer is not None:                         # RCCL gather of per-instance results over xGMI
            gatherer.gather(outs)                        # (enqueued; completes before the closing barrier)
        t = cfg.timings()                                # hipEvent times on the library's own streams
        return outs, t

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        cfg.synchronize()

    def run_steps(k, record):
        """k steps, software-pipelined: step i+1 is submitted before step i is waited for, so the GPU never
        idles between steps; every step's results are produced and returned inside the loop."""
        outs, pending = None, None
        for _ in range(k):
            h = cfg.submit_batch_device(d_sc, d_pts, ns, point_layout=point_layout)
            if pending is not None:
                outs, t = finish(pending)
                record(t)
            pending = h
        if pending is not None:
            outs, t = finish(pending)
            record(t)
        return outs

    acc_ms, acc_stage_ms, tot_ms, sort_ms, red_ms, fin_ms = [], [], [], [], [], []

    def record(t):
        acc_ms.append(t.accumulate_kernel_ms)
        acc_stage_ms.append(t.accumulate_ms)
        tot_ms.append(t.total_gpu_ms)
        sort_ms.append(t.sort_ms)
        red_ms.append(t.reduce_ms)
        fin_ms.append(t.final_ms)

    outs = run_steps(args.warmup, lambda t: None)
    barrier()
    t0 = time.perf_counter()
    outs = run_steps(args.steps, record)
    barrier()
    elapsed = time.perf_counter() - t0
    if gatherer is not None:
        allr = gatherer.fetch()                          # every rank holds every instance's result
        assert allr[rank * inst:(rank + 1) * inst] == outs, "gathered results differ from the local ones"
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tm = cfg.timings()
    window = tm.window_size

    # ---- roofline of the dominant kernel (bucket accumulation), per launch = one instance
    L = m.lib()
    # algorithmic bytes = SURVEY.md section 8(d)'s per-unit figure: one MSM under the REFERENCE's window policy (3 below
    # 32 points, else 15: msm.rs:137-141), whatever window this build picks for itself -- the job is the same
    ref_window = 15 if n >= 32 else 3
    a3 = L.msm_amd_algorithmic_bytes(n, ref_window, 1)
    acc_avg_ms = sum(acc_ms) / len(acc_ms)
    achieved = a3 / (acc_avg_ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(f"accumulate_log{args.log_size}_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "accumulate_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": a3, "avg_launch_ms": round(acc_avg_ms, 4),
                "whole_pipeline_GBps": round(L.msm_amd_algorithmic_bytes(n, ref_window, 0) /
                                             (sum(tot_ms) / len(tot_ms) * 1e-3) / 1e9, 2)}

    # ---- secondary roof (SURVEY 8d asks for it): the kernel is bound by quarter-rate 32-bit multiplies, not by HBM.
    # Instruction counts from csrc/bn254_fq29.hip.h: a 9-limb Montgomery product is 81 + 81 v_mad_u64_u32 + 9
    # v_mul_lo_u32 = 171 multiplier-pipe instructions, a squaring 135, two products with one reduction 252; a mixed
    # addition (7 mul + 2 sqr + 1 double product) 1719, the affine+affine start of a work item 1035.  Peak: 1.81 wave
    # instructions/ns/CU measured for v_mad_u64_u32 (profiles/r01_valu_rates_microbench.txt) x 256 CUs x 64 lanes.
    items = float(tm.reserved2[0])
    lane_madds = n * tm.num_windows - 2.0 * items          # first point of an item is free, second is the 1035 one
    mul_instr = lane_madds * 1719.0 + items * 1035.0
    valu = {"bound": "valu-int32-multiply", "achieved": round(mul_instr / (acc_avg_ms * 1e-3) / 1e12, 2),
            "peak": 29.65, "unit": "T lane-instr/s", "frac": round(mul_instr / (acc_avg_ms * 1e-3) / 29.65e12, 4),
            "work_items": int(items)}
    roofline["secondary"] = valu

    # ---- CPU baseline (rank 0, single-GPU run only): the oracle's restatement of halo2curves msm_best
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import bn254_ref as o
        from oracle import c_oracle as co
        cores = min(16, co.default_threads())
        h_pts = [cfg.to_host(d_pts[j], 64 * n) for j in range(inst)]
        h_sc = [cfg.to_host(d_sc[j], 32 * n) for j in range(inst)]
        t_cpu0 = time.perf_counter()
        done = 0
        cpu_outs = []
        while True:
            for j in range(inst):
                r = co.msm_best(h_sc[j], h_pts[j], n, cores)
                if done < inst:
                    cpu_outs.append(r)
                done += 1
            if time.perf_counter() - t_cpu0 > 10.0 or done >= 4 * inst:
                break
        t_cpu = time.perf_counter() - t_cpu0
        for j in range(inst):          # parity gate: bit-exact canonical affine result
            if o.decode_jacobian_mont_le(outs[j]) != o.decode_jacobian_mont_le(cpu_outs[j]):
                raise SystemExit(f"PARITY FAILURE: instance {j} GPU != CPU")
        cpu = {"value": round(done / t_cpu, 4), "unit": "MSM/s", "cores": cores, "kind": "port",
               "sample": f"{done} MSMs of 2^{args.log_size} points (the bench's own {inst} instances, "
                         f"{done // inst} pass(es)), oracle_msm_best = C restatement of halo2curves msm_best, "
                         f"{cores} threads", "bit_exact_vs_gpu": True}

    if rank == 0:
        total_msms = inst * world * args.steps
        line = {
            "metric": "BN254 G1 MSM/s at log_size=20 (5 instances)",
            "value": round(total_msms / elapsed, 3),
            "unit": "MSM/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x8 (256-bit Montgomery integer)",
            "data": "synthetic",
            "config": {"workload": f"log_size={args.log_size}, {inst} instances per GPU, h2c BN254 G1 "
                                   f"(gpu_msm_h2c pipeline, window {window})",
                       "instances_per_gpu": inst, "log_size": args.log_size, "window_size": window,
                       "parallelism": f"instance-sharded x{world}, RCCL all_gather of 96-byte results",
                       "pipelining": "step k+1 is submitted before step k's results are collected (submit/wait API)",
                       "bases": "precomputed window tables (built once, NOT the headline configuration)"
                                if args.precomputed_tables else
                                "persistent (converted once, NOT the headline configuration)"
                                if args.persistent_bases else "converted inside every MSM, as the reference does"},
            "stage_ms_per_msm": {"sort": round(sum(sort_ms) / len(sort_ms), 4),
                                 "accumulate": round(sum(acc_stage_ms) / len(acc_stage_ms), 4),
                                 "accumulate_kernel": round(acc_avg_ms, 4),
                                 "reduce": round(sum(red_ms) / len(red_ms), 4),
                                 "host_final": round(sum(fin_ms) / len(fin_ms), 4),
                                 "gpu_total": round(sum(tot_ms) / len(tot_ms), 4)},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    for j in range(inst):
        if tables:
            cfg.tables_free(d_pts[j])
        else:
            cfg.free(d_pts[j])
        cfg.free(d_sc[j])
    cfg.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
