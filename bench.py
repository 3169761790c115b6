"""Headline benchmark: BN254 G1 MSM/s at log_size=20 (5 instances per GPU), see BASELINE.json.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (metal::msm::gpu_msm_h2c pipeline) over one batch of 5 synthetic
2^20-point instances per GPU, inputs already resident in HBM (device generator).  Instances shard across
ranks with no data-path collective; the per-instance results (96 B each) are all-gathered over RCCL.
Prints ONE JSON line on rank 0.

Launching.  One process per GPU.  `python bench.py --gpus N` with N > 1 and no RANK in the environment starts
the N ranks itself as child processes BEFORE anything touches the GPU (multi_gpu.launch_local_ranks; the parent
only waits and forwards the exit code).  Under `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N` the ranks are already there (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment).  Either way
WORLD_SIZE must equal --gpus and every rank needs its own visible GPU, otherwise the run fails loudly instead of
printing a line for fewer GPUs.  N = 1 goes through the same RCCL path (a one-rank process group).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "metal-msm-gpu-acceleration_amd"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm-steps", type=int, default=64,
                    help="untimed steps run BEFORE the --warmup steps (same pipelined loop, same count on every rank: "
                         "the steps contain the gather collective), so that a short run -- the driver's --steps 20 -- "
                         "does not time the GPU's clock ramp; 0 disables.  Reported as device_prewarm_steps.  "
                         "profiles/r03_bench_warmup_sensitivity.txt")
    ap.add_argument("--log-size", type=int, default=20)
    ap.add_argument("--instances", type=int, default=5, help="instances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the drop-in-caller figures (host slices, lone calls)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget")
    ap.add_argument("--depth", type=int, default=1, choices=[1, 2, 3],
                    help="steps submitted ahead of the one being collected (submit/wait API, at most 4 batches in flight)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend of the result gather: nccl (= RCCL over xGMI, the measured configuration) "
                         "or gloo (rehearsal of the N-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="REHEARSAL ONLY (with --backend gloo): ranks share the visible GPUs (local_rank %% visible); "
                         "the line is marked and is not a scaling measurement")
    ap.add_argument("--window", type=int, default=0, help="force the window size (0 = the library's automatic choice)")
    ap.add_argument("--precomputed-tables", action="store_true",
                    help="NOT the headline: window tables 2^(c w) P precomputed once per set of bases (SURVEY §8f N4), "
                         "one bucket set")
    ap.add_argument("--table-window", type=int, default=0, help="window bits of --precomputed-tables (0 = automatic)")
    ap.add_argument("--persistent-bases", action="store_true",
                    help="NOT the headline: bases converted once and kept resident (SURVEY §8f N4); the default "
                         "re-converts them inside every MSM like the reference does (msm.rs:152-153)")
    return ap.parse_args(argv)


def resolve_world(args, environ):
    """(rank, local_rank, world, must_launch) from --gpus and the environment; raises SystemExit on a mislaunch.
    No torch, no GPU: this runs before anything else."""
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" in environ:
        world = int(environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report "
                             f"a line for a different number of GPUs")
        return int(environ["RANK"]), int(environ.get("LOCAL_RANK", environ["RANK"])), world, False
    return 0, 0, args.gpus, args.gpus > 1


KERNEL_SOURCES = ("k_accumulate.hip", "bn254_ec29.hip.h", "bn254_fq29.hip.h", "device_common.hip.h")


def kernel_source_digest():
    """sha256 over the sources of the dominant kernel: profiles/pmc_traffic.json records it (tools/profile_summary.py)
    and `roofline.traffic` is only replayed from that file while it still matches -- a changed kernel with the old
    counter figures beside it would be a number from another program."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, PKG, "csrc", name), "rb").read())
    return h.hexdigest()


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    rank, local_rank, world, must_launch = resolve_world(args, os.environ)
    mg = importlib.import_module(PKG + ".multi_gpu")
    if must_launch:
        # N ranks as children, started before this process has imported torch or touched HIP
        rc = mg.launch_local_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + argv)
        if rc != 0:
            print(f"bench.py: a rank failed (exit {rc}); no result line", file=sys.stderr)
        raise SystemExit(rc)
    if "RANK" not in os.environ:            # N = 1: same RCCL path, a process group of one
        os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(mg.free_port()))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    # stdout carries exactly ONE line, the result: everything libraries print while the run is set up and timed (the
    # RCCL banner at communicator creation, for one) goes to stderr; the descriptor is restored for the JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    visible = torch.cuda.device_count()     # does not initialise the GPU
    shared = args.allow_shared_gpu and args.backend == "gloo"
    if args.allow_shared_gpu and not shared:
        raise SystemExit("bench.py: --allow-shared-gpu is a rehearsal switch and needs --backend gloo "
                         "(RCCL does not run two ranks on one device)")
    if not shared and (visible <= local_rank or visible < world):
        raise SystemExit(f"bench.py: rank {rank} of {world} needs GPU {local_rank} but only {visible} GPU(s) are "
                         f"visible: --gpus {world} cannot run here")
    if visible < 1:
        raise SystemExit("bench.py: no GPU visible")
    gpu_index = local_rank % visible if shared else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    t_init = time.perf_counter()
    try:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" IS RCCL on ROCm
            coll_dev = dev
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            coll_dev = torch.device("cpu")
        # first contact between the ranks: one tiny all-reduce before anything else is built on the communicator
        probe = torch.ones(1, device=coll_dev)
        dist.all_reduce(probe)
        if int(probe.item()) != world:
            raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
    except Exception as e:   # noqa: BLE001 -- the text is what the operator of a new multi-GPU host needs
        print(f"bench.py: rank {rank}/{world}: communicator setup failed ({args.backend}, "
              f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '<unset>')}, "
              f"MASTER_ADDR={os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}): "
              f"{type(e).__name__}: {e}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    comm_init_s = time.perf_counter() - t_init
    started_file = os.environ.get("MSM_AMD_RANK_STARTED_FILE")
    if started_file:                                   # tells multi_gpu.launch_local_ranks that this rank's group works
        open(started_file, "w").close()

    m = importlib.import_module(PKG)
    cfg = m.setup_metal_state(gpu_index)           # fails loudly without a gfx950 device
    # this rank's host threads (submission, Horner passes, later its parity check) stay on the CPUs local to its GPU
    # when sysfs tells which those are (threads started from here on inherit the mask); never an error
    numa_pinned = (world > 1) and m.lib().msm_amd_pin_thread_to_device(gpu_index) == 0
    if args.window:
        cfg.set_window_size(args.window)
    n = 1 << args.log_size
    inst = args.instances
    d_pts, d_sc = [], []
    for g in mg.instance_ids(rank, world, inst):   # rank r owns global instances r*inst .. r*inst+inst-1
        dp, ds = cfg.generate_instance(mg.instance_seed(g), n, True)
        d_pts.append(dp)
        d_sc.append(ds)
    raw_pts = list(d_pts)
    ns = [n] * inst
    point_layout = m.POINT_H2C_AFFINE
    tables = []
    if args.precomputed_tables:
        tables = [cfg.tables_build_device(dp, n, window_size=args.table_window) for dp in d_pts]
        d_pts = tables
        point_layout = m.POINT_TABLES
    elif args.persistent_bases:
        d_pts = [cfg.bases_prepare_device(dp, n) for dp in d_pts]
        point_layout = m.POINT_PREPARED

    gatherer = mg.ResultGatherer(dist, coll_dev, inst)

    def finish(handle):
        outs = cfg.wait_batch(handle)                    # host Horner pass of the batch
        gatherer.gather(outs)                            # RCCL gather of per-instance results over xGMI
        t = cfg.timings()                                # (enqueued; completes before the closing barrier)
        return outs, t

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
        cfg.synchronize()

    def run_steps(k, record):
        """k steps, software-pipelined: up to `--depth` steps are submitted before the oldest one is waited for
        (the library allows four batches in flight), so the GPU never idles between steps even when the host is
        briefly late; every step's results are produced and returned inside the loop."""
        outs, pending = None, []
        for _ in range(k):
            pending.append(cfg.submit_batch_device(d_sc, d_pts, ns, point_layout=point_layout))
            if len(pending) > args.depth:
                outs, t = finish(pending.pop(0))
                record(t)
        while pending:
            outs, t = finish(pending.pop(0))
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

    # Device state under this load (rank 0): rocm-smi asked for shader clock and package power while the pre-warm and
    # warm-up steps run -- the same pipelined load as the timed steps, but OUTSIDE the timed region, so the child
    # processes cannot touch `value`.  The pipeline is power-limited (profiles/r04_clock_power_during_bench.txt): the
    # clock it actually runs at belongs beside peaks that are quoted at the nominal 2.4 GHz.  Never an error.
    state_samples, state_stop = [], None
    def sample_device_state():
        import re
        import subprocess
        while not state_stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json", "-d", str(gpu_index)],
                                   capture_output=True, text=True, timeout=10)
                card = next(iter(json.loads(r.stdout).values()))
                clk = re.search(r"(\d+)\s*Mhz", card.get("sclk clock speed:", ""), re.I)
                pw = next((v for k, v in card.items() if "Power" in k and "(W)" in k), None)
                if clk:
                    state_samples.append((int(clk.group(1)), None if pw is None else float(pw)))
            except Exception:   # noqa: BLE001 -- no rocm-smi, another output format: the field stays null
                return
    state_thread = None

    cold_value = None
    if args.prewarm_steps > 0:
        # what the plain (W warm-up, K timed) recipe gives on a device whose clocks have not ramped up yet -- reported
        # beside `value` as `value_without_prewarm` so that a reader comparing by (steps, warmup) sees both
        run_steps(args.warmup, lambda t: None)
        barrier()
        tc0 = time.perf_counter()
        run_steps(args.steps, lambda t: None)
        barrier()
        cold_value = inst * world * args.steps / (time.perf_counter() - tc0)
        if rank == 0 and not args.no_extras:
            import threading
            state_stop = threading.Event()
            state_thread = threading.Thread(target=sample_device_state, daemon=True)
            state_thread.start()
        run_steps(args.prewarm_steps, lambda t: None)    # clocks up before anything is counted (not warm-up STEPS: the
        if state_thread is not None:                     # (keep the load up until rocm-smi has answered at least twice)
            extra = 0
            while len(state_samples) < 2 and state_thread.is_alive() and extra < 40:
                run_steps(8, lambda t: None)
                extra += 1
            state_stop.set()
            state_thread.join(timeout=15)
    outs = run_steps(args.warmup, lambda t: None)        # W steps are still run, the K steps still timed alone)
    barrier()
    t0 = time.perf_counter()
    outs = run_steps(args.steps, record)
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- what came back over RCCL: every rank's results, and which ranks answered
    allr = gatherer.fetch()                              # every rank holds every instance's result
    if allr[rank * inst:(rank + 1) * inst] != outs:
        raise SystemExit(f"rank {rank}: gathered results differ from the local ones")
    ids = torch.full((1,), rank, dtype=torch.int32, device=coll_dev)
    seen = torch.empty(world, dtype=torch.int32, device=coll_dev)
    dist.all_gather_into_tensor(seen, ids)
    ranks_seen = len(set(int(x) for x in seen.cpu().tolist()))
    te = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    elapsed = float(te.item())
    if ranks_seen != world or len(allr) != world * inst or any(r == bytes(96) for r in allr):
        raise SystemExit(f"bench.py: only {ranks_seen} of {world} ranks answered the RCCL gather")
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
    traffic, traffic_src = None, None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pmc_doc = None
    if os.path.exists(tf):
        try:
            pmc_doc = json.load(open(tf))
            if pmc_doc.get("kernel_source_sha256") == kernel_source_digest():
                traffic = pmc_doc.get(f"accumulate_log{args.log_size}_bytes_per_launch")
                traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of an earlier run of this command on " \
                              "this kernel source, 2*FETCH_SIZE + WRITE_SIZE per launch; not measured by this process)"
            else:
                pmc_doc = None
                traffic_src = "refused: profiles/pmc_traffic.json was taken from another version of the kernel source " \
                              "(kernel_source_sha256 differs) -- re-run tools/profile.sh"
        except Exception:
            traffic, pmc_doc = None, None
    # the kernel's OWN bytes (its window, 4-byte sorted entries, 64-byte packed bases, 144-byte XYZZ buckets)
    own = (4 + 64) * n * tm.num_windows + 144 * tm.num_windows * (1 << max(window - 1, 3))
    roofline = {"bound": "hbm", "kernel": "accumulate_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": a3, "avg_launch_ms": round(acc_avg_ms, 4),
                "own_layout": {"bytes_per_launch": own, "achieved": round(own / (acc_avg_ms * 1e-3) / 1e9, 2),
                               "frac": round(own / (acc_avg_ms * 1e-3) / 8e12, 5),
                               "note": f"this build's window {window}: (4 B sorted entry + 64 B packed base) per "
                                       f"point and window + 144 B per XYZZ bucket"},
                "whole_pipeline_GBps": round(L.msm_amd_algorithmic_bytes(n, ref_window, 0) /
                                             (sum(tot_ms) / len(tot_ms) * 1e-3) / 1e9, 2)}

    # ---- secondary roof (SURVEY 8d asks for it): the kernel is bound by VALU issue, most of it multiplier instructions
    # (v_mad_u64_u32 / v_mul_lo_u32), not by HBM.  The instruction counts per mixed addition (pti_madd, 8M + 2S on
    # 9 x 29-bit limbs) and per affine + affine start of a work item (pti_mmadd, 4M + 2S, incl. the first product of
    # pti_madd that the compiler speculates above the path split) are NOT typed in here: the build disassembles the
    # shipped k_accumulate object and counts them (tools/isa_counts.py -> metal-msm-gpu-acceleration_amd/
    # isa_counts.json).
    # ONE peak (round 4), measured in shader cycles by the waves themselves (s_memtime; tools/microbench/mul_occ.hip,
    # profiles/r04_mul_occupancy_microbench.txt, row "straight"): back-to-back independent v_mad_u64_u32 issue every
    # 4.0 cycles per SIMD at the shipped kernel's occupancy of 2 waves per SIMD (8.0 per wave) and NOT faster with more
    # waves (whole-launch timing, LAUNCH column of the same file: 1.9 ns from 2 to 8 waves).  x 1024 SIMDs x 64 lanes at the
    # nominal 2.4 GHz = 39.3 T lane-instructions/s.  A simple instruction (v_add_u32) issues every 2.9 cycles per SIMD
    # at the same occupancy (profiles/r04_valu_peak_microbench.txt); `valu_issue_utilisation` prices every VALU
    # instruction of the launch (SQ_INSTS_VALU of profiles/pmc_traffic.json) at those two figures against the SIMD
    # cycles of the launch.
    MAD_CYCLES_2W, SIMPLE_CYCLES_2W, SIMDS, CLOCK_HZ = 4.0, 2.9, 1024, 2.4e9
    peak_mad = SIMDS * 64 * CLOCK_HZ / MAD_CYCLES_2W / 1e12
    isa_path = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "isa_counts.json")
    try:
        isa = json.load(open(isa_path))
        per_madd, per_start = float(isa["multiplier_per_mixed_addition"]), float(isa["multiplier_per_affine_start"])
        valu_per_madd = isa.get("valu_per_mixed_addition")
    except Exception as e:   # noqa: BLE001
        raise SystemExit(f"bench.py: {isa_path} missing or unreadable ({e}): run __graft_entry__.build()")
    items = float(tm.reserved2[0])
    lane_madds = n * tm.num_windows - 2.0 * items          # first point of an item is free, second is the affine start
    mul_instr = lane_madds * per_madd + items * per_start
    util = None
    if pmc_doc is not None and valu_per_madd:
        try:
            insts = float(pmc_doc["per_kernel"][next(k for k in pmc_doc["per_kernel"] if k.startswith("accumulate_kernel"))]["SQ_INSTS_VALU"])
            mult_share = per_madd / float(valu_per_madd)
            busy = insts * (mult_share * MAD_CYCLES_2W + (1.0 - mult_share) * SIMPLE_CYCLES_2W)
            util = round(busy / (SIMDS * acc_avg_ms * 1e-3 * CLOCK_HZ), 4)
        except Exception:   # noqa: BLE001
            util = None
    valu = {"bound": "valu-int32-multiply", "achieved": round(mul_instr / (acc_avg_ms * 1e-3) / 1e12, 2),
            "peak": round(peak_mad, 2), "unit": "T lane-instr/s",
            "frac": round(mul_instr / (acc_avg_ms * 1e-3) / (peak_mad * 1e12), 4),
            "peak_definition": "independent v_mad_u64_u32 back to back at 2 waves/SIMD (the shipped kernel's occupancy): "
                               "one every 4.0 shader cycles per SIMD, measured in-kernel with s_memtime "
                               "(profiles/r04_mul_occupancy_microbench.txt) x 1024 SIMDs x 64 lanes x 2.4 GHz",
            "valu_issue_utilisation": util,
            "valu_issue_utilisation_definition": "SQ_INSTS_VALU of the launch (profiles/pmc_traffic.json) x (multiplier "
                                                 "share x 4.0 + rest x 2.9 cycles, the 2-wave issue intervals) / "
                                                 "(1024 SIMDs x launch time x 2.4 GHz); null when the counter file "
                                                 "is not of this kernel source",
            "work_items": int(items),
            "multiplier_instructions_per_mixed_addition": int(per_madd),
            "multiplier_instructions_per_affine_start": int(per_start),
            "valu_instructions_per_mixed_addition": valu_per_madd,
            "counts_source": "metal-msm-gpu-acceleration_amd/isa_counts.json (tools/isa_counts.py over the compiler's "
                             "assembly of the shipped k_accumulate.hip, written by the build)"}
    roofline["secondary"] = valu

    # ---- parity on every rank + CPU baseline (rank 0, single-GPU run only)
    cpu, parity_ok, extras = None, True, None
    if not args.no_cpu_baseline:
        from oracle import bn254_ref as o
        from oracle import c_oracle as co
        cores_all = co.default_threads()
        h_pts = [cfg.to_host(raw_pts[j], 64 * n) for j in range(inst)]
        h_sc = [cfg.to_host(d_sc[j], 32 * n) for j in range(inst)]
        if world == 1:
            cores = cores_all                            # every core the node gives this process
            t_cpu0 = time.perf_counter()
            done, cpu_outs = 0, []
            while True:
                for j in range(inst):
                    r, cpu_info = co.msm_best_ex(h_sc[j], h_pts[j], n, cores, 0)
                    if done < inst:
                        cpu_outs.append(r)
                    done += 1
                if time.perf_counter() - t_cpu0 > args.cpu_seconds or done >= 8 * inst:
                    break
            t_cpu = time.perf_counter() - t_cpu0
            for j in range(inst):          # parity gate: bit-exact canonical affine result
                if o.decode_jacobian_mont_le(outs[j]) != o.decode_jacobian_mont_le(cpu_outs[j]):
                    raise SystemExit(f"PARITY FAILURE: instance {j} GPU != CPU")
            # the product's own CPU MSM (msm_amd_host_msm: what `gpu_profiler .. cpu` and msm_best's CPU side run) on the
            # same instances and threads, next to the restatement of the reference's CPU path
            t_p0 = time.perf_counter()
            p_done = 0
            while True:
                for j in range(inst):
                    pr = m.host_msm(h_sc[j], h_pts[j], n, threads=cores)
                    if p_done < inst and o.decode_jacobian_mont_le(pr) != o.decode_jacobian_mont_le(cpu_outs[j]):
                        raise SystemExit(f"PARITY FAILURE: instance {j} product CPU MSM != oracle")
                    p_done += 1
                if time.perf_counter() - t_p0 > args.cpu_seconds / 4 or p_done >= 8 * inst:
                    break
            t_prod = time.perf_counter() - t_p0
            gpu_rate = inst * world * args.steps / elapsed
            cpu = {"value": round(done / t_cpu, 4), "unit": "MSM/s", "cores": cores, "cpu_model": cpu_model(),
                   "product_cpu_msm": {"value": round(p_done / t_prod, 4), "unit": "MSM/s", "cores": cores,
                                       "kind": "msm_amd_host_msm (this library's batched-affine Pippenger, AVX-512 "
                                               "IFMA where the host has it): faster than the restatement, NOT the "
                                               "reference's CPU path",
                                       "sample": f"{p_done} MSMs in {t_prod:.1f} s", "bit_exact_vs_oracle": True},
                   "gpu_over_cpu": {"vs_reference_restatement": round(gpu_rate / (done / t_cpu), 1),
                                    "vs_product_cpu_msm": round(gpu_rate / (p_done / t_prod), 1)},
                   "logical_cpus_visible": len(os.sched_getaffinity(0)),
                   "kind": "port", "algorithm": co.MSM_BEST_ALGORITHM, "shape": cpu_info,
                   "sample": f"{done} MSMs of 2^{args.log_size} points (the bench's own {inst} instances, "
                             f"{done // inst} pass(es)) in {t_cpu:.1f} s, oracle_msm_best = C restatement of "
                             f"halo2curves msm_best on {cores} threads", "bit_exact_vs_gpu": True}
        else:
            # every rank checks its OWN instances against the oracle (threads = its share of the host cores)
            thr = max(1, cores_all // world)
            for j in range(inst):
                if o.decode_jacobian_mont_le(outs[j]) != o.decode_jacobian_mont_le(co.msm_best(h_sc[j], h_pts[j], n, thr)):
                    parity_ok = False
        if rank == 0 and world == 1 and not args.no_extras and point_layout == m.POINT_H2C_AFFINE:
            extras = drop_in_caller_figures(m, cfg, h_sc, h_pts, n, inst, outs)
    okt = torch.tensor([1 if parity_ok else 0], dtype=torch.int32, device=coll_dev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if int(okt.item()) != 1:
        raise SystemExit("PARITY FAILURE on at least one rank (GPU != CPU oracle)")

    if rank == 0:
        total_msms = inst * world * args.steps
        line = {
            "metric": "BN254 G1 MSM/s at log_size=20 (5 instances)",
            "value": round(total_msms / elapsed, 3),
            "unit": "MSM/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "device_prewarm_steps": args.prewarm_steps,   # untimed load before the warm-up steps (GPU clock ramp), see --help
            "value_without_prewarm": None if cold_value is None else round(cold_value, 3),   # same W, same K, cold clocks
            "device_state_under_load": None if not state_samples else {
                "sclk_mhz": [c for c, _ in state_samples],
                "package_power_w": [w for _, w in state_samples],
                "source": "rocm-smi --showclocks --showpower --json, asked while the pre-warm steps run (the same "
                          "pipelined load as the timed steps, outside the timed region); the roofline peaks are quoted "
                          "at the nominal 2.4 GHz"},
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x8 (256-bit Montgomery integer)",
            "data": "synthetic",
            "rccl_ranks_seen": ranks_seen,
            "rccl_init_s": round(comm_init_s, 3),
            "ipc_mode_legacy_env": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "<unset>"),
            "ranks_pinned_to_gpu_numa_node": bool(numa_pinned),
            "collective": "rccl" if args.backend == "nccl" else "gloo (rehearsal, not RCCL)",
            "parity": ("bit-exact vs the CPU oracle on every rank's own instances" if not args.no_cpu_baseline
                       else "not checked in this run (--no-cpu-baseline)"),
            "config": {"workload": f"log_size={args.log_size}, {inst} instances per GPU, h2c BN254 G1 "
                                   f"(gpu_msm_h2c pipeline, window {window})",
                       "instances_per_gpu": inst, "log_size": args.log_size, "window_size": window,
                       "inputs": "device-resident, device-generated (msm_amd_generate_instance) before the timed "
                                 "region; nothing crosses PCIe inside it except the 96-byte results",
                       "parallelism": f"instance-sharded x{world}, "
                                      + ("RCCL all_gather of 96-byte results" if args.backend == "nccl" else
                                         "gloo all_gather of 96-byte results (REHEARSAL"
                                         + (", ranks share a GPU: not a scaling measurement)" if shared else ")")),
                       "pipelining": f"{args.depth} step(s) are submitted ahead of the one whose results are being "
                                     f"collected (submit/wait API)",
                       "bases": "precomputed window tables (built once, NOT the headline configuration)"
                                if args.precomputed_tables else
                                "persistent (converted once, NOT the headline configuration)"
                                if args.persistent_bases else "converted inside every MSM, as the reference does"},
            "stage_ms_per_msm": {"sort": round(sum(sort_ms) / len(sort_ms), 4),
                                 "accumulate": round(sum(acc_stage_ms) / len(acc_stage_ms), 4),
                                 "accumulate_kernel": round(acc_avg_ms, 4),
                                 "reduce": round(sum(red_ms) / len(red_ms), 4),
                                 "host_final": round(sum(fin_ms) / len(fin_ms), 4),
                                 "gpu_total_sum_of_stage_spans": round(sum(tot_ms) / len(tot_ms), 4)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "drop_in_caller": extras,
        }
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    for j in range(inst):
        if tables:
            cfg.tables_free(d_pts[j])
        elif args.persistent_bases:
            cfg.free(d_pts[j])
        cfg.free(raw_pts[j])
        cfg.free(d_sc[j])
    cfg.close()
    dist.destroy_process_group()


def drop_in_caller_figures(m, cfg, h_sc, h_pts, n, inst, expect):
    """What a caller of the reference API gets (never `value`): the reference's bench hands HOST slices to msm_best
    (benches/msm_benchmark.rs:116-121), so upload over PCIe is inside these numbers."""
    import statistics
    res = {}
    ns = [n] * inst
    cfg.msm_batch(h_sc, h_pts, ns)                               # warm-up (workspace, registrations)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = cfg.msm_batch(h_sc, h_pts, ns)
    dt = time.perf_counter() - t0
    if outs != expect:
        raise SystemExit("PARITY FAILURE: host-slice batch differs from the device-resident results")
    res["e2e_host_slices_MSM_per_s"] = round(inst * reps / dt, 2)
    res["e2e_host_slices_note"] = (f"msm_amd_msm_batch on pageable host buffers: {inst} x 2^{n.bit_length() - 1} "
                                   f"points, 96 MiB per instance uploaded inside the timed region")
    # the SAME call with the library's bases cache switched on (msm_amd_set_bases_cache: opt-in, no API change for
    # the caller): the converted bases of every slice stay resident, only the 32 MiB of scalars cross PCIe
    cfg.set_bases_cache(64 * n * inst * 2)
    cfg.msm_batch(h_sc, h_pts, ns)                               # fills the cache (misses)
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = cfg.msm_batch(h_sc, h_pts, ns)
    dt = time.perf_counter() - t0
    if outs != expect:
        raise SystemExit("PARITY FAILURE: cached host-slice batch differs from the device-resident results")
    st = cfg.bases_cache_stats()
    if st["hits"] != inst * reps or st["misses"] != inst:
        raise SystemExit(f"bases cache did not behave as expected: {st}")
    res["e2e_host_slices_bases_cache_MSM_per_s"] = round(inst * reps / dt, 2)
    res["e2e_host_slices_bases_cache_note"] = (
        "the same msm_amd_msm_batch call on the same pageable slices with msm_amd_set_bases_cache (opt-in): converted "
        f"bases stay resident ({st['bytes'] >> 20} MiB), every call re-hashes ~2 k sampled records per slice; cache-off "
        "figure: e2e_host_slices_MSM_per_s")
    cfg.set_bases_cache(0)
    # the same call on buffers the caller page-locked once (msm_amd_host_register): DMA uploads
    for b in h_sc + h_pts:
        cfg.host_register(b)
    cfg.msm_batch(h_sc, h_pts, ns)
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = cfg.msm_batch(h_sc, h_pts, ns)
    dt = time.perf_counter() - t0
    if outs != expect:
        raise SystemExit("PARITY FAILURE: registered host-slice batch differs from the device-resident results")
    res["e2e_host_slices_registered_MSM_per_s"] = round(inst * reps / dt, 2)
    # registered slices AND the bases cache: scalars by DMA, bases resident
    cfg.set_bases_cache(64 * n * inst * 2)
    cfg.msm_batch(h_sc, h_pts, ns)
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = cfg.msm_batch(h_sc, h_pts, ns)
    dt = time.perf_counter() - t0
    if outs != expect:
        raise SystemExit("PARITY FAILURE: cached registered batch differs from the device-resident results")
    res["e2e_host_slices_registered_bases_cache_MSM_per_s"] = round(inst * reps / dt, 2)
    cfg.set_bases_cache(0)
    for b in h_pts:
        cfg.host_unregister(b)
    # bases resident (converted once: an SRS), scalars from registered host memory: 32 MiB per instance over PCIe
    prepared = [cfg.bases_upload(p, n) for p in h_pts]
    cfg.msm_batch(h_sc, prepared, ns, point_layout=m.POINT_PREPARED)
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = cfg.msm_batch(h_sc, prepared, ns, point_layout=m.POINT_PREPARED)
    dt = time.perf_counter() - t0
    if outs != expect:
        raise SystemExit("PARITY FAILURE: prepared-bases batch differs from the device-resident results")
    res["e2e_resident_bases_host_scalars_MSM_per_s"] = round(inst * reps / dt, 2)
    for b in h_sc:
        cfg.host_unregister(b)
    for d in prepared:
        cfg.free(d)

    def lone(k):
        sc, pt = h_sc[0][:32 * k], h_pts[0][:64 * k]
        m.gpu_msm_h2c(sc, pt, cfg)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            m.gpu_msm_h2c(sc, pt, cfg)
            ts.append(time.perf_counter() - t0)
        return round(statistics.median(ts) * 1e3, 3)

    def lone_registered(k):   # the same call on slices the caller page-locked once (msm_amd_host_register)
        sc, pt = h_sc[0][:32 * k], h_pts[0][:64 * k]
        cfg.host_register(sc)
        cfg.host_register(pt)
        try:
            m.gpu_msm_h2c(sc, pt, cfg)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                m.gpu_msm_h2c(sc, pt, cfg)
                ts.append(time.perf_counter() - t0)
        finally:
            cfg.host_unregister(sc)
            cfg.host_unregister(pt)
        return round(statistics.median(ts) * 1e3, 3)

    def lone_resident(k):
        dp, ds = cfg.alloc(64 * k), cfg.alloc(32 * k)
        cfg.to_device(dp, h_pts[0][:64 * k])
        cfg.to_device(ds, h_sc[0][:32 * k])
        cfg.msm_batch_device([ds], [dp], [k])
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            cfg.msm_batch_device([ds], [dp], [k])
            ts.append(time.perf_counter() - t0)
        cfg.free(dp)
        cfg.free(ds)
        return round(statistics.median(ts) * 1e3, 3)

    def lone_best(k):
        sc, pt = h_sc[0][:32 * k], h_pts[0][:64 * k]
        m.msm_best(sc, pt, cfg)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            m.msm_best(sc, pt, cfg)
            ts.append(time.perf_counter() - t0)
        return round(statistics.median(ts) * 1e3, 3)

    # The same host-slice batch through the C++ CLI in a child process: no PyTorch in that process, so it binds the
    # SYSTEM HIP runtime (7.2), whose staged pageable copies overlap kernels -- what a Rust caller of the C ABI gets.
    # (This process is bound to the HIP 7.0 runtime inside the PyTorch wheel, where they serialise.)
    def cli(extra):
        import subprocess
        exe = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "gpu_profiler")
        cmd = [exe, str(n.bit_length() - 1), str(inst), "gpu", "5", "true", "--warmup", "1", "--json"] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            return round(1e3 / d["avg_instance_ms"], 2)
        except Exception as e:   # noqa: BLE001
            return f"unavailable ({type(e).__name__})"

    # the product's own CPU MSM (csrc/host_msm.hip: where the reference calls halo2curves -- gpu_profiler's cpu mode, the
    # CPU half of gpu_with_cpu) on the same first instance: product code, NOT the cpu_baseline (that is the oracle's)
    t0 = time.perf_counter()
    cpu_out = m.host_msm(h_sc[0], h_pts[0], n, 0)
    dt_cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu_again = m.host_msm(h_sc[0], h_pts[0], n, 0)
    dt_cpu = time.perf_counter() - t0
    if cpu_out != expect[0] or cpu_again != expect[0]:
        raise SystemExit("PARITY FAILURE: the product's CPU MSM differs from the GPU result (byte comparison)")
    res["product_cpu_msm"] = {"ms_per_msm": round(dt_cpu * 1e3, 2), "first_call_ms": round(dt_cold * 1e3, 2),
                              "threads": m.lib().msm_amd_host_threads(), "byte_identical_to_gpu_result": True,
                              "note": "msm_amd_host_msm on instance 0, no GPU involved: the second call (the first one "
                                      "also faults in the library's scratch block)"}

    res["cli_system_runtime"] = {
        "e2e_host_slices_MSM_per_s": cli([]),
        "e2e_host_slices_bases_cache_MSM_per_s": cli(["--bases-cache", str((64 * n * inst * 2) >> 20)]),
        "note": f"gpu_profiler {n.bit_length() - 1} {inst} gpu 5 true [--bases-cache]: msm_amd_msm_batch on pageable host "
                "slices from a process without PyTorch (system HIP runtime); instances of the same seeds, results not "
                "re-checked here (tests/test_gpu_profiler_cli.py, tests/test_gpu_bases_cache.py do)"}

    def with_cache(fn, k):
        cfg.set_bases_cache(64 * k * 2)
        try:
            return fn(k)
        finally:
            cfg.set_bases_cache(0)

    res["single_call_ms"] = {"msm_best_host_2^20": lone_best(min(n, 1 << 20)),
                             "msm_best_host_2^20_bases_cache": with_cache(lone_best, min(n, 1 << 20)),
                             "gpu_msm_h2c_host_2^20_bases_cache": with_cache(lone, min(n, 1 << 20)),
                             "gpu_msm_h2c_host_2^20": lone(min(n, 1 << 20)),
                             "gpu_msm_h2c_host_registered_2^20": lone_registered(min(n, 1 << 20)),
                             "gpu_msm_h2c_host_2^18": lone(min(n, 1 << 18)),
                             "resident_2^20": lone_resident(min(n, 1 << 20)),
                             "resident_2^18": lone_resident(min(n, 1 << 18)),
                             "note": "median wall time of ONE blocking call, nothing else in flight; msm_best = the "
                                     "entry point the reference's criterion bench calls per instance "
                                     "(benches/msm_benchmark.rs:116-121): zero-scalar filter on the device + MSM.  "
                                     "Host calls of 2^19 points or more run as pipelined point ranges when the upload "
                                     "can overlap kernels: always for page-locked slices; for pageable slices only on a "
                                     "HIP >= 7.2 runtime -- this process is bound to the HIP 7.0 runtime of the PyTorch "
                                     "wheel, whose pageable copies serialise (profiles/r02_lone_call_split.txt: "
                                     "3.1 ms pageable on the system runtime)"}
    return res


if __name__ == "__main__":
    main()
