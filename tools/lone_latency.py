"""Latency of ONE blocking device-resident MSM call per size (development aid): median wall time and stage spans."""
import importlib
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("metal-msm-gpu-acceleration_amd")

logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,18,20").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = m.setup_metal_state()
if os.environ.get("LONE_C"):
    cfg.set_window_size(int(os.environ["LONE_C"]))
for lg in logs:
    n = 1 << lg
    dp, ds = cfg.generate_instance(0xB2540000, n, True)
    if os.environ.get("LONE_HOST"):   # the same call from HOST buffers (gpu_msm_h2c): the upload is part of the call
        hp, hs = cfg.to_host(dp, 64 * n), cfg.to_host(ds, 32 * n)
        if os.environ.get("LONE_HOST") == "registered":
            cfg.host_register(hp)
            cfg.host_register(hs)
        for _ in range(4):
            out = m.gpu_msm_h2c(hs, hp, cfg)
        wall = []
        for r in range(reps):
            t0 = time.perf_counter()
            out = m.gpu_msm_h2c(hs, hp, cfg)
            wall.append((time.perf_counter() - t0) * 1e3)
        t = cfg.timings()
        print(f"log={lg} c={t.window_size} parts={t.reserved} lone HOST-buffer call ({os.environ['LONE_HOST']}): median "
              f"{statistics.median(wall):.4f} ms, min {min(wall):.4f} ms | x={out[:8].hex()}", flush=True)
        if os.environ.get("LONE_HOST") == "registered":
            cfg.host_unregister(hp)
            cfg.host_unregister(hs)
        cfg.free(dp)
        cfg.free(ds)
        continue
    for _ in range(6):   # every workspace allocated
        out = cfg.msm_batch_device([ds], [dp], [n])[0]
    wall, acc, sub = [], {}, []
    for r in range(reps):   # the same call split into its two halves: how long does the host need to enqueue it?
        t0 = time.perf_counter()
        h = cfg.submit_batch_device([ds], [dp], [n])
        sub.append((time.perf_counter() - t0) * 1e3)
        cfg.wait_batch(h)
    for r in range(reps):
        t0 = time.perf_counter()
        out = cfg.msm_batch_device([ds], [dp], [n])[0]
        wall.append((time.perf_counter() - t0) * 1e3)
        t = cfg.timings()
        for k in ("convert_ms", "digits_ms", "sort_ms", "accumulate_ms", "reduce_ms", "final_ms"):
            acc.setdefault(k, []).append(getattr(t, k))
    med = {k: statistics.median(v) for k, v in acc.items()}
    print(f"log={lg} c={t.window_size} parts={t.reserved} lone call: median {statistics.median(wall):.4f} ms, min {min(wall):.4f} ms, "
          f"enqueue {statistics.median(sub):.4f} ms | "
          + " ".join(f"{k[:-3]}={v:.3f}" for k, v in med.items()) + f" | x={out[:8].hex()}", flush=True)
    cfg.free(dp)
    cfg.free(ds)
