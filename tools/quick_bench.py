"""Quick device-resident timing of the MSM pipeline (development aid, not the graded bench)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("metal-msm-gpu-acceleration_amd")

logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,18,20").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = m.setup_metal_state()
for lg in logs:
    n = 1 << lg
    dp, ds = cfg.generate_instance(0xB2540000, n, True)
    outs = {}
    for c in (0, 13):
        cfg.set_window_size(c)
        for r in range(reps):
            t0 = time.time()
            out = cfg.msm_batch_device([ds], [dp], [n])[0]
            wall = (time.time() - t0) * 1e3
            t = cfg.timings()
        outs[c] = out
        print(f"log={lg} c={t.window_size} W={t.num_windows} wall={wall:.3f}ms gpu={t.total_gpu_ms:.3f} "
              f"conv={t.convert_ms:.3f} digits={t.digits_ms:.3f} sort={t.sort_ms:.3f} acc={t.accumulate_ms:.3f} "
              f"reduce={t.reduce_ms:.3f} final(host)={t.final_ms:.3f}", flush=True)
    cfg.set_window_size(0)
    print("  results agree across window sizes:", outs[0] == outs[13], outs[0][:8].hex(), flush=True)
    cfg.free(dp)
    cfg.free(ds)
