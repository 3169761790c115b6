"""Sweep of the helper-thread count of the staged pageable upload (torch-hosted process = HIP 7.0 runtime):
python tools/dbg/stage_helpers.py  ->  MSM/s of msm_batch on 5 pageable 2^20-point slices per MSM_AMD_STAGE_HELPERS."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (binds the wheel's HIP runtime first, like bench.py)
m = importlib.import_module("metal-msm-gpu-acceleration_amd")
n, inst = 1 << 20, 5
for helpers in (1, 2, 3, 4, 5, 7, 3):
    os.environ["MSM_AMD_STAGE_HELPERS"] = str(helpers)
    cfg = m.setup_metal_state(0)
    hs, hp = [], []
    for j in range(inst):
        dp, ds = cfg.generate_instance(0xB2540000 + j, n, True)
        hp.append(cfg.to_host(dp, 64 * n)); hs.append(cfg.to_host(ds, 32 * n)); cfg.free(dp); cfg.free(ds)
    want = cfg.msm_batch(hs, hp, [n] * inst)
    line = f"helpers={helpers}:"
    for cache in (0, 64 * n * inst * 2):
        cfg.set_bases_cache(cache)
        cfg.msm_batch(hs, hp, [n] * inst)
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(3):
                outs = cfg.msm_batch(hs, hp, [n] * inst)
            best = max(best, inst * 3 / (time.perf_counter() - t0))
        assert outs == want
        line += f"  cache={'on ' if cache else 'off'} {best:6.1f} MSM/s"
    t0 = time.perf_counter()
    for _ in range(5):
        cfg.msm_batch(hs[:1], hp[:1], [n])
    line += f"  single 2^20 call {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms"
    print(line, flush=True)
    cfg.close()
