"""One MSM of 2^LOG points (default 26), device-resident, checked by the dlog identity (no MSM code on the checking side):
python tools/big_msm.py [LOG]"""
import importlib, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bn254_ref as o
from oracle import c_oracle as co
m = importlib.import_module("metal-msm-gpu-acceleration_amd")
log = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log
cfg = m.setup_metal_state(0)
rng = random.Random(log)
a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
t0 = time.time()
dp, ds = cfg.generate_instance(o.SEED_BASE + 2600 + log, n, True)
cfg.free(dp)
sb = cfg.to_host(ds, 32 * n)
print(f"scalars generated {time.time() - t0:.1f} s", flush=True)
pb, expect = co.dlog_instance(a0, d, sb, n)
print(f"dlog bases built on the host {time.time() - t0:.1f} s", flush=True)
del sb
dpts = cfg.alloc(64 * n)
cfg.to_device(dpts, pb)
del pb
for rep in range(3):
    t1 = time.perf_counter()
    out = cfg.msm_batch_device([ds], [dpts], [n])[0]
    dt = time.perf_counter() - t1
    ok = o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(expect)
    print(f"2^{log} points: {dt * 1e3:.1f} ms, window {cfg.last_window_size() if hasattr(cfg, 'last_window_size') else '?'}, dlog identity {'ok' if ok else 'MISMATCH'}", flush=True)
    assert ok
cfg.free(ds); cfg.free(dpts); cfg.close()
