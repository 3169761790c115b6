import sys, time, importlib
sys.path.insert(0, '.')
pkg = importlib.import_module("metal-msm-gpu-acceleration_amd")
from oracle import c_oracle as co, bn254_ref as o
cfg = pkg.setup_metal_state()
n = 1 << 12
pts, sc = co.gen_instance(o.SEED_BASE + 12, n)
want = cfg.msm(sc, pts, n)
dp, ds = cfg.alloc(64 * n), cfg.alloc(32 * n)
cfg.to_device(dp, pts); cfg.to_device(ds, sc)
for k in (1, 2, 2, 1):
    cfg.set_wait_timeout_ms(150)
    hold = cfg.test_hold(3000)
    t0 = time.perf_counter()
    h = cfg.submit_batch_device([ds] * k, [dp] * k, [n] * k)
    t1 = time.perf_counter()
    try:
        r = cfg.wait_batch(h)
        print(k, "no raise; submit %.1f ms wait %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3), r == [want] * k)
    except pkg.MsmError as e:
        print(k, "raised after %.1f ms: %s" % ((time.perf_counter() - t1) * 1e3, e))
    t2 = time.perf_counter()
    cfg.test_release(hold)
    print("  release took %.1f ms" % ((time.perf_counter() - t2) * 1e3))
    cfg.set_wait_timeout_ms(60000)
    cfg.synchronize()
    try:
        print("  rewait:", cfg.wait_batch(h) == [want] * k)
    except pkg.MsmError as e:
        print("  rewait err", e)
cfg.close()
