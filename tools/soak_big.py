#!/usr/bin/env python3
"""Soak of the LARGE-instance paths (2^19 .. 2^22 points): lone calls split into pipelined point ranges, host slices
(pageable, page-locked, through the bases cache), device-resident batches, msm_best, the multi-context entry points --
in random order, every result compared with the C oracle's answer for that instance.  Development aid.

  python tools/soak_big.py [--rounds 150] [--seed 1]
"""
import argparse
import importlib
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bn254_ref as o          # noqa: E402
from oracle import c_oracle as co          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    m = importlib.import_module("metal-msm-gpu-acceleration_amd")
    cfg = m.setup_metal_state()
    second = m.setup_metal_state(cfg.device())
    rng = random.Random(a.seed)
    sizes = [(1 << 19) + 5, 1 << 20, (1 << 21) + 1, 1 << 22]
    dev, host, want = {}, {}, {}
    t0 = time.time()
    for n in sizes:
        pb, sb = co.gen_instance(o.SEED_BASE + 7700 + (n & 0xFF) + n.bit_length(), n)
        want[n] = o.decode_jacobian_mont_le(co.msm_best(sb, pb, n))
        dp, ds = cfg.alloc(64 * n), cfg.alloc(32 * n)
        cfg.to_device(dp, pb)
        cfg.to_device(ds, sb)
        dev[n], host[n] = (dp, ds), (pb, sb)
    print(f"instances and oracle answers ready after {time.time() - t0:.1f} s", flush=True)
    checked = 0
    registered = set()
    for r in range(a.rounds):
        n = rng.choice(sizes)
        pb, sb = host[n]
        dp, ds = dev[n]
        os.environ["MSM_AMD_SPLIT"] = rng.choice(["1", "2", "4", "8", "0"])
        if os.environ["MSM_AMD_SPLIT"] == "0":
            del os.environ["MSM_AMD_SPLIT"]                      # the library's own choice
        cfg.set_bases_cache(rng.choice([0, 0, 1 << 30]))
        mode = rng.randrange(8)
        try:
            if mode == 0:
                outs = cfg.msm_batch_device([ds], [dp], [n])
            elif mode == 1:                                      # two sizes in one resident batch
                n2 = rng.choice(sizes)
                outs = cfg.msm_batch_device([ds, dev[n2][1]], [dp, dev[n2][0]], [n, n2])
                assert o.decode_jacobian_mont_le(outs[1]) == want[n2], ("batch2", r, n2)
                checked += 1
                outs = outs[:1]
            elif mode == 2:
                outs = [m.gpu_msm_h2c(sb, pb, cfg)]
            elif mode == 3:
                outs = [m.msm_best(sb, pb, cfg)]
            elif mode == 4:                                      # host-slice batch of two (uploader thread, cache fill / hit)
                outs = cfg.msm_batch([sb, sb], [pb, pb], [n, n])
                assert outs[0] == outs[1]
                outs = outs[:1]
            elif mode == 5:                                      # page-locked scalars (the stream-ordering fix of this round)
                if n not in registered and len(registered) < 2:
                    cfg.host_register(sb)
                    registered.add(n)
                outs = [m.gpu_msm_h2c(sb, pb, cfg)]
            elif mode == 6:
                outs = [m.msm_range_multi([cfg, second], sb, pb, n)]
            else:
                prepared = cfg.bases_upload(pb, n)
                try:
                    outs = [cfg.msm_prepared(sb, prepared, n)]
                finally:
                    cfg.free(prepared)
        finally:
            os.environ.pop("MSM_AMD_SPLIT", None)
        assert o.decode_jacobian_mont_le(outs[0]) == want[n], (r, mode, n)
        checked += 1
        if r % 25 == 24:
            print(f"round {r + 1}: {checked} MSMs ok", flush=True)
    for n in registered:
        cfg.host_unregister(host[n][1])
    second.close()
    print(f"big soak ok: {a.rounds} rounds, {checked} MSMs of 2^19..2^22 points checked against the oracle")


if __name__ == "__main__":
    main()
