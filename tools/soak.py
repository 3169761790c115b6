#!/usr/bin/env python3
"""Soak test of the multi-stream pipeline: batches of 1-12 instances of mixed sizes and windows, four batches in flight, lone calls in between, every result
compared with the C oracle (restated msm_best).  Development aid; the graded tests live in tests/.

  python tools/soak.py [--rounds 60] [--seed 1]
"""
import argparse
import importlib
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bn254_ref as o          # noqa: E402
from oracle import c_oracle as co          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    m = importlib.import_module("metal-msm-gpu-acceleration_amd")
    cfg = m.setup_metal_state()
    rng = random.Random(a.seed)
    sizes = [1, 7, 33, 100, 1000, 4097, 1 << 14, 1 << 16, (1 << 17) + 3, 1 << 18]
    pool, host = {}, {}
    for n in sizes:                                   # one resident instance per size, oracle answer cached
        dp, ds = cfg.generate_instance(o.SEED_BASE + 9000 + n, n, True)
        pb, sb = co.gen_instance(o.SEED_BASE + 9000 + n, n)
        pool[n] = (dp, ds, o.decode_jacobian_mont_le(co.msm_best(sb, pb, n)))
        host[n] = (pb, sb)
    # one instance built to hit the rare paths of the accumulate kernel in EVERY window: (k, P) next to (k, -P) (the sum
    # of a bucket vanishes: lane state back to "empty"), (k, P) twice (doubling), identity bases, zero scalars
    n_sp = 4097
    pb, sb = bytearray(host[n_sp][0]), bytearray(host[n_sp][1])
    for i in range(0, 128, 2):
        x, y = pb[64 * i:64 * i + 32], int.from_bytes(pb[64 * i + 32:64 * i + 64], "little")
        if i < 64:   # -P: y -> p - y in Montgomery form is (p - y_mont) as well
            pb[64 * (i + 1):64 * (i + 1) + 64] = bytes(x) + ((o.P - y) % o.P).to_bytes(32, "little")
        else:
            pb[64 * (i + 1):64 * (i + 1) + 64] = pb[64 * i:64 * i + 64]
        sb[32 * (i + 1):32 * (i + 1) + 32] = sb[32 * i:32 * i + 32]
    for i in rng.sample(range(128, n_sp), 30):
        pb[64 * i:64 * i + 64] = bytes(64)
    for i in rng.sample(range(128, n_sp), 30):
        sb[32 * i:32 * i + 32] = bytes(32)
    pb, sb = bytes(pb), bytes(sb)
    special = -n_sp                                   # key of the special instance in pool / host
    dp, ds = cfg.alloc(64 * n_sp), cfg.alloc(32 * n_sp)
    cfg.to_device(dp, pb)
    cfg.to_device(ds, sb)
    pool[special] = (dp, ds, o.decode_jacobian_mont_le(co.msm_best(sb, pb, n_sp)))
    host[special] = (pb, sb)
    keys = sizes + [special]
    second = m.setup_metal_state(cfg.device())        # a second context on the same device (multi-context entry points)
    size_of = lambda k: abs(k)                        # noqa: E731
    checked = 0
    for r in range(a.rounds):
        cfg.set_window_size(rng.choice([0, 0, 0, 5, 9, 13, 15, 16, 17]))
        handles = []
        for b in range(4):                            # four batches in flight, 1..5 instances each, mixed sizes
            pick = [rng.choice(keys) for _ in range(rng.choice([1, 2, 3, 5, 8, 12]))]   # 8+: threaded host passes
            h = cfg.submit_batch_device([pool[n][1] for n in pick], [pool[n][0] for n in pick], [size_of(n) for n in pick])
            handles.append((h, pick))
        for h, pick in rng.sample(handles, len(handles)):   # collected in random order
            outs = cfg.wait_batch(h)
            for n, out in zip(pick, outs):
                assert o.decode_jacobian_mont_le(out) == pool[n][2], (r, n)
                checked += 1
        if r % 3 == 0:                                # nothing in flight now: a LONE call (one stream, lone window policy),
            n = rng.choice(keys)                      # unsplit or forced into pipelined point ranges, device or host buffers
            parts = rng.choice(["1", "2", "3", "4", "8"])
            os.environ["MSM_AMD_SPLIT"] = parts
            cfg.set_bases_cache(rng.choice([0, 0, 64 << 20]))   # the opt-in bases cache on and off (host calls)
            try:
                mode = rng.random()
                if mode < 0.4:
                    out = cfg.msm_batch_device([pool[n][1]], [pool[n][0]], [size_of(n)])[0]
                elif mode < 0.7:
                    out = m.gpu_msm_h2c(host[n][1], host[n][0], cfg)
                elif mode < 0.8:
                    out = m.msm_best(host[n][1], host[n][0], cfg)
                elif mode < 0.88:                     # ONE instance over two contexts by point range
                    out = m.msm_range_multi([cfg, second], host[n][1], host[n][0], size_of(n))
                elif mode < 0.94:                     # the instance loop over two contexts
                    out = m.msm_batch_multi([cfg, second], [host[n][1]] * 3, [host[n][0]] * 3, [size_of(n)] * 3)
                    assert out[0] == out[1] == out[2]
                    out = out[0]
                else:                                 # a host-slice batch of three
                    out = cfg.msm_batch([host[n][1]] * 3, [host[n][0]] * 3, [size_of(n)] * 3)
                    assert out[0] == out[1] == out[2]
                    out = out[0]
            finally:
                del os.environ["MSM_AMD_SPLIT"]
            assert o.decode_jacobian_mont_le(out) == pool[n][2], ("lone", r, n, parts)
            checked += 1
    cfg.set_window_size(0)
    second.close()
    print(f"soak ok: {a.rounds} rounds, {checked} MSMs checked against the oracle")


if __name__ == "__main__":
    main()
