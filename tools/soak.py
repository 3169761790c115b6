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
    checked = 0
    for r in range(a.rounds):
        cfg.set_window_size(rng.choice([0, 0, 0, 5, 9, 13, 15, 16, 17]))
        handles = []
        for b in range(4):                            # four batches in flight, 1..5 instances each, mixed sizes
            pick = [rng.choice(sizes) for _ in range(rng.choice([1, 2, 3, 5, 8, 12]))]   # 8+: threaded host passes
            h = cfg.submit_batch_device([pool[n][1] for n in pick], [pool[n][0] for n in pick], pick)
            handles.append((h, pick))
        for h, pick in rng.sample(handles, len(handles)):   # collected in random order
            outs = cfg.wait_batch(h)
            for n, out in zip(pick, outs):
                assert o.decode_jacobian_mont_le(out) == pool[n][2], (r, n)
                checked += 1
        if r % 3 == 0:                                # nothing in flight now: a LONE call (one stream, lone window policy),
            n = rng.choice(sizes)                     # unsplit or forced into pipelined point ranges, device or host buffers
            parts = rng.choice(["1", "2", "3", "4", "8"])
            os.environ["MSM_AMD_SPLIT"] = parts
            try:
                if rng.random() < 0.5:
                    out = cfg.msm_batch_device([pool[n][1]], [pool[n][0]], [n])[0]
                else:
                    out = m.gpu_msm_h2c(host[n][1], host[n][0], cfg)
            finally:
                del os.environ["MSM_AMD_SPLIT"]
            assert o.decode_jacobian_mont_le(out) == pool[n][2], ("lone", r, n, parts)
            checked += 1
    cfg.set_window_size(0)
    print(f"soak ok: {a.rounds} rounds, {checked} MSMs checked against the oracle")


if __name__ == "__main__":
    main()
