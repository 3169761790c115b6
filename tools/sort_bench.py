#!/usr/bin/env python3
"""Micro-benchmark of the stand-alone device sort, mirroring benches/sort_buckets_indices_benchmark.rs:10-32 of
the reference: len = 17 * 2^log pairs for log in {16, 18, 20, 22}, keys uniform below 16 * len, values below len,
seed 42.  The reference times its CPU (rayon) sort on a shared Metal buffer; here the pairs are resident in HBM
and the time is the device time of msm_amd_sort_pairs_device.

Prints one JSON line per size: elements/s (criterion's Throughput::Elements) and the HBM rate of the algorithmic
traffic (per 8-bit pass: read for the histogram + read and write for the scatter = 24 B per pair).

  python tools/sort_bench.py [--logs 16,18,20,22] [--reps 5] [--full-keys]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logs", default="16,18,20,22")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--full-keys", action="store_true", help="sort on all 32 key bits (4 passes) instead of the "
                    "bits the key range needs")
    a = ap.parse_args()
    m = importlib.import_module("metal-msm-gpu-acceleration_amd")
    cfg = m.setup_metal_state()
    for log in map(int, a.logs.split(",")):
        n = 17 << log
        rng = np.random.default_rng(42)
        pairs = np.empty((n, 2), dtype=np.uint32)
        pairs[:, 0] = rng.integers(0, min(16 * n, 1 << 32), size=n, dtype=np.uint64).astype(np.uint32)
        pairs[:, 1] = rng.integers(0, n, size=n, dtype=np.uint32)
        key_bits = 32 if a.full_keys else max(1, int(16 * n - 1).bit_length())
        passes = (key_bits + 7) // 8
        d = cfg.alloc(8 * n)
        times = []
        for rep in range(a.reps + 1):
            cfg.to_device(d, pairs.tobytes())            # every repetition sorts the same unsorted input
            times.append(cfg.sort_pairs_device(d, n, key_bits))
        got = np.frombuffer(cfg.to_host(d, 8 * n), dtype=np.uint32).reshape(n, 2)
        cfg.free(d)
        ok = bool(np.all(got[1:, 0] >= got[:-1, 0])) and \
            bool(np.array_equal(np.sort(got.view(np.uint64).ravel()), np.sort(pairs.view(np.uint64).ravel())))
        ms = float(np.median(times[1:]))
        print(json.dumps({"bench": "sort_buckets_indices", "log_length": log, "pairs": n, "key_bits": key_bits,
                          "passes": passes, "ms": round(ms, 4), "elements_per_s": round(n / ms * 1e3),
                          "algorithmic_GBps": round(24 * n * passes / ms / 1e6, 1), "sorted_and_permutation": ok}))
        if not ok:
            sys.exit(1)
    cfg.close()


if __name__ == "__main__":
    main()
