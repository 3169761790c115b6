"""Accumulate kernel alone, variants alternating in ONE process (one context per variant): mean / min / std of the
kernel's HIP-event time over lone device-resident 2^20 calls.   python tools/ab_acc_kernel.py "1 6" [reps] [log]"""
import importlib
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("metal-msm-gpu-acceleration_amd")
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1 6").split()]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lg = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 1 << lg
ctxs = {}
for v in variants:
    os.environ["MSM_AMD_ACC_VARIANT"] = str(v)
    ctxs[v] = m.setup_metal_state()
data = {}
ref = None
for v in variants:
    dp, ds = ctxs[v].generate_instance(0xB2540000, n, True)
    data[v] = (dp, ds)
times = {v: [] for v in variants}
for r in range(reps + 3):
    for v in variants:
        dp, ds = data[v]
        out = ctxs[v].msm_batch_device([ds], [dp], [n])[0]
        if ref is None:
            ref = out
        assert out == ref, f"variant {v} disagrees"
        if r >= 3:
            times[v].append(ctxs[v].timings().accumulate_kernel_ms)
for v in variants:
    t = times[v]
    print(f"variant {v}: accumulate kernel alone mean {statistics.mean(t):.4f} ms  median {statistics.median(t):.4f}  "
          f"min {min(t):.4f}  stdev {statistics.pstdev(t):.4f}  ({len(t)} lone 2^{lg} calls, alternating)", flush=True)
