"""msm_best's size dispatch (src/metal/msm.rs:440-444 sends n < 2^17 to the CPU): where is the crossover on MI355X?

Times ONE blocking call per size through the C ABI: msm_amd_gpu_msm_h2c (GPU, host buffers, upload included) against
the product's host bucket method (msm_amd_gpu_with_cpu with split_at = 0: every point goes to the CPU half), 1 thread
and all threads.  Prints a table and the largest n for which the host wins; kCpuDispatchBelow in csrc/msm_host.hip is
set from it (profiles/r02_crossover.txt)."""
import importlib
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("metal-msm-gpu-acceleration_amd")
from oracle import c_oracle as co  # noqa: E402  (input generator only)


def med(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3


def main():
    cfg = m.setup_metal_state()
    os.environ.setdefault("MSM_AMD_CPU_BELOW", "0")
    pts, sc = co.gen_instance(0xB2540000 + 9, 1 << 14)
    print(f"{'n':>6} {'gpu_ms':>9} {'host1_ms':>9} {'hostN_ms':>9}")
    last_host_win = 0
    for lg in range(0, 15):
        n = 1 << lg
        s, p = sc[:32 * n], pts[:64 * n]
        gpu = med(lambda: m.gpu_msm_h2c(s, p, cfg), 15)
        h1 = med(lambda: m.gpu_with_cpu(s, p, cfg, split_at=0, cpu_threads=1), 5 if n > 512 else 15)
        hn = med(lambda: m.gpu_with_cpu(s, p, cfg, split_at=0, cpu_threads=0), 5 if n > 512 else 15)
        if min(h1, hn) < gpu:
            last_host_win = n
        print(f"{n:>6} {gpu:>9.3f} {h1:>9.3f} {hn:>9.3f}", flush=True)
    print(f"host bucket method wins up to n = {last_host_win}")
    cfg.close()


if __name__ == "__main__":
    main()
