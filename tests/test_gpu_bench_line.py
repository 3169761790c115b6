"""The bench contract on a real GPU: `python bench.py` prints exactly ONE JSON line on stdout with the agreed fields."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--cpu-seconds", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    x = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rccl_ranks_seen"):
        assert key in x, key
    assert x["unit"] == "MSM/s" and x["n_gpus"] == 1 and x["steps"] == 3 and x["warmup"] == 1
    assert x["higher_is_better"] is True and x["scaling"] == "weak" and x["vs_baseline"] is None
    assert x["rccl_ranks_seen"] == 1 and x["value"] > 100 and "workload" in x["config"]
    assert abs(x["value"] - 5 * 1000.0 / x["ms_per_step"]) / x["value"] < 0.02      # 5 instances per step
    r = x["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.02 < r["frac"] < 1.0
    c = x["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["bit_exact_vs_gpu"] is True
    assert x["drop_in_caller"]["single_call_ms"]["resident_2^18"] < 5.0
    # round 3: where the inputs live, build-derived instruction counts, the bases-cache figures beside the cache-off ones
    assert "device-resident" in x["config"]["inputs"] and x["device_prewarm_steps"] == 64
    sec = r["secondary"]
    assert 1500 <= sec["multiplier_instructions_per_mixed_addition"] <= 1600 and "isa_counts.json" in sec["counts_source"]
    assert 0.3 < sec["frac"] < 0.9
    d = x["drop_in_caller"]
    assert d["e2e_host_slices_bases_cache_MSM_per_s"] > d["e2e_host_slices_MSM_per_s"] > 50
    assert d["product_cpu_msm"]["byte_identical_to_gpu_result"] is True
    assert isinstance(d["cli_system_runtime"]["e2e_host_slices_bases_cache_MSM_per_s"], float)
    # late round 4: the clock and package power the device runs this load at (rocm-smi, outside the timed region); the
    # field is null where rocm-smi does not answer, never an error
    assert "device_state_under_load" in x
    st = x["device_state_under_load"]
    if st is not None:
        assert len(st["sclk_mhz"]) >= 1 and all(500 < c < 3000 for c in st["sclk_mhz"])
