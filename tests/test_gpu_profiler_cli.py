"""gpu_profiler CLI (src/bin/gpu_profiler.rs:17-172): argument order, the two report lines, the modes, and the
instance-file cache (preprocess.rs:143-202).  The CLI is a child process: one at a time, each a few hundred ms."""
import json
import os
import subprocess

import pytest

from oracle import bn254_ref as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metal-msm-gpu-acceleration_amd", "gpu_profiler")


def run(*args, ok=True):
    r = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert (r.returncode == 0) == ok, r.stderr
    return r


def test_cli_is_built():
    assert os.access(EXE, os.X_OK), "run __graft_entry__.build()"


@pytest.mark.gpu
def test_report_lines_and_result_against_oracle():
    seed = 0xB2540000 + 500
    r = run(10, 2, "gpu", 1, "--seed", seed, "--json")
    for line in ("Log instance size: 10", "Number of instances: 2", "Run mode: gpu", "Retries: 1",
                 "Total Execution Time:", "Average Instance Execution Time:"):
        assert line in r.stderr
    out = json.loads(r.stdout)
    pts, scs = o.gen_instance(seed, 1024)
    want = o.msm_pippenger(scs, pts, 8)
    assert out["result0_x_le_hex"] == o.int_to_le_bytes32(o.fq_to_mont(want[0])).hex()


@pytest.mark.gpu
def test_all_modes_agree_on_the_result():
    res = {m: json.loads(run(12, 1, m, 1, "--json").stdout)["result0_x_le_hex"]
           for m in ("gpu", "gpu_resident", "gpu_cpu", "best_gpu", "cpu", "check")}
    assert len(set(res.values())) == 1, res
    assert "Invalid RUN_MODE" in run(12, 1, "bogus", ok=False).stderr      # gpu_profiler.rs:167-170


@pytest.mark.gpu
def test_instance_file_cache_roundtrip(tmp_path):
    plain = json.loads(run(11, 2, "gpu", 1, "--json").stdout)["result0_x_le_hex"]
    first = run(11, 2, "gpu", 1, "--json", "--vec-dir", tmp_path)
    assert "Saving MSM instances to file" in first.stderr
    assert os.path.getsize(tmp_path / "msm_11x2.bin") == 8 + 2 * (16 + 2048 * 144)
    again = run(11, 2, "gpu_resident", 1, "--json", "--vec-dir", tmp_path)
    assert "Loading MSM instances from file" in again.stderr
    assert json.loads(first.stdout)["result0_x_le_hex"] == plain == json.loads(again.stdout)["result0_x_le_hex"]
    # a cache file of another shape is refused like the reference does (preprocess.rs:186-195)
    os.rename(tmp_path / "msm_11x2.bin", tmp_path / "msm_12x2.bin")
    assert "File mismatch" in run(12, 2, "gpu", 1, "--vec-dir", tmp_path, ok=False).stderr
