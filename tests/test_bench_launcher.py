"""bench.py's own N-rank launcher: child environments, exit-code forwarding, loud mislaunch
failures.  No GPU: the children here are stubs, and the one real bench.py run must FAIL because this container
has no GPU."""
import importlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
bench = importlib.import_module("bench")


def test_rank_environments():
    envs = mg.rank_environments(4, 29517, base_env={"PATH": "/bin", "RANK": "9"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["LOCAL_WORLD_SIZE"] == "4" for e in envs)
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29517" for e in envs)
    assert all(e["PATH"] == "/bin" and "HSA_ENABLE_IPC_MODE_LEGACY" not in e for e in envs)     # inherited, never defaulted
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
               for e in mg.rank_environments(2, 1, base_env={"HSA_ENABLE_IPC_MODE_LEGACY": "1"}))
    with pytest.raises(ValueError):
        mg.rank_environments(0, 1)


STUB = r"""
import json, os, sys
out = sys.argv[1]
r = os.environ["RANK"]
json.dump({k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")},
          open(os.path.join(out, "rank%s.json" % r), "w"))
sys.exit(int(sys.argv[2]) if r == sys.argv[3] else 0)
"""


def test_launcher_starts_n_children_and_each_answers(tmp_path):
    rc = mg.launch_local_ranks(3, [sys.executable, "-c", STUB, str(tmp_path), "0", "-1"], timeout=60)
    assert rc == 0
    got = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [g["RANK"] for g in got] == ["0", "1", "2"]
    assert len({g["MASTER_PORT"] for g in got}) == 1 and all(g["WORLD_SIZE"] == "3" for g in got)


def test_launcher_forwards_a_failing_rank(tmp_path):
    rc = mg.launch_local_ranks(2, [sys.executable, "-c", STUB, str(tmp_path), "7", "1"], timeout=60)
    assert rc == 7


def test_launcher_kills_the_survivors_of_a_failed_rank(tmp_path):
    hang = "import os, sys, time\nif os.environ['RANK'] == '0': sys.exit(5)\ntime.sleep(600)\n"
    rc = mg.launch_local_ranks(2, [sys.executable, "-c", hang], timeout=60)
    assert rc == 5


FIRST_CONTACT = r"""
import os, sys
mode = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "<unset>")
if mode != sys.argv[1]:
    print("hipIpcGetMemHandle: invalid argument (stub) with mode", mode, file=sys.stderr)
    sys.exit(3)
open(os.environ["MSM_AMD_RANK_STARTED_FILE"], "w").close()
if len(sys.argv) > 2 and os.environ["RANK"] == "1":
    sys.exit(int(sys.argv[2]))
"""


def test_first_contact_failure_is_retried_once_with_the_other_ipc_mode(tmp_path):
    """A rank that dies before its process group works: the parent starts ONE fresh set of children with
    HSA_ENABLE_IPC_MODE_LEGACY flipped, says which setting worked, and keeps every rank's stderr."""
    said = []
    base = {k: v for k, v in os.environ.items()}
    base["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    rc = mg.launch_local_ranks(2, [sys.executable, "-c", FIRST_CONTACT, "1"], base_env=base, timeout=60,
                               rank_dir=str(tmp_path), log=said.append)
    assert rc == 0
    assert any("came up with HSA_ENABLE_IPC_MODE_LEGACY=1" in m for m in said)
    assert "invalid argument (stub) with mode 0" in (tmp_path / "attempt0" / "rank0.err").read_text()
    # inherited setting works: one attempt, nothing said about modes
    said.clear()
    assert mg.launch_local_ranks(2, [sys.executable, "-c", FIRST_CONTACT, "0"], base_env=base, timeout=60,
                                 rank_dir=str(tmp_path / "b"), log=said.append) == 0 and not said
    # a failure AFTER every rank's group worked is not retried
    said.clear()
    rc = mg.launch_local_ranks(2, [sys.executable, "-c", FIRST_CONTACT, "0", "9"], base_env=base, timeout=60,
                               rank_dir=str(tmp_path / "c"), log=said.append)
    assert rc == 9 and len(said) == 1 and not (tmp_path / "c" / "attempt1").exists()
    # neither mode works: two attempts, the failure is reported
    rc = mg.launch_local_ranks(2, [sys.executable, "-c", FIRST_CONTACT, "7"], base_env=base, timeout=60,
                               rank_dir=str(tmp_path / "d"), log=said.append)
    assert rc == 3 and (tmp_path / "d" / "attempt1" / "rank1.err").exists()


def test_resolve_world():
    a = bench.parse_args(["--gpus", "4"])
    assert bench.resolve_world(a, {}) == (0, 0, 4, True)                  # no RANK: bench.py launches the 4 ranks
    assert bench.resolve_world(a, {"RANK": "2", "LOCAL_RANK": "2", "WORLD_SIZE": "4"}) == (2, 2, 4, False)
    with pytest.raises(SystemExit):                                       # mislaunch: 1 rank, --gpus 4
        bench.resolve_world(a, {"RANK": "0", "WORLD_SIZE": "1"})
    a1 = bench.parse_args([])
    assert bench.resolve_world(a1, {}) == (0, 0, 1, False)
    with pytest.raises(SystemExit):
        bench.resolve_world(bench.parse_args(["--gpus", "0"]), {})


def test_bench_gpus_2_without_two_gpus_fails_loudly():
    """No GPU here: `bench.py --gpus 2` must exit non-zero and print no result line (never n_gpus: 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "n_gpus" not in p.stdout
    assert "cannot run here" in p.stderr or "rank failed" in p.stderr
