"""Sharding across the GPUs of one node: independent instances by rank (the headline
configuration), or ONE huge instance by point range (`point_range`, `sharded_msm`).

MSM instances are independent (the reference loops over them: gpu_profiler.rs:104-106,
benches/msm_benchmark.rs:29-34), so rank r simply owns a contiguous block of instances and there is NO
data-path collective; the only exchange is one all-gather of the 96-byte results (RCCL over xGMI when the
backend is "nccl", gloo on CPU in the tests).
"""
from __future__ import annotations

SEED_BASE = 0xB2540000
RESULT_BYTES = 96


def instance_ids(rank: int, world: int, per_rank: int):
    """Global instance numbers owned by `rank` (weak scaling: per_rank instances on every rank)."""
    if not (0 <= rank < world) or per_rank <= 0:
        raise ValueError("bad rank/world/per_rank")
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def instance_seed(global_instance: int) -> int:
    """Seed of the deterministic generator for a global instance number."""
    return SEED_BASE + global_instance


def rank_environments(nproc: int, master_port: int, base_env=None, master_addr: str = "127.0.0.1"):
    """The N child environments of a one-node launch, one per GPU: RANK / LOCAL_RANK / WORLD_SIZE /
    LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT on top of `base_env` (default: this process's environment) --
    what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would hand its workers."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    import os
    base = dict(os.environ if base_env is None else base_env)
    # HSA_ENABLE_IPC_MODE_LEGACY is INHERITED, never defaulted here: which setting a host's driver wants (dmabuf or
    # legacy IPC handles between the ranks' processes) is the operator's knowledge; launch_local_ranks tries the other
    # one once when the ranks cannot even set up their communicator
    envs = []
    for r in range(nproc):
        e = dict(base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                  "MASTER_ADDR": master_addr, "MASTER_PORT": str(master_port)})
        envs.append(e)
    return envs


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"


def _run_ranks(nproc, argv, envs, rank_dir, timeout):
    """One attempt: start the children, wait.  Returns (exit code, ranks that reported "started")."""
    import os
    import subprocess
    import time
    procs, errs = [], []
    for r, e in enumerate(envs):
        e = dict(e)
        e["MSM_AMD_RANK_STARTED_FILE"] = os.path.join(rank_dir, "rank%d.started" % r)
        f = open(os.path.join(rank_dir, "rank%d.err" % r), "wb")
        errs.append(f)
        procs.append(subprocess.Popen(list(argv), env=e, stderr=f))   # stdout: inherited (rank 0 prints the line)
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        pending = set(range(nproc))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:                       # exact PIDs we started, nothing by pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for f in errs:
            f.close()
    started = [r for r in range(nproc) if os.path.exists(os.path.join(rank_dir, "rank%d.started" % r))]
    return rc, started


def launch_local_ranks(nproc: int, argv, master_port: int = 0, base_env=None, timeout=None, rank_dir=None,
                       retry_other_ipc_mode=True, log=None) -> int:
    """Start `nproc` worker processes (`argv` = full command line, e.g. [sys.executable, "bench.py", ...]), one per
    GPU, and wait for them.  The caller must NOT have touched the GPU: workers are plain children (subprocess), the
    parent is never replaced.  Rank 0 inherits stdout (it prints the result line); every rank's stderr goes to
    `rank_dir`/rank<k>.err (a fresh temporary directory by default) and is replayed on the parent's stderr at the end,
    so a rank that dies in communicator setup leaves its RCCL / HIP error text behind.  Returns 0 only when EVERY
    rank exited 0; when one rank fails the others are terminated (a rank that died would leave the rest in a
    collective forever) and the first non-zero code is returned.

    First contact with a multi-GPU host: a worker touches the file named by MSM_AMD_RANK_STARTED_FILE once its
    process group works (bench.py: after the first barrier).  If a rank fails before every rank has done so, and
    `retry_other_ipc_mode` is set, the parent -- which never touched the GPU -- starts ONE fresh set of children with
    HSA_ENABLE_IPC_MODE_LEGACY flipped (dmabuf <-> legacy IPC handles: cross-process sharing fails with
    `hipIpcGetMemHandle: invalid argument` under the wrong one) and says which setting worked."""
    import os
    import sys
    import tempfile
    log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))
    base = dict(os.environ if base_env is None else base_env)
    attempts = [base]
    if retry_other_ipc_mode and nproc > 1:
        other = dict(base)
        other[IPC_VAR] = "1" if base.get(IPC_VAR, "") == "0" else "0"
        attempts.append(other)
    rc = 1
    for k, env in enumerate(attempts):
        d = tempfile.mkdtemp(prefix="msm_amd_ranks_") if rank_dir is None else os.path.join(rank_dir, "attempt%d" % k)
        os.makedirs(d, exist_ok=True)
        envs = rank_environments(nproc, master_port or free_port(), env)
        rc, started = _run_ranks(nproc, argv, envs, d, timeout)
        setting = "%s=%s" % (IPC_VAR, env.get(IPC_VAR, "<unset>"))
        for r in range(nproc):                # replay what the ranks wrote (rank 0 first), failures or not
            try:
                text = open(os.path.join(d, "rank%d.err" % r), "rb").read().decode("utf-8", "replace")
            except OSError:
                text = ""
            if text and (rc != 0 or r == 0):
                sys.stderr.write("".join("[rank %d] %s\n" % (r, ln) for ln in text.splitlines()[-60:]))
        if rc == 0:
            if k > 0:
                log("launch_local_ranks: the ranks came up with %s (the inherited setting failed in communicator setup)"
                    % setting)
            return 0
        log("launch_local_ranks: a rank failed (exit %d) with %s; ranks that had a working process group: %s; "
            "per-rank stderr in %s" % (rc, setting, started or "none", d))
        if len(started) == nproc:
            break                             # the failure came after communicator setup: the other IPC mode is no cure
    return rc


def all_gather_results(local_results, dist=None, device=None):
    """All-gather the per-instance 96-byte results of every rank; returns world*per_rank byte strings in
    global instance order.  `dist` is torch.distributed (already initialised) or None for one process."""
    blob = b"".join(local_results)
    if len(blob) != RESULT_BYTES * len(local_results):
        raise ValueError("every result must be 96 bytes")
    if dist is None:
        return list(local_results)
    import torch
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    world = dist.get_world_size()
    out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    raw = bytes(out.cpu().numpy().tobytes())
    return [raw[i * RESULT_BYTES:(i + 1) * RESULT_BYTES] for i in range(world * len(local_results))]


class ResultGatherer:
    """Per-step all-gather of the 96-byte results without a host synchronisation inside the step: staging and
    device tensors are allocated once, the copy to the device is asynchronous from pinned memory, the collective
    is enqueued on the backend's stream, and `fetch()` (one device->host copy) is only called when the caller
    wants to look at the gathered bytes.  The pinned staging buffer and its device twin exist DEPTH times and are
    used round-robin; before a slot's host bytes are overwritten the event recorded after its previous
    host->device copy is waited for, so a slow step can never gather the next step's bytes."""

    DEPTH = 4

    def __init__(self, dist, device, per_rank):
        import torch
        self.dist, self.device, self.per_rank = dist, device, per_rank
        self.world = dist.get_world_size() if dist is not None else 1
        nbytes = RESULT_BYTES * per_rank
        self.cuda = device is not None and getattr(device, "type", "cpu") == "cuda"
        dev = device if device is not None else torch.device("cpu")
        self.stage = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=self.cuda) for _ in range(self.DEPTH)]
        self.mine = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(self.DEPTH)]
        self.copied = [torch.cuda.Event() if self.cuda else None for _ in range(self.DEPTH)]
        self.used = [False] * self.DEPTH
        self.work = [None] * self.DEPTH
        self.all = torch.empty(self.world * nbytes, dtype=torch.uint8, device=dev)
        self.turn = 0
        self.steps = 0

    def gather(self, local_results):
        blob = b"".join(local_results)
        if len(blob) != RESULT_BYTES * self.per_rank:
            raise ValueError("expected %d results of 96 bytes" % self.per_rank)
        import torch
        k = self.turn
        self.turn = (k + 1) % self.DEPTH
        if self.work[k] is not None:
            self.work[k].wait()                   # the collective that last read this slot's device twin is ordered
            self.work[k] = None                   # before what follows on the current stream (no host block on RCCL)
        if self.cuda and self.used[k]:
            self.copied[k].synchronize()          # the copy that last read this pinned slot has finished
        self.stage[k].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        self.mine[k].copy_(self.stage[k], non_blocking=True)
        if self.cuda:
            self.copied[k].record()
            self.used[k] = True
        if self.dist is not None:
            self.work[k] = self.dist.all_gather_into_tensor(self.all, self.mine[k], async_op=True)
        else:
            self.all.copy_(self.mine[k])
        self.steps += 1

    def drain(self):
        """Order every outstanding collective before the current stream (call before reading `all`)."""
        for k in range(self.DEPTH):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None

    def fetch(self):
        self.drain()
        raw = bytes(self.all.cpu().numpy().tobytes())
        return [raw[i * RESULT_BYTES:(i + 1) * RESULT_BYTES] for i in range(self.world * self.per_rank)]


def point_range(rank: int, world: int, n: int):
    """[begin, end) of the points rank `rank` owns when ONE instance of n points is split across `world` GPUs
    (the same algebra as the reference's GPU + CPU split by point range, msm.rs:385-419)."""
    if not (0 <= rank < world) or n < 0:
        raise ValueError("bad rank/world/n")
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def sharded_msm(local_msm, sum_points, rank: int, world: int, n: int, dist=None, device=None):
    """One MSM over n points on `world` GPUs.  `local_msm(begin, end)` returns this rank's 96-byte partial result
    for its point range (e.g. `lambda b, e: cfg.msm_batch_device([ds + 32 * b], [dp + 64 * b], [e - b])[0]`);
    the partials are all-gathered (world x 96 bytes over RCCL: latency only, a point addition is not a reduce-op)
    and every rank adds them with `sum_points` (`msm_amd_sum_points`).  Ranks with an empty range contribute the
    identity (z = 0)."""
    begin, end = point_range(rank, world, n)
    partial = local_msm(begin, end) if end > begin else bytes(RESULT_BYTES)
    partials = all_gather_results([partial], dist, device)
    return sum_points(partials)
