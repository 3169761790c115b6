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
    wants to look at the gathered bytes."""

    def __init__(self, dist, device, per_rank):
        import torch
        self.dist, self.device, self.per_rank = dist, device, per_rank
        self.world = dist.get_world_size() if dist is not None else 1
        nbytes = RESULT_BYTES * per_rank
        pin = device is not None and getattr(device, "type", "cpu") == "cuda"
        self.stage = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pin)
        dev = device if device is not None else torch.device("cpu")
        self.mine = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.all = torch.empty(self.world * nbytes, dtype=torch.uint8, device=dev)

    def gather(self, local_results):
        blob = b"".join(local_results)
        if len(blob) != RESULT_BYTES * self.per_rank:
            raise ValueError("expected %d results of 96 bytes" % self.per_rank)
        import torch
        self.stage.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        self.mine.copy_(self.stage, non_blocking=True)
        if self.dist is not None:
            self.dist.all_gather_into_tensor(self.all, self.mine)
        else:
            self.all.copy_(self.mine)

    def fetch(self):
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
