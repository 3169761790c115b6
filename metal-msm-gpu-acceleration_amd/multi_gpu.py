"""Instance sharding across the GPUs of one node.

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
    if dist is None or dist.get_world_size() == 1:
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
