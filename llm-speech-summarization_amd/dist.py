"""Multi-GPU plumbing for the inference path: replicas only.

Each utterance is an independent encode -> prefill -> decode chain and a full bf16 replica uses < 3 % of
one MI355X's 288 GB, so ranks shard the utterance list and never exchange data on the path; the only
collectives are the throughput report's barrier / MAX-reduce / counter sum (RCCL on GPUs — backend
"nccl" is RCCL on ROCm — gloo in the CPU tests).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(lengths: Sequence[int], rank: int, world: int) -> List[int]:
    """Length-balanced round-robin: utterances sorted by length (longest first) are dealt to ranks in
    serpentine order, so every rank gets ~equal audio seconds and every index exactly one owner."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, slot = divmod(pos, world)
        owner = slot if rnd % 2 == 0 else world - 1 - slot
        if owner == rank:
            mine.append(idx)
    return sorted(mine)


def max_over_ranks(value: float, device) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values: Sequence[float], device) -> List[float]:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
