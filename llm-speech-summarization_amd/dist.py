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


class BucketedAllReduce:
    """Sum-all-reduce of named fp32 gradient buffers in buckets, launched as soon as a bucket is final so the
    exchange overlaps the rest of the backward pass (KD step).

    On GPUs the collective is RCCL (`backend="nccl"`) issued on a side HIP stream behind an event recorded on
    the compute stream; xGMI is point-to-point, so buckets are kept large (>= `min_bucket_bytes`) — a ring
    all-reduce is bound by one ~153 GB/s link whatever the bucket count, while tiny buckets only add launch
    latency.  On CPU tensors (gloo, used by the tests) the same code runs without streams.
    """

    def __init__(self, grads, group=None, min_bucket_bytes: int = 32 << 20):
        import torch.distributed as dist
        self.dist = dist
        self.grads = grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.min_bytes = min_bucket_bytes
        self._names: List[str] = []
        self._bytes = 0
        self._pending = []
        any_t = next(iter(grads.values()))
        self.cuda = any_t.is_cuda
        self.stream = torch.cuda.Stream(device=any_t.device) if (self.cuda and self.world > 1) else None

    def ready(self, names: Sequence[str]) -> None:
        """Mark gradient buffers as final for this optimizer step; flushes a bucket once it is large enough."""
        if self.world == 1:
            return
        for n in names:
            self._names.append(n)
            self._bytes += self.grads[n].numel() * 4
        if self._bytes >= self.min_bytes:
            self.flush()

    def flush(self) -> None:
        if self.world == 1 or not self._names:
            return
        names, self._names, self._bytes = self._names, [], 0
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                flat = torch.cat([self.grads[n].reshape(-1) for n in names])
                work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            flat = torch.cat([self.grads[n].reshape(-1) for n in names])
            work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat, names))

    def finish(self) -> int:
        """Wait for every bucket, scatter the sums back into the gradient buffers.  Returns the bucket count."""
        self.flush()
        n = len(self._pending)

        def scatter():
            for work, flat, names in self._pending:
                work.wait()
                off = 0
                for name in names:
                    g = self.grads[name]
                    g.copy_(flat[off:off + g.numel()].view_as(g))
                    off += g.numel()

        if self.cuda and self.stream is not None:
            with torch.cuda.stream(self.stream):
                scatter()
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            scatter()
        self._pending = []
        return n
