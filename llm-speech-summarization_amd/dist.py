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


class GradArena:
    """All fp32 gradient buffers of one optimizer step as views into ONE flat allocation, laid out in the order the backward
    pass finishes them (projector, layer N-1 ... 0, positional conv / feature projection, conv stack).  A bucket of the
    all-reduce is then a contiguous slice of `flat`: the collective runs in place on it, no gather copy before and no scatter
    copy after (the round-1 reducer moved 2 x 1.27 GB per optimizer step through torch.cat / copy_)."""

    ALIGN = 64   # elements: every view starts on a 256-byte boundary

    def __init__(self, named_shapes, device):
        self.order: List[str] = []
        self.span = {}
        off = 0
        for name, shape in named_shapes:
            n = 1
            for d in shape:
                n *= int(d)
            self.order.append(name)
            self.span[name] = (off, n, tuple(int(d) for d in shape))
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.views = {name: self.flat[o:o + n].view(shape) for name, (o, n, shape) in self.span.items()}
        self.index = {name: i for i, name in enumerate(self.order)}

    def end_offset(self, i: int) -> int:
        """Offset just past the i-th buffer's padded slot (== start of buffer i+1)."""
        if i + 1 < len(self.order):
            return self.span[self.order[i + 1]][0]
        return self.flat.numel()

    def zero_(self):
        self.flat.zero_()


def _call_with_timeout(fn, seconds: float, what: str):
    """Run fn() on a daemon thread; TimeoutError if it has not returned after `seconds` (the thread is left behind)."""
    import threading
    box = {}

    def run():
        try:
            box["r"] = fn()
        except BaseException as e:     # noqa: BLE001 — re-raised on the caller's thread
            box["e"] = e

    t = threading.Thread(target=run, daemon=True, name=what)
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise TimeoutError(f"{what} did not return within {seconds:.0f} s")
    if "e" in box:
        raise box["e"]
    return box.get("r")


class BucketedAllReduce:
    """Sum-all-reduce of the gradient arena in buckets, launched as soon as a bucket is final so the exchange overlaps the
    rest of the backward pass (KD step).

    `ready(names)` marks buffers final; whenever the final PREFIX of the arena (in backward order) has grown by at least
    `min_bucket_bytes`, that contiguous slice is all-reduced in place.  On GPUs the collective is RCCL (`backend="nccl"`)
    issued on a side HIP stream behind an event recorded on the compute stream; xGMI is point-to-point, so buckets are kept
    large — a ring all-reduce is bound by one ~153 GB/s link whatever the bucket count, while tiny buckets only add launch
    latency.  On CPU tensors (gloo, used by the tests) the same code runs without streams.
    """

    def __init__(self, arena: GradArena, group=None, min_bucket_bytes: int = 32 << 20, single_rank: bool = False, backend: str = "auto"):
        import torch.distributed as dist
        self.dist = dist
        self.arena = arena
        self.group = group
        # backend "sl": the collective is the library's own RCCL binding (sl_comm_init / sl_allreduce_sum, include/speechllm.h group 11),
        # launched on the side stream on a slice pointer of the arena; "torch": torch.distributed's all_reduce on the same slice (gloo on
        # CPU tensors — the tests here — or nccl).  "auto" = "sl" for device arenas in an nccl process group, unless SL_COMM_BACKEND=torch.
        self.backend = backend
        if backend == "auto":
            use_sl = (arena.flat.is_cuda and dist.is_initialized() and dist.get_backend(group) == "nccl" and os.environ.get("SL_COMM_BACKEND", "sl") != "torch")
            self.backend = "sl" if use_sl else "torch"
        self.comm = None
        # single_rank: issue the collectives even in a group of one (identity sums) — how tests/test_dp_gpu.py drives the
        # RCCL backend, its streams and events on a one-GPU box
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if single_rank and dist.is_initialized() and self.world == 1:
            self.world = 2 ** 30    # only ever compared with 1
        self.min_bytes = min_bucket_bytes
        self._final = [False] * len(arena.order)
        self._n_final = 0          # buffers [0, _n_final) are final
        self._sent = 0             # element offset up to which the arena has been handed to the collective
        self._pending = []
        self.cuda = arena.flat.is_cuda
        self.stream = torch.cuda.Stream(device=arena.flat.device) if (self.cuda and self.world > 1) else None
        self.bucket_log: List[tuple] = []     # (start, end) element ranges of the buckets launched so far in this step
        self.last_buckets: List[tuple] = []   # ... of the previous optimizer step
        if self.backend == "sl" and self.world > 1:
            # all ranks or none: a rank whose communicator did not come up would otherwise wait in a different collective than the rest
            ok = 1
            try:
                self._init_sl_comm()
            except Exception as e:     # noqa: BLE001 — reported, then decided collectively
                ok = 0
                print(f"[dist] sl_comm_init failed on this rank ({e}); asking the group to fall back to torch.distributed", flush=True)
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                flag = torch.tensor([ok], device=arena.flat.device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if not ok:
                self.close()
                self.backend = "torch"

    def _init_sl_comm(self) -> None:
        """One RCCL communicator per process (= per GPU) through the C ABI: rank 0 draws the unique id (sl_comm_unique_id), the
        existing process group — torchrun's rendezvous — carries its 128 bytes to the other ranks, every rank calls sl_comm_init."""
        import ctypes as C
        from . import _lib as L
        dist = self.dist
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        ident = torch.zeros(L.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_ubyte * L.COMM_ID_BYTES)()
            L.check(L.lib().sl_comm_unique_id(buf), "sl_comm_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            dev_id = ident.to(self.arena.flat.device)          # an nccl group moves device tensors
            dist.broadcast(dev_id, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            ident = dev_id.cpu()
        raw = (C.c_ubyte * L.COMM_ID_BYTES)(*ident.tolist())
        handle = C.c_void_p()

        def init():
            with torch.cuda.device(self.arena.flat.device):    # the current device is per thread
                L.check(L.lib().sl_comm_init(C.byref(handle), raw, rank, world), "sl_comm_init")

        # ncclCommInitRank is a collective: a rank that cannot reach its peers blocks inside it.  It runs on a helper thread so that
        # this rank can give up after SL_COMM_INIT_TIMEOUT_S (default 180 s) and take part in the group's fall-back vote instead.
        _call_with_timeout(init, float(os.environ.get("SL_COMM_INIT_TIMEOUT_S", "180")), "sl_comm_init")
        self.comm = handle

    def close(self) -> None:
        if self.comm is not None:
            from . import _lib as L
            torch.cuda.synchronize(self.arena.flat.device)
            L.check(L.lib().sl_comm_destroy(self.comm), "sl_comm_destroy")
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ready(self, names: Sequence[str]) -> None:
        """Mark gradient buffers as final for this optimizer step; launches a bucket once the final prefix is large enough."""
        if self.world == 1:
            return
        for n in names:
            self._final[self.arena.index[n]] = True
        while self._n_final < len(self._final) and self._final[self._n_final]:
            self._n_final += 1
        upto = self.arena.end_offset(self._n_final - 1) if self._n_final > 0 else 0
        if (upto - self._sent) * 4 >= self.min_bytes:
            self._launch(upto)

    def _launch(self, upto: int) -> None:
        if upto <= self._sent:
            return
        chunk = self.arena.flat[self._sent:upto]
        self.bucket_log.append((self._sent, upto))
        self._sent = upto
        if self.cuda and self.comm is not None:
            from . import _lib as L
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)                         # the bucket's gradients are complete on the kernels' stream
            with torch.cuda.device(self.arena.flat.device):
                L.check(L.lib().sl_allreduce_sum(self.comm, chunk.data_ptr(), chunk.numel(), L.SL_F32, self.stream.cuda_stream), "sl_allreduce_sum")
            work = None                                        # ordered on the side stream; finish() joins the streams
        elif self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                work = self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(work)

    def flush(self) -> None:
        """Launch whatever part of the final prefix has not been sent yet."""
        if self.world == 1:
            return
        self._launch(self.arena.end_offset(self._n_final - 1) if self._n_final > 0 else 0)

    def finish(self) -> int:
        """Called after the backward pass: everything is final.  Sends the rest of the arena, waits for every bucket (the
        compute stream waits on the side stream; nothing is copied).  Returns the bucket count."""
        if self.world == 1:
            return 0
        self._launch(self.arena.flat.numel())
        n = len(self._pending)
        if self.cuda and self.stream is not None:
            with torch.cuda.stream(self.stream):
                for work in self._pending:
                    if work is not None:
                        work.wait()
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for work in self._pending:
                work.wait()
        self._pending = []
        self._final = [False] * len(self._final)
        self._n_final, self._sent = 0, 0
        self.last_buckets, self.bucket_log = self.bucket_log, []
        return n
