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


def _call_with_timeout(fn, seconds: float, what: str, on_late=None):
    """Run fn() on a daemon thread; TimeoutError if it has not returned after `seconds`.  The thread is left behind; if fn() does
    return after the deadline, `on_late(result)` runs on that thread (the caller has moved on: whatever fn() made must be torn down
    there, not leaked)."""
    import threading
    box = {}
    lock = threading.Lock()

    def run():
        try:
            r = fn()
        except BaseException as e:     # noqa: BLE001 — re-raised on the caller's thread
            with lock:
                box["e"] = e
            return
        with lock:
            late = box.get("gave_up", False)
            if not late:
                box["r"] = r
        if late and on_late is not None:
            try:
                on_late(r)
            except Exception:          # noqa: BLE001 — nothing left to report to
                pass

    t = threading.Thread(target=run, daemon=True, name=what)
    t.start()
    t.join(seconds)
    with lock:
        if "e" in box:
            raise box["e"]
        if "r" in box or not t.is_alive():
            return box.get("r")
        box["gave_up"] = True
    raise TimeoutError(f"{what} did not return within {seconds:.0f} s")


def negotiate_comm(dist, group, device, make_id, init_rank, id_bytes: int, log=print):
    """The `sl` backend's start-up handshake, written so that EVERY rank issues the same collectives in the same order whatever
    fails where (a rank that skipped one would meet its peers in a different collective on the same process group: a hang):

      1. broadcast from rank 0 of `id_bytes + 1` bytes: the RCCL unique id and a status byte (0 = rank 0 could not draw an id);
      2. only if the status byte is 1: `init_rank(id, rank, world)` on every rank (ncclCommInitRank is itself a collective of the
         ranks that reach it; a rank that cannot times out inside `init_rank`);
      3. all_reduce(MIN) of one flag: every rank learns whether ALL communicators came up.

    `make_id() -> bytes` runs on rank 0 only, `init_rank(id_bytes, rank, world) -> handle` on every rank.  Returns (handle or None,
    all_ok).  The callables raise on failure; nothing raised here escapes before step 3 has run on this rank."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    msg = torch.zeros(id_bytes + 1, dtype=torch.uint8)
    if rank == 0:
        try:
            raw = bytes(make_id())
            assert len(raw) == id_bytes
            msg[:id_bytes] = torch.tensor(list(raw), dtype=torch.uint8)
            msg[id_bytes] = 1
        except Exception as e:     # noqa: BLE001 — reported, then decided collectively
            log(f"[dist] rank 0 could not draw a communicator id ({e}); the group falls back to torch.distributed", flush=True)
    if world > 1:
        dev_msg = msg.to(device)               # an nccl group moves device tensors, gloo host tensors
        dist.broadcast(dev_msg, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        msg = dev_msg.cpu()
    handle, ok = None, int(msg[id_bytes].item())
    if ok:
        try:
            handle = init_rank(bytes(msg[:id_bytes].tolist()), rank, world)
        except Exception as e:     # noqa: BLE001
            ok = 0
            log(f"[dist] communicator init failed on rank {rank} ({e}); asking the group to fall back to torch.distributed", flush=True)
    if world > 1:
        flag = torch.tensor([ok], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    return handle, bool(ok)


class BucketedAllReduce:
    """Sum-all-reduce of the gradient arena in buckets, launched as soon as a bucket is final so the exchange overlaps the
    rest of the backward pass (KD step).

    `ready(names)` marks buffers final; whenever the final PREFIX of the arena (in backward order) has grown by at least
    `min_bucket_bytes`, that contiguous slice is all-reduced in place.  On GPUs the collective is RCCL (`backend="nccl"`)
    issued on a side HIP stream behind an event recorded on the compute stream; xGMI is point-to-point, so buckets are kept
    large — a ring all-reduce is bound by one ~153 GB/s link whatever the bucket count, while tiny buckets only add launch
    latency.  On CPU tensors (gloo, used by the tests) the same code runs without streams.

    The owner calls `close()` when it is done (Trainer does): tearing an RCCL communicator down is a collective and is never
    run from `__del__`.
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
        self.requested_backend = backend
        if backend == "auto":
            use_sl = (arena.flat.is_cuda and dist.is_initialized() and dist.get_backend(group) == "nccl" and os.environ.get("SL_COMM_BACKEND", "sl") != "torch")
            self.backend = "sl" if use_sl else "torch"
            self.requested_backend = self.backend
        self.comm = None
        self.fell_back = False     # the `sl` backend was asked for (or chosen) and the group voted to use torch.distributed instead
        self.poisoned = False      # this rank gave up inside ncclCommInitRank: it will not try the `sl` backend again in this process
        # single_rank: issue the collectives even in a group of one (identity sums) — how tests/test_dp_gpu.py drives the
        # RCCL backend, its streams and events on a one-GPU box
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.real_world = self.world
        if single_rank and self.world == 1:    # (also without a process group: a one-rank communicator needs no rendezvous)
            self.world = 2 ** 30    # only ever compared with 1
        self.min_bytes = min_bucket_bytes
        self._final = [False] * len(arena.order)
        self._n_final = 0          # buffers [0, _n_final) are final
        self._sent = 0             # element offset up to which the arena has been handed to the collective
        self._pending = []
        self.cuda = arena.flat.is_cuda
        self.stream = torch.cuda.Stream(device=arena.flat.device) if (self.cuda and self.world > 1) else None
        # called as after_bucket(upto) behind every bucket launched on `stream` (GPU runs only): KDTrainer chains the AdamW of the summed
        # prefix there while the backward produces the next bucket
        self.after_bucket = None
        self.bucket_log: List[tuple] = []     # (start, end) element ranges of the buckets launched so far in this step
        self.last_buckets: List[tuple] = []   # ... of the previous optimizer step
        # exchange timing (events on the side stream around every bucket, read back in finish()): what bench.py's kd_step.comm reports
        self._ev: List[tuple] = []
        self._ev_compute_end = None
        self.last_exchange_ms = 0.0           # sum of the buckets' durations on the side stream, previous optimizer step
        self.last_exposed_ms = 0.0            # part of it that ran after the backward's last kernel (not overlapped)
        self.last_bucket_ms: List[float] = [] # each bucket's duration on the side stream, previous optimizer step (launch order)
        self.time_exchange = False
        if self.backend == "sl" and self.world > 1:
            self.comm, ok = self._negotiate()
            if not ok:                 # all ranks or none
                self._drop_comm()
                self.backend = "torch"
                self.fell_back = True

    # ---- the `sl` backend's communicator ----
    _poisoned_process = False          # a helper thread of this process is (or was) stuck inside ncclCommInitRank

    def _negotiate(self):
        """One RCCL communicator per process (= per GPU) through the C ABI: rank 0 draws the unique id (sl_comm_unique_id), the
        existing process group — torchrun's rendezvous — carries its 128 bytes + a status byte to the other ranks, every rank calls
        sl_comm_init, one MIN vote decides (negotiate_comm: the same collectives on every rank whatever fails).
        SL_COMM_TEST_FAIL=`unique_id` | `init:<rank>` makes that step fail (the fall-back tests)."""
        import ctypes as C
        from . import _lib as L
        fail = os.environ.get("SL_COMM_TEST_FAIL", "")
        device = self.arena.flat.device

        def make_id():
            if fail == "unique_id" or BucketedAllReduce._poisoned_process:
                raise RuntimeError("forced failure (SL_COMM_TEST_FAIL)" if fail else "this process gave up on an earlier sl_comm_init")
            buf = (C.c_ubyte * L.COMM_ID_BYTES)()
            L.check(L.lib().sl_comm_unique_id(buf), "sl_comm_unique_id")
            return bytes(buf)

        def init_rank(ident: bytes, rank: int, world: int):
            if fail == f"init:{rank}":
                raise RuntimeError("forced failure (SL_COMM_TEST_FAIL)")
            if BucketedAllReduce._poisoned_process:
                raise RuntimeError("this process gave up on an earlier sl_comm_init")
            raw = (C.c_ubyte * L.COMM_ID_BYTES)(*ident)

            def init():
                handle = C.c_void_p()
                with torch.cuda.device(device):                    # the current device is per thread
                    L.check(L.lib().sl_comm_init(C.byref(handle), raw, rank, world), "sl_comm_init")
                return handle

            def late(handle):                                      # the peers showed up after this rank voted no: do not leak it
                if handle is not None and handle.value:
                    with torch.cuda.device(device):
                        L.lib().sl_comm_abort(handle)

            # ncclCommInitRank is a collective: a rank that cannot reach its peers blocks inside it.  It runs on a helper thread so
            # that this rank can give up after SL_COMM_INIT_TIMEOUT_S (default 180 s) and take part in the vote instead.
            try:
                return _call_with_timeout(init, float(os.environ.get("SL_COMM_INIT_TIMEOUT_S", "180")), "sl_comm_init", on_late=late)
            except TimeoutError:
                self.poisoned = True
                BucketedAllReduce._poisoned_process = True
                raise

        return negotiate_comm(self.dist, self.group, device, make_id, init_rank, L.COMM_ID_BYTES)

    def _drop_comm(self) -> None:
        """After a fall-back vote: a communicator that did come up on this rank is aborted (ncclCommAbort is local; the collective
        ncclCommDestroy could wait for peers that never made theirs)."""
        if self.comm is not None:
            from . import _lib as L
            comm, self.comm = self.comm, None
            with torch.cuda.device(self.arena.flat.device):
                L.lib().sl_comm_abort(comm)

    def close(self) -> None:
        """Tear the communicator down (a collective: every rank's owner calls it, after its last finish())."""
        if self.comm is not None:
            from . import _lib as L
            comm, self.comm = self.comm, None
            torch.cuda.synchronize(self.arena.flat.device)
            L.check(L.lib().sl_comm_destroy(comm), "sl_comm_destroy")

    def abort(self) -> None:
        """Failure-path tear-down (an exception or KeyboardInterrupt on THIS rank while its peers may sit inside an all-reduce): local
        only — ncclCommAbort does not wait for the other ranks, nothing is synchronised, no collective is issued.  The owner re-raises
        afterwards so that the process exits non-zero and the launcher can end the job (close() in a `finally` made a
        failing rank block in synchronize / ncclCommDestroy, and torchrun never saw the failure)."""
        self._pending, self._ev = [], []
        if self.comm is not None:
            from . import _lib as L
            comm, self.comm = self.comm, None
            try:
                with torch.cuda.device(self.arena.flat.device):
                    L.lib().sl_comm_abort(comm)
            except Exception:          # noqa: BLE001 — already failing: nothing may mask the original error
                pass

    @staticmethod
    def wire_bytes(payload_bytes: int, world: int) -> dict:
        """Algorithmic bytes one rank SENDS for an in-place sum of `payload_bytes` over `world` ranks (= what it receives), by algorithm:
        a ring moves 2 (N-1)/N of the payload through ONE xGMI link direction (reduce-scatter then all-gather, N-1 steps each); the
        direct form (every rank owns 1/N, peers write their shares to the owner, the owner writes the sum back) moves the same
        2 (N-1)/N in total but spread over the N-1 links of the fully connected node, i.e. 2/N of the payload per link (SURVEY §5 /
        BASELINE.md §3: 14.6 ms vs 2.1 ms for the 1.274 GB arena at 8 ranks)."""
        n = max(1, int(world))
        total = 2 * (n - 1) * payload_bytes // n
        return {"per_rank_sent_total": int(total), "ring_per_link": int(total), "direct_per_link": int(total // max(1, n - 1))}

    def comm_info(self) -> dict:
        """What actually carries the exchange on this rank (bench.py's kd_step.comm; asserted by tests/test_dp_gpu.py)."""
        nranks = None
        if self.comm is not None:
            from . import _lib as L
            nranks = int(L.lib().sl_comm_world(self.comm))
        spans = self.last_buckets
        return {"backend": self.backend, "requested_backend": self.requested_backend, "fell_back": bool(self.fell_back),
                "rccl_nranks": nranks, "group_world": int(self.real_world),
                "group_backend": (self.dist.get_backend(self.group) if self.dist.is_initialized() else None),
                "buckets": len(spans), "bucket_bytes": [int((b - a) * 4) for a, b in spans],
                "bucket_ms": [round(float(v), 3) for v in self.last_bucket_ms],
                "algo_bytes_on_wire": self.wire_bytes(int(sum((b - a) * 4 for a, b in spans)), nranks if nranks else self.real_world),
                "measured_exchange_ms": round(float(self.last_exchange_ms), 3), "exposed_ms": round(float(self.last_exposed_ms), 3),
                "overlap_frac": (round(1.0 - self.last_exposed_ms / self.last_exchange_ms, 4) if self.last_exchange_ms > 0 else None)}

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
        timed = self.cuda and self.time_exchange
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)                         # the bucket's gradients are complete on the kernels' stream
            t0 = t1 = None
            if timed:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.stream)
            if self.comm is not None:
                from . import _lib as L
                with torch.cuda.device(self.arena.flat.device):
                    L.check(L.lib().sl_allreduce_sum(self.comm, chunk.data_ptr(), chunk.numel(), L.SL_F32, self.stream.cuda_stream), "sl_allreduce_sum")
                work = None                                    # ordered on the side stream; finish() joins the streams
            else:
                if self.backend == "sl":
                    raise RuntimeError("BucketedAllReduce: backend 'sl' without a communicator (closed?)")
                with torch.cuda.stream(self.stream):
                    work = self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                    if timed:
                        work.wait()                            # stream-ordered: makes t1 follow the collective on the side stream
                        work = None
            if timed:
                t1.record(self.stream)
                self._ev.append((t0, t1))
            if self.after_bucket is not None:
                if work is not None:
                    with torch.cuda.stream(self.stream):
                        work.wait()                            # stream-ordered: the hook may chain work on self.stream's tail
                    work = None
                self.after_bucket(upto)                        # the arena below `upto` is summed once self.stream reaches this point
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
        compute stream waits on the side stream; nothing is copied).  Returns the bucket count.  With `time_exchange` set (bench.py)
        the buckets' durations on the side stream are read back here (one host synchronisation per optimizer step)."""
        if self.world == 1:
            return 0
        timed = self.cuda and self.time_exchange and self.stream is not None
        if timed:
            self._ev_compute_end = torch.cuda.Event(enable_timing=True)
            self._ev_compute_end.record(torch.cuda.current_stream())     # the backward's last kernel
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
        if timed and self._ev:
            self._ev[-1][1].synchronize()
            self.last_bucket_ms = [float(a.elapsed_time(b)) for a, b in self._ev]
            self.last_exchange_ms = float(sum(self.last_bucket_ms))
            tail = float(self._ev_compute_end.elapsed_time(self._ev[-1][1]))
            self.last_exposed_ms = min(self.last_exchange_ms, max(0.0, tail))
        self._ev = []
        self._pending = []
        self._final = [False] * len(self._final)
        self._n_final, self._sent = 0, 0
        self.last_buckets, self.bucket_log = self.bucket_log, []
        return n
