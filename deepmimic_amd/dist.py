"""Env shards across the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Each env is an independent world (the reference runs one `cDeepMimicCore` per MPI worker, mpi_run.py:16-24), so the
hot path has no exchange step: rank r owns the contiguous global env ids [r*n, (r+1)*n) and its reset RNG streams are
keyed by the global id (`env_id_offset`), which makes every trajectory independent of the partition.  The only
collective is the per-control-step all-gather of the learner record {state[S], reward, terminate}.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from .core import BatchEnv


def shard_range(num_envs_total: int, world: int, rank: int):
    """Contiguous split of [0, N): (first global env id, count) of `rank`; earlier ranks take the remainder."""
    base, rem = divmod(num_envs_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


class ShardedEnv:
    """One shard of a node-wide batch of imitate envs + the record gather."""

    def __init__(self, tables, num_envs_total: int, rank: Optional[int] = None, world: Optional[int] = None,
                 device_id: Optional[int] = None, seed: int = 0, **kw):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.first, self.count = shard_range(num_envs_total, self.world, self.rank)
        self.total = num_envs_total
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device_id is None else device_id
        self.env = BatchEnv(tables, self.count, device_id=dev, seed=seed, env_id_offset=self.first, **kw)
        self.S, self.A = self.env.S, self.env.A

    def record_width(self):
        return self.S + 2

    def pack_record(self, out):
        """[n x (S+2)] float32: state | reward | terminate, the per-env record the learner consumes each control step."""
        rec = np.empty((self.count, self.S + 2), np.float32)
        rec[:, :self.S] = out["state"]; rec[:, self.S] = out["reward"]; rec[:, self.S + 1] = out["terminate"]
        return rec

    def gather(self, rec_local):
        """All-gather the shard records in global env order.  torch tensor in (CPU for gloo, device for nccl) -> tensor out.
        With uneven shards the result lives in a buffer that the next gather() overwrites."""
        import torch
        import torch.distributed as dist
        t = rec_local if isinstance(rec_local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(rec_local))
        if self.world == 1:
            return t
        base, rem = divmod(self.total, self.world)
        if rem == 0:
            out = torch.empty((self.total, t.shape[1]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t.contiguous())
            return out
        # uneven split: pad every shard to the largest one (collectives want equal sizes), compact after the gather.  The three
        # buffers are allocated once per (shape, dtype, device) and reused every control step.
        big = base + 1
        key = (t.shape[1], t.dtype, t.device)
        if getattr(self, "_uneven_key", None) != key:
            self._uneven_key = key
            self._pad = torch.zeros((big, t.shape[1]), dtype=t.dtype, device=t.device)
            self._gathered = torch.empty((self.world * big, t.shape[1]), dtype=t.dtype, device=t.device)
            self._compact = torch.empty((self.total, t.shape[1]), dtype=t.dtype, device=t.device)
        self._pad[:t.shape[0]].copy_(t)
        dist.all_gather_into_tensor(self._gathered, self._pad)
        for r in range(self.world):
            f, c = shard_range(self.total, self.world, r)
            self._compact[f:f + c].copy_(self._gathered[r * big:r * big + c])
        return self._compact


class RecordExchange:
    """Double-buffered, copy-free all-gather of the learner record, overlapped with the next control step.

    Each rank owns `depth` flat float32 buffers laid out [states n*S | rewards n | terminate n (int32 bits)]; the step
    kernel writes its outputs straight into the views of buffer k % depth (`dm_step_batch` takes raw device pointers), so
    there is no pack kernel.  `launch(slot)` issues one `all_gather_into_tensor` of the flat buffer with `async_op=True`:
    on the nccl (= RCCL) backend the collective is ordered after everything already enqueued on the current stream and
    runs on the process group's own stream, i.e. concurrently with the step kernel of control step k+1 that the caller
    enqueues next; `result(slot)` makes the current stream wait for it.  xGMI is point-to-point: the 8 shards of 3.75 MB
    each travel on their own links, there is no ring to bound.  On gloo (CPU tests) the same calls run synchronously.
    """

    def __init__(self, n: int, S: int, world: int, device, depth: int = 2, env: Optional[BatchEnv] = None):
        import torch
        self.n, self.S, self.world, self.depth = n, S, world, depth
        # The collective is ordered after the work on torch's CURRENT stream; the step kernel that fills the buffer must run on
        # that stream too.  A BatchEnv runs on its own non-blocking stream unless told otherwise, so pass it here (or call
        # env.set_stream(torch.cuda.current_stream().cuda_stream) yourself): without it the gather races the step kernel.
        if env is not None and torch.device(device).type == "cuda":
            env.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.chunk = n * S + 2 * n
        self.local = [torch.zeros(self.chunk, dtype=torch.float32, device=device) for _ in range(depth)]
        self.all = [torch.zeros(world * self.chunk, dtype=torch.float32, device=device) for _ in range(depth)]
        self.work = [None] * depth

    def views(self, slot: int):
        """(states [n,S] f32, rewards [n] f32, terminate [n] i32) aliasing local buffer `slot`."""
        import torch
        b = self.local[slot]
        n, S = self.n, self.S
        return b[:n * S].view(n, S), b[n * S:n * S + n], b[n * S + n:].view(torch.int32)

    def begin(self, slot: int):
        """Call before enqueueing the step that writes buffer `slot`: orders it after the gather last launched on that
        slot (control step k - depth), then returns the output views."""
        self.wait(slot)
        return self.views(slot)

    def launch(self, slot: int):
        """Call right after enqueueing the step that wrote buffer `slot` on the current stream."""
        import torch.distributed as dist
        if not dist.is_initialized():
            if self.world != 1:
                raise RuntimeError("RecordExchange with world > 1 needs an initialised torch.distributed process group")
            self.all[slot].copy_(self.local[slot])
            return
        self.work[slot] = dist.all_gather_into_tensor(self.all[slot], self.local[slot], async_op=True)

    def wait(self, slot: int):
        """Block the current stream (not the host, on nccl) until the gather last launched on `slot` is complete."""
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None

    def result(self, slot: int):
        """(states [world,n,S], rewards [world,n], terminate [world,n] i32): global env id = rank * n + i."""
        import torch
        self.wait(slot)
        a = self.all[slot].view(self.world, self.chunk)
        n, S = self.n, self.S
        return a[:, :n * S].unflatten(1, (n, S)), a[:, n * S:n * S + n], a[:, n * S + n:].view(torch.int32)


class CabiRecordExchange:
    """The same double-buffered exchange through the C-ABI (dm_comm_* / dm_gather_records, include/dm_hip.h): RCCL is driven by
    libdm_hip.so on its own stream, ordered against the env's stream by HIP events -- no torch collective on the data path.
    torch is used for the device buffers and, when a process group exists, to ship rank 0's unique id (the bootstrap a C++ host
    would do with MPI_Bcast)."""

    def __init__(self, env: BatchEnv, world: int, rank: int, device, depth: int = 2, force_rccl: bool = False):
        import torch
        from .core import Comm
        assert 1 <= depth <= 4
        self.env, self.world, self.rank, self.depth = env, world, rank, depth
        self.n, self.S = env.N, env.S
        self.chunk = self.n * self.S + 2 * self.n
        self.local = [torch.zeros(self.chunk, dtype=torch.float32, device=device) for _ in range(depth)]
        self.all = [torch.zeros(world * self.chunk, dtype=torch.float32, device=device) for _ in range(depth)]
        uid = None
        if world > 1 or force_rccl:
            import torch.distributed as dist
            box = [Comm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            uid = box[0]
        self.comm = Comm(world, rank, device_id=torch.device(device).index or 0, unique_id=uid)

    def views(self, slot: int):
        import torch
        b = self.local[slot]
        n, S = self.n, self.S
        return b[:n * S].view(n, S), b[n * S:n * S + n], b[n * S + n:].view(torch.int32)

    def begin(self, slot: int):
        self.comm.wait(self.env, slot)        # the env stream waits for the gather of step k - depth before overwriting the buffer
        return self.views(slot)

    def launch(self, slot: int):
        self.comm.gather(self.env, slot, self.local[slot].data_ptr(), self.all[slot].data_ptr(), self.chunk)

    def wait(self, slot: int):
        self.comm.wait(self.env, slot)

    def result(self, slot: int):
        import torch
        self.wait(slot)
        self.env.synchronize()
        a = self.all[slot].view(self.world, self.chunk)
        n, S = self.n, self.S
        return a[:, :n * S].unflatten(1, (n, S)), a[:, n * S:n * S + n], a[:, n * S + n:].view(torch.int32)


class _DeviceArray:
    """a raw device allocation as a `__cuda_array_interface__` object (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr: int, n: int, typestr: str = "<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


def all_reduce_normalizer(norm, device="cpu", group=None):
    """`MPIUtil.reduce_sum` of learning/normalizer.py:48-50 for a `DeviceNormalizer`: the pending sums {new_count, new_sum, new_sum_sq} of every rank are
    summed IN PLACE (RCCL on a GPU, gloo on the CPU test harness); `norm.update()` afterwards folds the same totals on every rank, so the ranks'
    statistics stay identical (the reference's check_synced).  Ordered on torch's current stream: keep the normaliser on that stream
    (`norm.set_stream(torch.cuda.current_stream().cuda_stream)`; the default, the null stream, is torch's default stream)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    ptr, n = norm.pending_ptr()
    dev = torch.device(device)
    if dev.type == "cuda":
        t = torch.as_tensor(_DeviceArray(ptr, n), device=dev)
    else:
        t = torch.from_numpy(np.ctypeslib.as_array((C.c_double * n).from_address(ptr)))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
