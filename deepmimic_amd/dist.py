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
        """All-gather the shard records in global env order.  torch tensor in (CPU for gloo, device for nccl) -> tensor out."""
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
        # uneven split: pad every shard to the largest one (collectives want equal sizes), trim after the gather
        big = base + 1
        pad = torch.zeros((big, t.shape[1]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        out = torch.empty((self.world * big, t.shape[1]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad)
        return torch.cat([out[r * big:r * big + shard_range(self.total, self.world, r)[1]] for r in range(self.world)], 0)


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

    def __init__(self, n: int, S: int, world: int, device, depth: int = 2):
        import torch
        self.n, self.S, self.world, self.depth = n, S, world, depth
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
