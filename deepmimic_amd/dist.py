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
