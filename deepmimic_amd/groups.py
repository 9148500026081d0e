"""A batch of envs as G independent groups, each its own `dm_ctx` on its own HIP stream (round 4).

Why: a 4096-env launch of the step kernel is ONE round of waves (2048 waves on 2048 slots), so it lasts as long as its slowest
wave while the slots of the fast waves idle ("the wave-time tail": mean wave 4.30 M cycles, slowest 4.91 M).  Two
half-batches on two streams drift apart in phase: when the fast waves of group A are done, the waves of group B that share their
SIMDs run uncontended, and A's next control step starts when A's own slowest wave is done, not the whole batch's.  Measured on one
MI355X (tools/gpu_ab_groups.py, profiles/r04_ab_groups.json): 2.095 M -> 2.236 M env-steps/s at G = 2 (+6.7 %); G >= 4 is slower (2.23 / 2.22 M at
G = 4 / 8 once HIP has enough hardware queues, GPU_MAX_HW_QUEUES >= 16; under the default 4 queues streams alias and serialise:
profiles/r04_bench_groups_by_hw_queues.jsonl), so 2 is the only useful value and the default.  Every kernel gains once the batch fills the
chip's 2048 wave slots (>= 4096 envs): dog3d_pace, whose 4096-wave launch is two ragged rounds, +15 % (0.797 -> 0.919 M), dribble_amp +9 %,
the one-character humanoid kernel +5 %; below 4096 envs a second group loses 1-4 % (profiles/r04_bench_env_sweep*.jsonl).

Nothing in the C-ABI changes: a group is a `dm_create` of its own with `env_id_offset` = its first global env id, which keys every
reset / goal / perturbation draw by the GLOBAL id -- env i's trajectory is the same in any grouping (tests/test_groups.py), exactly
as it is under the multi-GPU sharding of SURVEY 8(e).  A C / C++ host does the same with two contexts.

The reference has no analogue (one env per process, mpi_run.py:16-24); the protocol per group is that of `BatchEnv`.
A learner keeps the groups apart the way double-buffered samplers do: policy(A) -> step(A) on stream A while B is stepping.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .core import BatchEnv
from .model import SceneTables


def split_even(num_envs: int, groups: int) -> List[int]:
    """Group sizes: equal and EVEN (two characters share a wavefront: env 2b and 2b + 1 of a ctx), or one group if that cannot be had."""
    if groups <= 1 or num_envs % (2 * groups) != 0:
        return [num_envs]
    return [num_envs // groups] * groups


class EnvGroups:
    """G `BatchEnv`s over the contiguous global env ids [env_id_offset, env_id_offset + num_envs)."""

    def __init__(self, tables: SceneTables, num_envs: int, groups: int = 2, env_id_offset: int = 0, **kw):
        sizes = split_even(int(num_envs), int(groups))
        self.N, self.G = int(num_envs), len(sizes)
        self.first = [int(env_id_offset) + sum(sizes[:g]) for g in range(self.G)]     # first GLOBAL env id of each group
        self.start = [sum(sizes[:g]) for g in range(self.G)]                          # first row of each group in a whole-batch array
        self.count = sizes
        self.envs = [BatchEnv(tables, sizes[g], env_id_offset=self.first[g], **kw) for g in range(self.G)]
        e = self.envs[0]
        self.S, self.A, self.P, self.D, self.J, self.duration = e.S, e.A, e.P, e.D, e.J, e.duration
        self.amp_size = e.amp_size

    def rows(self, g: int) -> slice:
        return slice(self.start[g], self.start[g] + self.count[g])

    def reset(self, kin_times=None, max_times=None):
        for g, e in enumerate(self.envs):
            kt = None if kin_times is None else np.asarray(kin_times, dtype=np.float64)[self.rows(g)]
            mt = max_times if (max_times is None or np.isscalar(max_times)) else np.asarray(max_times, dtype=np.float64)[self.rows(g)]
            e.reset(kin_times=kt, max_times=mt)

    def set_streams(self, handles: Sequence[int]):
        """one external HIP stream per group (e.g. torch.cuda.Stream().cuda_stream); without it every group runs on its ctx's own stream"""
        assert len(handles) == self.G
        for e, h in zip(self.envs, handles):
            e.set_stream(int(h))

    def step_group_device(self, g: int, actions_ptr: int, states_ptr: int, rewards_ptr: int, term_ptr: int, valid_ptr: int, end_ptr: int, **kw):
        """Control step of group g, asynchronous on its stream.  The pointers are the WHOLE-BATCH device arrays (row = env id minus
        env_id_offset); the group reads / writes its own rows."""
        o, e = self.start[g], self.envs[g]
        off = lambda p, width: (p + 4 * o * width) if p else 0      # (every array of the step interface is 4-byte: float32 / int32, in both precisions)
        if kw.get("amp_ptr"):
            kw = dict(kw, amp_ptr=off(kw["amp_ptr"], e.amp_size))   # the AMP observation array is whole-batch too
        e.step_device(off(actions_ptr, e.A), off(states_ptr, e.S), off(rewards_ptr, 1), off(term_ptr, 1), off(valid_ptr, 1), off(end_ptr, 1), **kw)

    def step_device(self, actions_ptr: int, states_ptr: int, rewards_ptr: int, term_ptr: int, valid_ptr: int, end_ptr: int, **kw):
        for g in range(self.G):
            self.step_group_device(g, actions_ptr, states_ptr, rewards_ptr, term_ptr, valid_ptr, end_ptr, **kw)

    def step(self, actions=None, timestep: float = 1.0 / 600, n_updates: int = 20, **kw):
        """Host-array convenience (tests): every group's `BatchEnv.step`, results concatenated in env order."""
        outs = []
        for g, e in enumerate(self.envs):
            a = None if actions is None else np.asarray(actions, dtype=np.float32).reshape(self.N, self.A)[self.rows(g)]
            outs.append(e.step(a, timestep, n_updates, **kw))
        return {k: np.concatenate([o[k] for o in outs], axis=0) for k in outs[0]}

    def get_state(self):
        """`BatchEnv.get_state` of every group (each synchronises its own stream), rows concatenated in env order"""
        parts = [e.get_state() for e in self.envs]
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}

    def debug(self, name: str):
        """`BatchEnv.debug` of every group, rows concatenated in env order (e.g. "fallback": the per-env fallback-substep counters)"""
        return np.concatenate([e.debug(name) for e in self.envs], axis=0)

    def get_manifolds(self):
        return np.concatenate([e.get_manifolds() for e in self.envs], axis=0)

    def bench_rollout(self, warmup: int, steps: int, **kw) -> float:
        """Fixed-action rollout of every group, each from its own host thread through the C loop (`dm_bench_rollout`, ctypes drops the
        GIL), all released together; returns the wall-clock milliseconds from the common start to the last group's end."""
        import threading
        import time
        for e in self.envs:
            if warmup:
                e.bench_rollout(warmup, 0, **kw)
        gate = threading.Barrier(self.G + 1)

        def run(e):
            gate.wait()
            e.bench_rollout(0, steps, **kw)
        th = [threading.Thread(target=run, args=(e,)) for e in self.envs]
        for x in th:
            x.start()
        gate.wait(); t0 = time.perf_counter()
        for x in th:
            x.join()
        return 1e3 * (time.perf_counter() - t0)

    def synchronize(self):
        for e in self.envs:
            e.synchronize()

    def close(self):
        for e in self.envs:
            e.close()
