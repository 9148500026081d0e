"""Device-resident vectorised env on top of the C-ABI (deepmimic_amd/core.py): observations, rewards and flags stay in HBM as
torch tensors, one kernel launch per 30 Hz control step, asynchronous on torch's current stream (whatever is current at each call) -- the way a GPU learner consumes
the path (the facade in compat/ is the drop-in for the reference's one-env-per-process driver; this is the batched form of the same
protocol: SetAction ; 20 x Update ; RecordState / CalcReward / CheckTerminate ; Reset of finished episodes, DeepMimic.py:62-80).

torch is plumbing here (device memory, streams); the stepping itself is libdm_hip.so.  From 4096 envs per GPU on, `TorchVecEnvGroups` (two env groups on their
own streams, below) is the faster form of the same thing: +2 ... 15 % depending on the scene (profiles/r04_bench_scenes_groups.json).

`--timer_type exp` scenes are served like the uniform timer since round 4: the in-kernel auto-reset draws min(min + Exp, max) (util/Timer.cpp:64-67)."""
from __future__ import annotations

from typing import Dict, Tuple

from .core import BatchEnv
from .model import SceneTables


class TorchVecEnv:
    def __init__(self, tables: SceneTables, num_envs: int, device: str = "cuda:0", seed: int = 0, timestep: float = 1.0 / 600,
                 updates_per_step: int = 20, amp_obs: bool = False, **env_kwargs):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("TorchVecEnv needs a GPU device (deepmimic_amd has no CPU path)")
        with torch.cuda.device(self.device):
            torch.zeros(1, device=self.device)                     # torch's context first (one HIP runtime in the process)
            self.env = BatchEnv(tables, num_envs, device_id=self.device.index or 0, seed=seed, **env_kwargs)
        self.n, self.obs_dim, self.act_dim, self.goal_dim = self.env.N, self.env.S, self.env.A, self.env.G
        self.timestep, self.updates = float(timestep), int(updates_per_step)
        f32, i32 = dict(dtype=torch.float32, device=self.device), dict(dtype=torch.int32, device=self.device)
        self.obs = torch.zeros((self.n, self.obs_dim), **f32); self.reward = torch.zeros(self.n, **f32)
        self.terminate = torch.zeros(self.n, **i32); self.valid = torch.zeros(self.n, **i32); self.episode_end = torch.zeros(self.n, **i32)
        self.amp_obs = torch.zeros((self.n, self.env.amp_size), **f32) if (amp_obs and self.env.amp_size) else None
        self.goal = torch.zeros((self.n, self.goal_dim), **f32) if self.goal_dim else None      # RecordGoal of the last step (goal scenes), device resident
        # the launches go to the caller's CURRENT torch stream (looked up at every call): ordered against the caller's work on both sides
        # without events (torch's default stream has the null handle; BatchEnv.set_stream maps it to the legacy default stream)
        self._stream_handle = None

    def _enter(self):
        h = int(self.torch.cuda.current_stream(self.device).cuda_stream)
        if h != self._stream_handle:
            self.env.set_stream(h); self._stream_handle = h

    def _leave(self):
        pass

    def _launch(self, actions_ptr, n_updates, auto_reset):
        self.env.step_device(actions_ptr, self.obs.data_ptr(), self.reward.data_ptr(), self.terminate.data_ptr(), self.valid.data_ptr(),
                             self.episode_end.data_ptr(), timestep=self.timestep, n_updates=n_updates, auto_reset=auto_reset,
                             amp_ptr=self.amp_obs.data_ptr() if self.amp_obs is not None else 0)

    def reset(self):
        """Reset every env (clip time, episode length, clip and yaw drawn on the device) and return the first observations."""
        self._enter()
        self.env.reset()
        self._launch(0, 0, False)                                   # RecordState of the reset state, no update
        self._leave()
        return self.obs

    def step(self, actions) -> Tuple["object", "object", "object", Dict[str, "object"]]:
        """One control step for every env.  `actions`: (N, A) float32 on this device.  Envs whose episode ended during the step are
        reset inside the launch; their `obs` row is already the first observation of the next episode, `reward` / `terminate` describe
        the step that ended (eTerminateNull 0 / Fail 1 / Succ 2; episode_end also covers the episode timer).
        `done` = every env that was reset inside the launch: episode_end (cDeepMimicCore::IsEpisodeEnd) OR an INVALID episode (valid == 0:
        cSceneSimChar::CheckValidEpisode failed, a link velocity beyond 100 -- the reference's driver ends and DISCARDS such an episode,
        DeepMimic.py:62-80 / learning/rl_agent.py end_episode; info["valid"] tells the two apart).  A learner that bootstraps across a row with
        done == False can therefore never stitch two episodes together."""
        t = self.torch
        if actions.device != self.device or actions.dtype != t.float32 or tuple(actions.shape) != (self.n, self.act_dim) or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous float32 (N, A) tensor on %s" % self.device)
        self._enter()
        self._launch(actions.data_ptr(), self.updates, True)
        self._leave()
        info = {"terminate": self.terminate, "valid": self.valid}
        if self.amp_obs is not None:
            info["amp_obs"] = self.amp_obs
        if self.goal_dim:
            self.env.last_goals_device(self.goal.data_ptr())        # device-to-device on the same stream, behind the step kernel: no host sync
            info["goal"] = self.goal
        return self.obs, self.reward, (self.episode_end != 0) | (self.valid == 0), info

    def close(self):
        self.env.close()


class TorchVecEnvGroups:
    """`TorchVecEnv` as G env groups on their own streams (deepmimic_amd/groups.py; round 4): the batch's tensors are shared, group g owns the rows
    `rows(g)`, and its control step runs on `stream(g)` -- a learner runs policy(g) on that stream right before `step_group(g, ...)` and works on
    another group meanwhile, the way double-buffered samplers do; the groups then drift apart in phase and fill each other's wave-time tail
    (closed loop with the on-device policy, 4096 humanoids: 2.14 M env-steps/s against 2.05 M with one launch per step, profiles/r04_policy_bench.json).
    `step(actions)` is the synchronous convenience: all groups, then the caller's current stream waits for them.  Env i follows the same trajectory
    as in a `TorchVecEnv` of the whole batch (draws are keyed by the global env id; tests/test_vec_env.py)."""

    def __init__(self, tables: SceneTables, num_envs: int, groups: int = 2, device: str = "cuda:0", seed: int = 0, timestep: float = 1.0 / 600,
                 updates_per_step: int = 20, amp_obs: bool = False, **env_kwargs):
        import torch
        from .groups import EnvGroups
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("TorchVecEnvGroups needs a GPU device (deepmimic_amd has no CPU path)")
        with torch.cuda.device(self.device):
            torch.zeros(1, device=self.device)
            self.g = EnvGroups(tables, num_envs, groups=groups, device_id=self.device.index or 0, seed=seed, **env_kwargs)
        self.G, self.n, self.obs_dim, self.act_dim = self.g.G, self.g.N, self.g.S, self.g.A
        self.goal_dim = self.g.envs[0].G
        self.timestep, self.updates = float(timestep), int(updates_per_step)
        f32, i32 = dict(dtype=torch.float32, device=self.device), dict(dtype=torch.int32, device=self.device)
        self.obs = torch.zeros((self.n, self.obs_dim), **f32); self.reward = torch.zeros(self.n, **f32)
        self.terminate = torch.zeros(self.n, **i32); self.valid = torch.zeros(self.n, **i32); self.episode_end = torch.zeros(self.n, **i32)
        self.amp_obs = torch.zeros((self.n, self.g.amp_size), **f32) if (amp_obs and self.g.amp_size) else None
        self.goal = torch.zeros((self.n, self.goal_dim), **f32) if self.goal_dim else None
        # the contexts' own streams (created back to back: distinct hardware queues), visible to torch as external streams
        self.streams = [torch.cuda.ExternalStream(e.own_stream(), device=self.device) for e in self.g.envs]

    def rows(self, g: int) -> slice:
        return self.g.rows(g)

    def stream(self, g: int):
        return self.streams[g]

    def _launch(self, g, actions_ptr, n_updates, auto_reset):
        self.g.step_group_device(g, actions_ptr, self.obs.data_ptr(), self.reward.data_ptr(), self.terminate.data_ptr(), self.valid.data_ptr(),
                                 self.episode_end.data_ptr(), timestep=self.timestep, n_updates=n_updates, auto_reset=auto_reset,
                                 amp_ptr=self.amp_obs.data_ptr() if self.amp_obs is not None else 0)
        if self.goal is not None and n_updates:
            self.g.envs[g].last_goals_device(self.goal[self.rows(g)].data_ptr())      # device-to-device on the group's stream, behind its kernel

    def _info(self, r):
        info = {"terminate": self.terminate[r], "valid": self.valid[r]}
        if self.amp_obs is not None:
            info["amp_obs"] = self.amp_obs[r]
        if self.goal is not None:
            info["goal"] = self.goal[r]
        return info

    def reset(self):
        self.g.reset()
        cur = self.torch.cuda.current_stream(self.device)
        for g in range(self.G):
            self.streams[g].wait_stream(cur)      # reads of self.obs the caller still has queued on its stream come first (write-after-read)
            self._launch(g, 0, 0, False)
        self._join()
        return self.obs

    def _check_actions(self, actions):
        t = self.torch
        if actions.device != self.device or actions.dtype != t.float32 or tuple(actions.shape) != (self.n, self.act_dim) or not actions.is_contiguous():
            raise ValueError("actions must be the contiguous float32 (N, A) tensor of the whole batch on %s" % self.device)

    def step_group(self, g: int, actions):
        """Control step of group g, asynchronous on `stream(g)`.  `actions`: the WHOLE batch's (N, A) tensor (the group reads its rows) written on
        `stream(g)` (or ordered before it by the caller).  Returns views of the group's rows: (obs, reward, done, info) -- valid on `stream(g)`."""
        t = self.torch
        self._check_actions(actions)
        self._launch(g, actions.data_ptr(), self.updates, True)
        r = self.rows(g)
        with t.cuda.stream(self.streams[g]):
            done = (self.episode_end[r] != 0) | (self.valid[r] == 0)
        return self.obs[r], self.reward[r], done, self._info(r)

    def _join(self):
        cur = self.torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def step(self, actions):
        """all groups (each on its stream, ordered behind the caller's current stream), then the current stream waits for them"""
        self._check_actions(actions)             # (a float64 / transposed / CPU tensor would be read as raw fp32 rows)
        cur = self.torch.cuda.current_stream(self.device)
        for g in range(self.G):
            self.streams[g].wait_stream(cur)
            self._launch(g, actions.data_ptr(), self.updates, True)
        self._join()
        return self.obs, self.reward, (self.episode_end != 0) | (self.valid == 0), self._info(slice(None))

    def close(self):
        self.g.close()
