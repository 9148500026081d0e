"""env-steps/s of every scene kind the path steps on the device, on ONE box: imitate (the headline workload), imitate_amp (AMP observation written
every step), the five goal-conditioned task scenes (goal vector + AMP observation + task reward; multi-clip datasets), and imitate with random
perturbations.  Open-loop clip tracking with auto-reset, 4096 envs, outputs into device tensors (deepmimic_amd.vec_env).
usage: python tools/gpu_scene_bench.py [envs] [scene label: run only this one]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deepmimic_amd import model  # noqa: E402
from deepmimic_amd.vec_env import TorchVecEnv, TorchVecEnvGroups  # noqa: E402
import time  # noqa: E402

SCENES = [("imitate", "humanoid3d_walk", None), ("imitate + perturbs", "humanoid3d_walk", "perturb"), ("imitate_amp", "humanoid3d_walk", "amp"),
          ("target_amp", "amp_target_zombie", None), ("heading_amp", "amp_heading_zombie", None), ("heading_amp (4 clips)", "amp_heading_clips4", None),
          ("heading_amp_getup", "amp_heading_getup", None), ("strike_amp", "amp_strike_punch", None), ("dribble_amp", "amp_dribble_zombie", None)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    only = sys.argv[2] if len(sys.argv) > 2 else None
    out = {"envs": n, "scenes": {}}
    for label, asset, mod in SCENES:
        if only and label != only:
            continue
        t = model.load_asset(asset)
        # train mode at the END of the timer annealing (time_end_lim_*, scenes/RLSceneSimChar.cpp:338-347): the arg files start it at 0.5 s episodes
        if t.cfg.time_end_lim_max is not None:
            t.cfg.time_lim_min, t.cfg.time_lim_max = float(t.cfg.time_end_lim_min), float(t.cfg.time_end_lim_max)
        if mod == "perturb":
            t.cfg.enable_rand_perturbs = True; t.cfg.perturb_time_min, t.cfg.perturb_time_max = 1.0, 2.0
        if mod == "amp":
            t.cfg.scene = "imitate_amp"
        v = TorchVecEnv(t, n, seed=1234, amp_obs=True)
        v.reset()

        def launch():
            v._enter()
            v.env.step_device(0, v.obs.data_ptr(), v.reward.data_ptr(), v.terminate.data_ptr(), v.valid.data_ptr(), v.episode_end.data_ptr(),
                              timestep=v.timestep, n_updates=v.updates, auto_reset=True, open_loop=True,
                              amp_ptr=v.amp_obs.data_ptr() if v.amp_obs is not None else 0)
            v._leave()
        for _ in range(60):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        ends = 0
        for _ in range(100):
            launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        out["scenes"][label] = {"asset": asset, "state_dim": v.obs_dim, "goal_dim": v.goal_dim, "amp_obs_dim": 0 if v.amp_obs is None else int(v.amp_obs.shape[1]),
                                "ms_per_step": ms, "env_steps_per_s": n / (ms * 1e-3), "episodes_ended_in_last_step": int(v.episode_end.sum().item()),
                                "finite": bool(torch.isfinite(v.obs).all().item() and torch.isfinite(v.reward).all().item())}
        v.close()
        # the same scene as two env groups on their own streams (deepmimic_amd/groups.py): wall clock over 100 control steps of both groups
        g = TorchVecEnvGroups(t, n, groups=2, seed=1234, amp_obs=True)
        g.reset()

        def glaunch():
            for k in range(g.G):
                g.g.step_group_device(k, 0, g.obs.data_ptr(), g.reward.data_ptr(), g.terminate.data_ptr(), g.valid.data_ptr(), g.episode_end.data_ptr(),
                                      timestep=g.timestep, n_updates=g.updates, auto_reset=True, open_loop=True,
                                      amp_ptr=g.amp_obs.data_ptr() if g.amp_obs is not None else 0)
        for _ in range(60):
            glaunch()
        g.g.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            glaunch()
        g.g.synchronize(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out["scenes"][label]["two_groups_env_steps_per_s"] = n * 100 / dt
        out["scenes"][label]["two_groups_finite"] = bool(torch.isfinite(g.obs).all().item() and torch.isfinite(g.reward).all().item())
        g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
