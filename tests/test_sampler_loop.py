"""The device-resident sampler loop of INTEGRATION.md section 4 (policy with exploration coin -> control step -> normaliser record; update + bind per
iteration), as written there: imitate scene and a goal scene (goal block as the policy's second input).  GPU only: everything is device pointers."""
import numpy as np
import pytest

from deepmimic_amd import model


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["humanoid3d_walk", "amp_heading_zombie"])
def test_device_resident_sampler_loop(hip_lib, scene):
    import torch
    from deepmimic_amd.normalizer import DeviceNormalizer
    from deepmimic_amd.policy import Policy, random_weights
    from deepmimic_amd.vec_env import TorchVecEnv
    t = model.load_asset(scene)
    n = 256
    env = TorchVecEnv(t, n, seed=3, lib_path=hip_lib)
    obs = env.reset()
    S, G, A = env.obs_dim, env.goal_dim, env.act_dim
    offs = env.env.offsets_scales()
    s_norm = DeviceNormalizer(S, groups_ids=offs["state_norm_groups"], clip=10.0, lib_path=hip_lib)           # rl_agent.py:212-222: s_norm and g_norm apart
    s_norm.set_mean_std(-offs["state_offset"], 1.0 / offs["state_scale"])
    g_norm = DeviceNormalizer(G, clip=10.0, lib_path=hip_lib) if G else None
    w = random_weights(S + G, A, seed=1)
    w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
    actor = Policy(w, s_clip=10.0, lib_path=hip_lib)
    s_norm.bind_policy(actor)
    if G:
        g_norm.bind_policy(actor, first_column=S)
    dev = obs.device
    actions = torch.zeros((n, A), device=dev); logp = torch.zeros(n, device=dev); flags = torch.zeros(n, dtype=torch.int32, device=dev)
    goal = torch.zeros((n, max(G, 1)), device=dev)
    if G:
        goal.copy_(torch.from_numpy(env.env.query_goal()).to(dev))
    iters, steps, explored, dones = 3, 12, 0, 0
    mean_before = s_norm.mean.copy()
    for it in range(iters):
        for k in range(steps):
            actor.forward_device_ex(obs.data_ptr(), n, actions.data_ptr(), goals_ptr=goal.data_ptr() if G else 0, goal_dim=G, logp_ptr=logp.data_ptr(),
                                    exp_flags_ptr=flags.data_ptr(), exp_rate=0.5, sample=True, seed=77, step=it * steps + k)
            explored += int(flags.sum().item())
            obs, reward, done, info = env.step(actions)
            if G:
                goal = info["goal"]
            s_norm.record_device(obs.data_ptr(), n)
            if G:
                g_norm.record_device(goal.data_ptr(), n)
            dones += int(done.sum().item())
            assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(actions).all()) and bool(torch.isfinite(logp).all())
        s_norm.update(); s_norm.bind_policy(actor)
        if G:
            g_norm.update(); g_norm.bind_policy(actor, first_column=S)
            assert g_norm.count == (it + 1) * steps * n
        assert s_norm.count == (it + 1) * steps * n
    total = iters * steps * n
    assert 0.4 < explored / total < 0.6                                  # the coin
    assert np.isfinite(s_norm.mean).all() and np.isfinite(s_norm.std).all() and (s_norm.std >= 0.02).all()
    assert np.abs(s_norm.mean - mean_before).max() > 1e-3                  # the statistics moved towards the rollouts
    assert float(reward.mean().item()) >= 0.0
    if G:
        assert np.isfinite(g_norm.mean).all() and g_norm.std.min() >= 0.02
        g_norm.close()
    env.close(); actor.close(); s_norm.close()
