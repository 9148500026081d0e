"""CheckValidEpisode per update (cSceneSimChar::CheckValidEpisode -> cSimCharacter::HasVelExploded, sim/SimCharacter.cpp:571-586; the
reference's driver asks after EVERY update and ends the episode there, DeepMimic.py:62-80).  The step kernels test the link velocities of
the state each update starts from (EnvSim::kin_pre) and, with DM_END_EPISODE_EARLY, take no further update once it is invalid; without it
the latch still reaches the `valid` output.  Scenario: a character in free fall 50 m up, spinning about z at 55 ... 70 rad/s: the stable-PD
torques and the centrifugal terms drive some link's angular velocity past 100 rad/s a few updates into the control step -- deterministic, no
contact (linear velocities cannot get there: the coordinate-velocity clamp is 100 scaled units = 25 m/s)."""
import numpy as np
import pytest

import parity_common as pc
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
from oracle_lib import Oracle


def _falling(t, o, wz):
    o.reset(0.1)
    p, v = o.sim_state()
    p = p.copy(); v = np.zeros_like(v)
    p[1] = 50.0; v[5] = wz
    return p, v


def _run(lib, prec, packing, tol):
    t = model.load_asset("humanoid3d_walk")
    vys = [60.0, 30.0, 55.0, 70.0]                 # root spin (rad/s): invalid after update 5, never, after update 7, after update 3 (fp64 oracle)
    n = len(vys)
    env = BatchEnv(t, n, precision=prec, lib_path=lib, wave_packing=packing)
    env.reset(kin_times=np.full(n, 0.1), max_times=np.inf)
    st = env.get_state()
    oracles, P, V = [], [], []
    for e in range(n):
        o = Oracle(t); p, v = _falling(t, o, vys[e]); o.set_sim_state(p, v); oracles.append(o); P.append(p); V.append(v)
    env.set_state(pose=np.array(P), vel=np.array(V), tar=st["tar"], kin=st["kin"], clocks=st["clocks"], flags=st["flags"])
    out = env.step(None, pc.DT, 20, open_loop=True, end_early=True)
    got = env.get_state()
    done = []
    for e, o in enumerate(oracles):
        a = o.pose_to_action(o.kin_eval(o.kin_time())[0]); o.set_action(a)
        k = o.control_step(20, pc.DT, end_early=True); done.append(k)
        assert int(out["valid"][e]) == int(o.check_valid_episode()), (e, k)
        assert abs(float(got["clocks"][e][3]) - k * pc.DT) < 1e-12, (e, k, got["clocks"][e])          # the scene timer counts the updates taken
        ps, vs = o.sim_state()
        assert np.abs(got["pose"][e] - ps).max() < tol and np.abs(got["vel"][e] - vs).max() < tol * 100, (e, k)
        assert np.abs(out["state"][e] - o.record_state()).max() < max(tol * 100, 2e-5)
    assert done == [5, 20, 7, 3], done
    assert list(out["valid"]) == [0, 1, 0, 0]
    # without the early end every update runs; the launch still reports that it passed through an invalid state
    env.set_state(pose=np.array(P), vel=np.array(V), tar=st["tar"], kin=st["kin"], clocks=st["clocks"], flags=st["flags"])
    out = env.step(None, pc.DT, 20, open_loop=True, end_early=False)
    assert list(out["valid"]) == [0, 1, 0, 0]
    assert np.allclose(env.get_state()["clocks"][:, 3], 20 * pc.DT)


@pytest.mark.parametrize("packing", [1, 2])
def test_invalid_episode_ends_at_its_update_emulator(emu_lib, packing):
    _run(emu_lib, 64, packing, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,packing,tol", [(64, 1, 1e-9), (64, 2, 1e-9), (32, 2, 2e-4), (32, 1, 2e-4)])
def test_invalid_episode_ends_at_its_update_gpu(hip_lib, prec, packing, tol):
    _run(hip_lib, prec, packing, tol)


# ---- `--enable_root_rot_fail` per update (cSceneImitate::CheckRootRotFail, scenes/SceneImitate.cpp:466-492) ---------------------
# A character in free fall yawing at 25 ... 45 rad/s turns more than 90 degrees away from the clip's root rotation some updates into the second
# control step; the driver's per-update IsEpisodeEnd ends the episode THERE (terminate = fail), not at the next action boundary.
def _run_root_rot(lib, prec, packing, tol):
    t = model.load_asset("humanoid3d_walk")
    t.cfg.enable_root_rot_fail = True
    wys = [45.0, 25.0, 0.0, 35.0]
    n = len(wys)
    env = BatchEnv(t, n, precision=prec, lib_path=lib, wave_packing=packing)
    env.reset(kin_times=np.full(n, 0.1), max_times=np.inf)
    st = env.get_state()
    oracles, P, V = [], [], []
    for e in range(n):
        o = Oracle(t); o.reset(0.1)
        p, v = o.sim_state(); p = p.copy(); v = np.zeros_like(v); p[1] = 50.0; v[4] = wys[e]
        o.set_sim_state(p, v); oracles.append(o); P.append(p); V.append(v)
    env.set_state(pose=np.array(P), vel=np.array(V), tar=st["tar"], kin=st["kin"], clocks=st["clocks"], flags=st["flags"])
    total = [0] * n; ended = [False] * n
    for step in range(3):
        out = env.step(None, pc.DT, 20, open_loop=True, end_early=True)
        got = env.get_state()
        for e, o in enumerate(oracles):
            if ended[e]:
                continue
            a = o.pose_to_action(o.kin_eval(o.kin_time())[0]); o.set_action(a)
            k = o.control_step(20, pc.DT, end_early=True); total[e] += k
            assert abs(float(got["clocks"][e][3]) - total[e] * pc.DT) < 1e-12, (step, e, k, got["clocks"][e][3] / pc.DT)
            assert int(out["terminate"][e]) == o.check_terminate() and int(out["episode_end"][e]) == int(o.is_episode_end()), (step, e)
            assert np.abs(got["pose"][e] - o.sim_state()[0]).max() < tol, (step, e)
            ended[e] = o.is_episode_end()
    assert ended == [True, True, False, True], (ended, total)
    assert total[0] < total[3] < total[1] < 60 and total[2] == 60, total          # faster spin -> earlier end, none of them at an action boundary
    assert all(x % 20 != 0 for x in (total[0], total[1], total[3])), total


@pytest.mark.parametrize("packing", [1, 2])
def test_root_rot_fail_ends_at_its_update_emulator(emu_lib, packing):
    _run_root_rot(emu_lib, 64, packing, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,packing,tol", [(64, 1, 1e-9), (32, 2, 5e-4), (32, 1, 5e-4)])
def test_root_rot_fail_ends_at_its_update_gpu(hip_lib, prec, packing, tol):
    _run_root_rot(hip_lib, prec, packing, tol)


# ---- an invalid episode under DM_AUTO_RESET: the launch resets it like the driver does; the caller must be able to see that ------------
def _run_invalid_auto_reset(lib, prec, packing):
    t = model.load_asset("humanoid3d_walk")
    n = 4
    env = BatchEnv(t, n, precision=prec, lib_path=lib, wave_packing=packing, seed=3)
    env.reset(kin_times=np.full(n, 0.1), max_times=np.inf)
    st = env.get_state()
    P, V = [], []
    for e, wz in enumerate([70.0, 0.0, 0.0, 60.0]):
        o = Oracle(t); p, v = _falling(t, o, wz)
        if wz == 0.0:
            p, v = st["pose"][e].copy(), st["vel"][e].copy()         # envs 1, 2: an ordinary tracking step
        P.append(p); V.append(v)
    env.set_state(pose=np.array(P), vel=np.array(V), tar=st["tar"], kin=st["kin"], clocks=st["clocks"], flags=st["flags"])
    ep0 = env.get_state()["flags"][:, 2].copy()
    out = env.step(None, pc.DT, 20, open_loop=True, auto_reset=True)
    got = env.get_state()
    assert list(out["valid"]) == [0, 1, 1, 0] and list(out["episode_end"]) == [0, 0, 0, 0] and list(out["terminate"]) == [0, 0, 0, 0]
    # the invalid envs were reset inside the launch: new episode counter, timer back at 0, a sane first observation
    assert list(got["flags"][:, 2] - ep0) == [1, 0, 0, 1]
    assert got["clocks"][0, 3] == 0.0 and got["clocks"][3, 3] == 0.0 and abs(got["clocks"][1, 3] - 20 * pc.DT) < 1e-12
    assert np.isfinite(out["state"]).all() and np.abs(got["pose"][[0, 3], 1]).max() < 2.0
    return out


@pytest.mark.parametrize("packing", [1, 2])
def test_invalid_episode_is_reset_and_reported_emulator(emu_lib, packing):
    _run_invalid_auto_reset(emu_lib, 64, packing)


@pytest.mark.gpu
def test_vec_env_done_covers_invalid_episodes_gpu(hip_lib):
    """TorchVecEnv.step: done = episode_end | (valid == 0) -- every env the launch reset is flagged"""
    import torch
    from deepmimic_amd.vec_env import TorchVecEnv
    _run_invalid_auto_reset(hip_lib, 32, 2)
    t = model.load_asset("humanoid3d_walk")
    venv = TorchVecEnv(t, 4, seed=3, lib_path=hip_lib)
    venv.reset()
    st = venv.env.get_state()
    o = Oracle(t); p, v = _falling(t, o, 70.0)
    pose, vel = st["pose"].copy(), st["vel"].copy(); pose[2], vel[2] = p, v
    venv.env.set_state(pose=pose, vel=vel, tar=st["tar"], kin=st["kin"], clocks=st["clocks"], flags=st["flags"])
    a = torch.zeros((4, venv.act_dim), device=venv.device)
    obs, rew, done, info = venv.step(a)
    torch.cuda.synchronize()
    assert done.tolist() == [False, False, True, False] and info["valid"].tolist() == [1, 1, 0, 1]
