"""`--scene imitate_amp`: AMP observations, zero reward, fall-only termination.

CPU part: oracle known answers + device path on the emulator build; GPU part through the C-ABI (marked gpu)."""
import copy
import os

import numpy as np
import pytest

from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
from oracle_lib import Oracle

DT = 1.0 / 600
REF = "/root/reference"


def amp_tables(name, local_root=False):
    t = model.load_asset(name)
    t.cfg = copy.deepcopy(t.cfg)
    t.cfg.scene = "imitate_amp"
    t.cfg.enable_amp_obs_local_root = local_root
    return t


# ---------------------------------------------------------------- oracle known answers
def test_amp_obs_size_matches_reference_formula(oracle_built):
    """GetAMPObsSize (SceneImitateAMP.cpp:76-86,213-257): humanoid 2 x (1 + 6 + 8 x 6 + 4 + 4 x 3 + 42) = 226;
    dog 2 x (1 + 6 + 18 x 6 + 4 x ... ) from its own joint table."""
    o = Oracle(amp_tables("humanoid3d_walk"))
    assert o.amp_obs_size() == 226
    t = amp_tables("dog3d_pace")
    n_sph = int((t.joint_mat[1:, model.JD_TYPE] == model.JT_SPHERICAL).sum())
    n_rev = int((t.joint_mat[1:, model.JD_TYPE] == model.JT_REVOLUTE).sum())
    n_ee = int((t.joint_mat[:, model.JD_IS_EE] != 0).sum())
    assert Oracle(t).amp_obs_size() == 2 * (1 + 6 + 6 * n_sph + n_rev + 3 * n_ee + t.pose_dim - 1)


def test_amp_agent_equals_expert_right_after_reset(oracle_built):
    """After Reset at clip time t the sim character *is* the clip pose and the history is the clip one control period
    earlier (InitHist), so RecordAMPObsAgent == RecordAMPObsExpert(t) up to the root x/z the observation ignores --
    as long as the reset did not have to lift the character out of the ground."""
    t = amp_tables("humanoid3d_walk")
    o = Oracle(t)
    for tt in (0.3, 0.71):
        o.reset(tt)
        a, e = o.amp_obs_agent(), o.amp_obs_expert(tt)
        lift = o.sim_state()[0][1] - o.kin_eval(tt)[0][1]
        ps = o.amp_obs_size() // 2 - (t.pose_dim - 1)
        a2 = a.copy(); a2[0] -= lift; a2[ps] -= lift
        assert np.abs(a2 - e).max() < 1e-12


def test_amp_scene_reward_zero_and_fall_only_termination(oracle_built):
    t = amp_tables("humanoid3d_walk")
    o = Oracle(t); o.reset(0.0)
    assert o.calc_reward() == 0.0 and o.check_terminate() == 0
    for k in range(40):                                   # zero actions: the character collapses -> Fail
        o.set_action(np.zeros(o.A))
        for u in range(20):
            o.update(DT)
    assert o.check_terminate() == 1 and o.calc_reward() == 0.0


def test_amp_history_is_the_state_at_the_last_action_latch(oracle_built):
    t = amp_tables("humanoid3d_walk")
    o = Oracle(t); o.reset(0.2)
    p0, v0 = o.sim_state()
    o.set_action(o.pose_to_action(o.kin_state()[0]))
    for u in range(20):
        o.update(DT)
    pp, pv = o.prev_state()
    assert np.abs(pp - p0).max() < 1e-12 and np.abs(pv - v0).max() < 1e-12
    a = o.amp_obs_agent()
    ps = (o.amp_obs_size() - 2 * (t.pose_dim - 1)) // 2
    assert abs(a[0] - o.sim_state()[0][1]) < 1e-12 and abs(a[ps] - p0[1]) < 1e-12       # root heights now / at the latch


# ---------------------------------------------------------------- device path vs oracle
def amp_rollout_compare(name, precision, lib_path, steps, t0s, wave_packing=0, local_root=False, facade_order=False):
    """Open-loop rollout of len(t0s) imitate_amp envs; per step compares RecordAMPObsAgent, reward, flags with the oracle."""
    t = amp_tables(name, local_root)
    n = len(t0s)
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing)
    assert env.amp_size == Oracle(t).amp_obs_size()
    env.reset(kin_times=np.array(t0s, dtype=np.float64), max_times=np.inf)
    oracles = []
    for t0 in t0s:
        o = Oracle(t); o.reset(t0); oracles.append(o)
    # InitHist: observation available right after the reset
    a0 = env.query_amp()
    d0 = max(np.abs(a0[e] - o.amp_obs_agent()).max() for e, o in enumerate(oracles))
    da, ok = np.zeros(n), True
    for k in range(steps):
        if facade_order:                                   # SetAction / 20 x Update / queries as separate calls
            acts = np.array([o.pose_to_action(o.kin_state()[0]) for o in oracles]).astype(np.float32)
            env.set_action(acts); env.update(DT, 20)
            q = env.query(); amp = env.query_amp()
            out = dict(reward=q["reward"], terminate=q["terminate"], amp_obs=amp)
            for e, o in enumerate(oracles):
                o.set_action(acts[e].astype(np.float64))
        else:
            out = env.step(None, DT, 20, open_loop=True, amp=True)
            for o in oracles:
                o.set_action(o.pose_to_action(o.kin_state()[0]))
        for e, o in enumerate(oracles):
            for u in range(20):
                o.update(DT)
            da[e] = max(da[e], np.abs(out["amp_obs"][e] - o.amp_obs_agent()).max())
            ok &= float(out["reward"][e]) == 0.0 and int(out["terminate"][e]) == o.check_terminate()
    return d0, da, ok


def amp_fp32_floor(name, steps, t0):
    """max |AMP obs(fp64 oracle) - AMP obs(fp32 oracle)| over a free-running open-loop rollout: the single-precision noise
    floor of the algorithm itself."""
    t = amp_tables(name)
    o, f = Oracle(t), Oracle(t, variant="f32")
    o.reset(t0); f.reset(t0)
    worst = 0.0
    for k in range(steps):
        for x in (o, f):
            x.set_action(x.pose_to_action(x.kin_state()[0]))
            for u in range(20):
                x.update(DT)
        worst = max(worst, np.abs(o.amp_obs_agent() - f.amp_obs_agent()).max())
    return worst


def test_amp_agent_obs_emulator_fp64(emu_lib):
    d0, da, ok = amp_rollout_compare("humanoid3d_walk", 64, emu_lib, 2, [0.0, 0.37], wave_packing=1)
    assert ok and d0 < 1e-6 and da.max() < 1e-5, (d0, da)


def test_amp_agent_obs_emulator_duo_and_local_root(emu_lib):
    d0, da, ok = amp_rollout_compare("humanoid3d_walk", 64, emu_lib, 2, [0.1, 0.52], wave_packing=2, local_root=True)
    assert ok and d0 < 1e-6 and da.max() < 1e-5, (d0, da)


def test_amp_agent_obs_emulator_facade_call_order(emu_lib):
    d0, da, ok = amp_rollout_compare("humanoid3d_walk", 64, emu_lib, 2, [0.25], wave_packing=1, facade_order=True)
    assert ok and da.max() < 1e-5, (d0, da)


def test_amp_expert_obs_emulator(emu_lib):
    for name in ("humanoid3d_walk", "dog3d_pace"):
        t = amp_tables(name)
        env = BatchEnv(t, 2, precision=64, lib_path=emu_lib, wave_packing=1)
        o = Oracle(t)
        times = np.array([0.0, 0.013, 0.4, o.duration - 1e-3, 1.7 * o.duration])      # t - dt < 0 and t beyond one cycle
        o.reset(0.0)
        gh = o.kin_state()[2][1]                       # kin origin height after the reset (the ground lift), what the reference passes
        got = env.amp_expert(len(times), times, ground_h=gh)
        for i, tt in enumerate(times):
            want = o.amp_obs_expert(tt)
            assert np.abs(got[i] - want).max() < 1e-6, (name, tt, np.abs(got[i] - want).max())
        rnd = env.amp_expert(16)                                                      # RNG-drawn times
        assert np.isfinite(rnd).all() and np.abs(rnd).max() < 50 and not np.array_equal(rnd[0], rnd[1])
        assert not np.array_equal(rnd, env.amp_expert(16))                            # next call, next stream


def test_amp_entry_points_refuse_plain_imitate_scene(emu_lib):
    env = BatchEnv(model.load_asset("humanoid3d_walk"), 1, precision=64, lib_path=emu_lib)
    assert env.amp_size == 0
    with pytest.raises(RuntimeError, match="imitate_amp"):
        env.query_amp()
    with pytest.raises(RuntimeError, match="imitate_amp"):
        env.amp_expert(2)


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name,pack,prec,tol", [("humanoid3d_walk", 1, 64, 1e-5), ("humanoid3d_walk", 2, 64, 1e-5),
                                                ("dog3d_pace", 0, 64, 1e-5), ("humanoid3d_walk", 2, 32, 2e-3)])
def test_amp_agent_obs_gpu(hip_lib, name, pack, prec, tol):
    """Free-running open-loop rollout up to just before the open-loop character falls (control step ~13; the fall itself is
    a chaotic many-contact motion in which even the fp64 kernel and the fp64 oracle separate).  fp32: the observation
    holds joint velocities (rad/s, up to ~10) of a contact simulation, hence the absolute tolerance 2e-3."""
    steps = 12 if prec == 64 else 10
    t0s = [0.0, 0.37, 0.6, 0.9]
    d0, da, ok = amp_rollout_compare(name, prec, hip_lib, steps, t0s, wave_packing=pack)
    if prec == 32:                      # the float build of the oracle, free-running, is itself ~5e-3 away from the fp64 oracle
        tol = max(tol, 2 * max(amp_fp32_floor(name, steps, t0) for t0 in t0s))
    assert ok and d0 < 1e-5 and da.max() < tol, (d0, da, tol)


@pytest.mark.gpu
def test_amp_expert_obs_gpu(hip_lib):
    t = amp_tables("humanoid3d_spinkick")
    env = BatchEnv(t, 2, precision=32, lib_path=hip_lib)
    o = Oracle(t)
    times = np.linspace(0.0, 2.0 * o.duration, 257)
    o.reset(0.0)
    got = env.amp_expert(len(times), times, ground_h=o.kin_state()[2][1])
    want = np.array([o.amp_obs_expert(tt) for tt in times])
    assert np.abs(got - want).max() < 2e-4, np.abs(got - want).max()


@pytest.mark.gpu
def test_amp_auto_reset_keeps_end_of_path_observation(hip_lib):
    """4096 envs with zero actions: every episode ends by a fall; the AMP observation returned with the terminal flags is
    that of the finished episode (root height of a fallen character), while `state` already belongs to the new episode."""
    t = amp_tables("humanoid3d_walk")
    env = BatchEnv(t, 4096, precision=32, lib_path=hip_lib)
    env.reset()
    seen = 0
    for k in range(45):
        out = env.step(np.zeros((4096, env.A), np.float32), DT, 20, auto_reset=True, amp=True)
        assert np.isfinite(out["amp_obs"]).all()
        done = out["terminate"] == 1
        if done.any():
            seen += int(done.sum())
            assert (out["amp_obs"][done, 0] < 0.6).all()                # fallen: root low
            base = 1 if t.enable_phase_input else 0
            assert (out["state"][done, base] > 0.7).all()               # new episode: root at standing height
    assert seen > 3000


def test_facade_amp_surface(emu_lib, monkeypatch):
    """cDeepMimicCore AMP methods (DeepMimicCore.h:75-81) as learning/amp_agent.py uses them: sizes, offset/scale/groups,
    agent observation after each control step, expert samples."""
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepmimic_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from DeepMimicCore import DeepMimicCore
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    t = amp_tables("humanoid3d_walk")
    core = DeepMimicCore.cDeepMimicCore(False)
    core.SeedRand(5); core.LoadTables(t, 10); core.Init()
    assert core.GetName() == "Imitate AMP" and core.GetAMPObsSize() == 226 and core.EnableAMPTaskReward() is False
    assert core.GetAMPObsOffset() == [0.0] * 226 and core.GetAMPObsScale() == [1.0] * 226 and core.GetAMPObsNormGroup() == [0] * 226
    o = Oracle(t)
    kt = float(core._env.get_state()["clocks"][0][0])
    o.reset(kt)
    rng = np.random.default_rng(0)
    for k in range(2):
        a = (0.1 * rng.normal(size=o.A)).astype(np.float32)
        core.SetAction(0, [float(x) for x in a]); o.set_action(a.astype(np.float64))
        for u in range(20):
            core.Update(DT); o.update(DT)
        assert core.NeedNewAction(0)
        got = np.array(core.RecordAMPObsAgent(0))
        assert got.shape == (226,) and np.abs(got - o.amp_obs_agent()).max() < 1e-5
        assert core.CalcReward(0) == 0.0
    ex = np.array(core.RecordAMPObsExpert(0))
    assert ex.shape == (226,) and np.isfinite(ex).all()
    plain = DeepMimicCore.cDeepMimicCore(False)
    plain.LoadTables(model.load_asset("humanoid3d_walk"), 10); plain.Init()
    assert plain.GetAMPObsSize() == 0 and plain.RecordAMPObsAgent(0) == [] and plain.RecordAMPObsExpert(0) == []
