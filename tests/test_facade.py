"""cDeepMimicCore facade (deepmimic_amd/compat/DeepMimicCore): call protocol of DeepMimic.py:62-80 / env/deepmimic_env.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "deepmimic_amd", "compat")
REF = "/root/reference"


def _core_module():
    if COMPAT not in sys.path:
        sys.path.insert(0, COMPAT)
    from DeepMimicCore import DeepMimicCore
    return DeepMimicCore


def _run_protocol(core, o, n_updates, rng):
    """update_world of DeepMimic.py:62-80 against the oracle driven with the same actions."""
    dt = 1.0 / 600
    n_act = 0
    for u in range(n_updates):
        assert core.NeedNewAction(0) == o.need_new_action()
        if core.NeedNewAction(0):
            s = np.array(core.RecordState(0)); g = core.RecordGoal(0)
            assert s.shape == (o.S,) and g == []
            assert np.abs(s - o.record_state()).max() < 2e-5
            assert abs(core.CalcReward(0) - o.calc_reward()) < 1e-5
            a = (0.2 * rng.normal(size=o.A)).astype(np.float32)
            core.SetAction(0, [float(x) for x in a]); o.set_action(a.astype(np.float64)); n_act += 1
        core.Update(dt); o.update(dt)
        assert core.CheckValidEpisode() == o.check_valid_episode()
        assert core.IsEpisodeEnd() == o.is_episode_end()
        assert core.CheckTerminate(0) == o.check_terminate()
    return n_act


def _check_static_surface(core, tables):
    assert core.IsRLScene() and core.GetNumAgents() == 1 and core.GetActionSpace(0) == 1
    assert core.GetStateSize(0) == tables.state_dim and core.GetActionSize(0) == tables.action_dim and core.GetGoalSize(0) == 0
    assert len(core.BuildStateOffset(0)) == tables.state_dim and len(core.BuildStateScale(0)) == tables.state_dim
    assert len(core.BuildActionBoundMin(0)) == tables.action_dim and len(core.BuildActionScale(0)) == tables.action_dim
    assert core.BuildGoalOffset(0) == [] and core.BuildGoalNormGroups(0) == []
    assert (core.GetRewardMin(0), core.GetRewardMax(0), core.GetRewardFail(0), core.GetRewardSucc(0)) == (0.0, 1.0, 0.0, 1.0)
    assert core.EnableAMPTaskReward() is False and core.GetAMPObsSize() == 0 and core.EnableDraw() is False


def test_facade_protocol_emulator(emu_lib, monkeypatch):
    from deepmimic_amd import model
    from oracle_lib import Oracle
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    mod = _core_module()
    t = model.load_asset("humanoid3d_walk")
    core = mod.cDeepMimicCore(False)
    core.SeedRand(5); core.LoadTables(t, num_update_substeps=10); core.Init()
    _check_static_surface(core, t)
    kin_t = float(core._env.get_state()["clocks"][0][0])          # Reset drew the clip time from the counter RNG
    o = Oracle(t); o.reset(kin_t)
    assert core.NeedNewAction(0) and core.GetNumUpdateSubsteps() == 10 and core.GetTime() == 0.0
    n_act = _run_protocol(core, o, 41, np.random.default_rng(0))
    assert n_act == 3
    core.Reset()
    assert core.NeedNewAction(0) and core.GetTime() == 0.0
    with pytest.raises(RuntimeError):
        core.SetAction(0, [0.0] * 3)
    with pytest.raises(RuntimeError):
        mod.cDeepMimicCore(True)


def test_facade_protocol_physics_v2_emulator(emu_lib, monkeypatch):
    """DM_PHYSICS=2: the same driver protocol against the oracle under cfg.physics = 2"""
    from deepmimic_amd import model
    from oracle_lib import Oracle
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64"); monkeypatch.setenv("DM_PHYSICS", "2")
    mod = _core_module()
    t = model.load_asset("humanoid3d_walk")
    core = mod.cDeepMimicCore(False)
    core.SeedRand(5); core.LoadTables(t, num_update_substeps=10); core.Init()
    assert core._env.physics == 2 and core._batch          # (batched since round 4: the rollback snapshot carries the manifolds)
    kin_t = float(core._env.get_state()["clocks"][0][0])
    o = Oracle(t, physics=2, max_contacts=core._env.max_contacts); o.reset(kin_t)
    assert _run_protocol(core, o, 41, np.random.default_rng(0)) == 3


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "env")), reason="reference checkout not present")
def test_reference_env_wrapper_runs_unmodified(emu_lib, monkeypatch):
    """The reference's own env/deepmimic_env.py (imported from /root/reference, unmodified) on top of the facade."""
    from deepmimic_amd import model
    from oracle_lib import Oracle
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64"); monkeypatch.setenv("DM_DATA_ROOT", REF)
    _core_module()
    monkeypatch.syspath_prepend(REF)
    if "mpi4py" not in sys.modules:            # env/env.py -> learning/normalizer.py -> util/mpi_util.py imports mpi4py (absent here)
        import types
        comm = types.SimpleNamespace(Get_size=lambda: 1, Get_rank=lambda: 0)
        fake = types.ModuleType("mpi4py"); fake.MPI = types.SimpleNamespace(COMM_WORLD=comm, SUM=None)
        monkeypatch.setitem(sys.modules, "mpi4py", fake)
    for m in [k for k in sys.modules if k == "env" or k.startswith("env.") or k == "util" or k.startswith("util.") or k.startswith("learning")]:
        monkeypatch.delitem(sys.modules, m, raising=False)
    from env.deepmimic_env import DeepMimicEnv
    args = ["--arg_file", "args/run_humanoid3d_walk_args.txt"]
    env = DeepMimicEnv(args, False)
    t = model.load_scene_from_args(args, data_root=REF)
    assert env.get_state_size(0) == 227 and env.get_action_size(0) == 28 and env.get_goal_size(0) == 0
    assert env.get_num_agents() == 1 and env.is_rl_scene() and env.get_num_update_substeps() == 10
    assert env.build_state_offset(0).shape == (227,) and env.build_action_bound_min(0).shape == (28,)
    import enum
    env.set_mode(enum.Enum('Mode', {'TRAIN': 0, 'TEST': 1}).TRAIN)     # learning/rl_agent.py Mode enum: set_mode passes mode.value
    env.reset()
    kin_t = float(env._core._env.get_state()["clocks"][0][0])
    o = Oracle(t); o.reset(kin_t)
    rng = np.random.default_rng(1)
    for u in range(22):
        if env.need_new_action(0):
            s = env.record_state(0)
            assert np.abs(s - o.record_state()).max() < 2e-5 and abs(env.calc_reward(0) - o.calc_reward()) < 1e-5
            a = 0.1 * rng.normal(size=28)
            env.set_action(0, a); o.set_action(a.astype(np.float32).astype(np.float64))
        env.update(1.0 / 600); o.update(1.0 / 600)
        assert env.check_valid_episode() == o.check_valid_episode() and env.is_episode_end() == o.is_episode_end()
    env.shutdown()


@pytest.mark.gpu
def test_facade_protocol_gpu(hip_lib, monkeypatch):
    from deepmimic_amd import model
    from oracle_lib import Oracle
    monkeypatch.setenv("DM_HIP_LIB", hip_lib); monkeypatch.setenv("DM_PRECISION", "64")
    mod = _core_module()
    t = model.load_asset("humanoid3d_walk")
    core = mod.cDeepMimicCore(False)
    core.SeedRand(9); core.LoadTables(t, num_update_substeps=10); core.Init()
    _check_static_surface(core, t)
    o = Oracle(t); o.reset(float(core._env.get_state()["clocks"][0][0]))
    assert _run_protocol(core, o, 101, np.random.default_rng(2)) == 6


def test_timer_annealing_follows_sample_count(emu_lib, monkeypatch):
    """a3: SetSampleCount blends time_lim_* -> time_end_lim_* with lerp = clamp(count / anneal_samples)^4
    (scenes/RLSceneSimChar.cpp:223-227,330-347); test mode pins time_end_lim_max (:277-284).  Args of
    args/train_humanoid3d_walk_args.txt: 0.5 s -> 20 s over 32e6 samples."""
    import copy
    from deepmimic_amd import model
    cfg = copy.deepcopy(model.load_asset("humanoid3d_walk").cfg)
    cfg.time_lim_min = cfg.time_lim_max = 0.5
    cfg.time_end_lim_min = cfg.time_end_lim_max = 20.0
    cfg.anneal_samples = 32000000
    assert model.timer_limits(cfg, False, 0) == (0.5, 0.5)
    lo, hi = model.timer_limits(cfg, False, 16000000)
    assert abs(lo - (0.5 + 19.5 / 16)) < 1e-12 and lo == hi
    assert model.timer_limits(cfg, False, 10 ** 9) == (20.0, 20.0)
    assert model.timer_limits(cfg, True, 0) == (20.0, 20.0)
    unset = model.SceneConfig(anneal_samples=1000)                       # limits never given: 0 * inf in the reference
    assert model.timer_limits(unset, False, 1000) == (np.inf, np.inf)

    DeepMimicCore = _core_module()
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    t = model.load_asset("humanoid3d_walk"); t.cfg = cfg
    core = DeepMimicCore.cDeepMimicCore(False)
    core.SeedRand(3); core.LoadTables(t, 10); core.Init()

    def episode_len():
        core.Reset(); n = 0
        while not core.IsEpisodeEnd() and n < 40:
            core.Update(1.0 / 30); n += 1
        return n
    assert episode_len() in (15, 16)                                   # 0.5 s at 30 Hz (15 x 1/30 rounds just below 0.5)
    core.SetSampleCount(16000000)
    core.Reset()
    assert core._env.get_state()["clocks"][0][4] == pytest.approx(0.5 + 19.5 / 16)
    core.SetMode(1)                                                    # eModeTest
    core.Reset()
    assert core._env.get_state()["clocks"][0][4] == pytest.approx(20.0)


def _drive(core, n_updates, seed, peek_at=()):
    """the reference driver's loop (DeepMimic.py:62-80): returns everything it observed"""
    rng = np.random.default_rng(seed)
    dt, seen = 1.0 / 600, []
    for u in range(n_updates):
        if core.NeedNewAction(0):
            seen.append(("s", np.array(core.RecordState(0)), core.CalcReward(0)))
            core.SetAction(0, [float(x) for x in (0.2 * rng.normal(size=core.GetActionSize(0))).astype(np.float32)])
        core.Update(dt)
        if u in peek_at:                                   # a caller that looks at an intermediate update of the control step
            seen.append(("p", np.array(core.RecordState(0)), core.CalcReward(0)))
        valid, end = core.CheckValidEpisode(), core.IsEpisodeEnd()
        seen.append(("f", valid, end, core.CheckTerminate(0), core.GetTime()))
        if not valid or end:
            seen.append(("e", np.array(core.RecordState(0)), core.CalcReward(0)))
            core.Reset()
    return seen


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[0] == y[0]
        if x[0] == "f":
            assert x[1:4] == y[1:4] and abs(x[4] - y[4]) < 1e-12, (x, y)
        else:
            assert np.array_equal(x[1], y[1]) and x[2] == y[2]


@pytest.mark.parametrize("physics", ["1", "2"])
def test_batched_control_step_equals_update_by_update(emu_lib, monkeypatch, physics):
    """(physics 2 since round 4: the rollback snapshot carries the ground manifolds, dm_get_manifolds.)
    DM_FACADE_BATCH: one launch per control step (+ rollback / replay when a caller looks inside the step) gives bit-identical
    observations, rewards, flags and clock to one launch per update; the open-loop-ish random actions make the walker fall, so the
    early episode end and the reset path are exercised too."""
    from deepmimic_amd import model
    mod = _core_module()
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64"); monkeypatch.setenv("DM_PHYSICS", physics)
    t = model.load_asset("humanoid3d_walk")
    t.cfg.time_lim_min = t.cfg.time_lim_max = t.cfg.time_end_lim_min = t.cfg.time_end_lim_max = 0.175   # the episode timer ends inside a control step (update 105; 20 per step)
    runs = {}
    for batch in ("1", "0"):
        monkeypatch.setenv("DM_FACADE_BATCH", batch)
        core = mod.cDeepMimicCore(False)
        core.SeedRand(4); core.LoadTables(t, 10); core.Init()
        runs[batch] = (_drive(core, 260, seed=3, peek_at=(47, 131)), dict(core.stats))
        core.Shutdown()
    _same(runs["1"][0], runs["0"][0])
    sb, su = runs["1"][1], runs["0"][1]
    assert su["launches"] == su["updates"] == 260
    assert sb["rollbacks"] == 2
    assert sb["launches"] <= 260 // 20 + 2 + 2 * 20 + 4, sb          # ~1 per control step, + the two replays
    assert any(x[0] == "e" for x in runs["1"][0])                    # an episode ended inside the window


@pytest.mark.parametrize("asset", ["amp_dribble_zombie", "amp_heading_getup", "amp_strike_punch"])
def test_batched_control_step_equals_update_by_update_goal_scenes(emu_lib, monkeypatch, asset):
    """the same for the task scenes: a rollback has to bring back the goal row (target, timers, draw counter, get-up / hit state) and,
    for dribble_amp, the ball -- RecordGoal is part of what the driver sees"""
    from deepmimic_amd import model
    mod = _core_module()
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    t = model.load_asset(asset)
    t.cfg.time_lim_min = t.cfg.time_lim_max = t.cfg.time_end_lim_min = t.cfg.time_end_lim_max = 0.175
    runs = {}
    for batch in ("1", "0"):
        monkeypatch.setenv("DM_FACADE_BATCH", batch)
        core = mod.cDeepMimicCore(False)
        core.SeedRand(4); core.LoadTables(t, 10); core.Init()
        rng = np.random.default_rng(3)
        seen = []
        for u in range(150):
            if core.NeedNewAction(0):
                seen.append((np.array(core.RecordState(0)), np.array(core.RecordGoal(0)), core.CalcReward(0)))
                core.SetAction(0, [float(x) for x in (0.2 * rng.normal(size=core.GetActionSize(0))).astype(np.float32)])
            core.Update(1.0 / 600)
            if u in (47, 91):
                seen.append((np.array(core.RecordState(0)), np.array(core.RecordGoal(0)), core.CalcReward(0)))
            end = core.IsEpisodeEnd()
            seen.append((np.array([float(end), core.CheckTerminate(0), core.GetTime()]), np.zeros(1), 0.0))
            if end:
                core.Reset()
        runs[batch] = (seen, dict(core.stats))
        core.Shutdown()
    assert len(runs["1"][0]) == len(runs["0"][0])
    for x, y in zip(runs["1"][0], runs["0"][0]):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2]
    assert runs["1"][1]["rollbacks"] == 2 and runs["1"][1]["launches"] < runs["0"][1]["launches"] // 3


def test_imitate_amp_time_warp_test_return(emu_lib, monkeypatch):
    """cSceneImitateAMP::CalcReward in test mode = the time-warp alignment cost at the episode end (SceneImitateAMP.cpp:173-205):
    0 while the episode runs; at the end DTW(sim joints, kin joints) over the action-boundary samples + 1 per step the episode
    fell short of the buffer.  A character that tracks the clip perfectly for the whole buffer would score ~0."""
    from deepmimic_amd import model
    mod = _core_module()
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    t = model.load_asset("humanoid3d_walk")
    t.cfg.scene = "imitate_amp"
    t.cfg.time_lim_min = t.cfg.time_lim_max = t.cfg.time_end_lim_min = t.cfg.time_end_lim_max = 0.2     # 6 control steps
    core = mod.cDeepMimicCore(False)
    core.SeedRand(2); core.LoadTables(t, 10); core.Init()
    core.SetMode(core.eModeTest); core.Reset()
    assert core._tw is not None and core._tw["size"] == int(np.ceil(30 * 0.2)) + 2
    steps = 0
    while True:
        if core.NeedNewAction(0):
            assert core.CalcReward(0) == 0.0
            core.SetAction(0, [0.0] * core.GetActionSize(0)); steps += 1
        core.Update(1.0 / 600)
        if core.IsEpisodeEnd():
            break
    n = len(core._tw["sim"])
    assert n == steps + 1 == 7                                   # the reset sample + one per action
    r = core.CalcReward(0)
    want = model.time_warp_cost(np.array(core._tw["sim"]), np.array(core._tw["kin"])) + (core._tw["size"] - n)
    assert r == want and (core._tw["size"] - n) <= r < (core._tw["size"] - n) + 0.5
    # the first pair of samples is the reset state: sim == kin, distance 0
    assert np.abs(core._tw["sim"][0] - core._tw["kin"][0]).max() < 1e-9 + 0.02     # (root lifted by the ground-clearance of the reset)
    core.SetMode(core.eModeTrain); core.Reset()
    core.SetAction(0, [0.0] * core.GetActionSize(0)); core.Update(1.0 / 600)
    assert core.CalcReward(0) == 0.0


def test_facade_exponential_timer(emu_lib, monkeypatch):
    """`--timer_type exp --time_lim_exp` (util/Timer.cpp:27-45, 64-67) through ParseArgs: every Reset draws min(time_lim_min + Exp(time_lim_exp), time_lim_max);
    test mode pins time_end_lim_max (cRLSceneSimChar::ResetTimers)"""
    from deepmimic_amd import model, streams
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    monkeypatch.setenv("DM_RNG", "counter")      # this test pins the counter-based exponential draw of BatchEnv.reset; the reference-order draws: tests/test_ref_rng.py
    mod = _core_module()
    t = model.load_asset("humanoid3d_walk")
    t.cfg.timer_type = "exp"; t.cfg.time_lim_min, t.cfg.time_lim_max, t.cfg.time_lim_exp = 0.5, 4.0, 1.0
    t.cfg.time_end_lim_min, t.cfg.time_end_lim_max, t.cfg.time_end_lim_exp = 0.5, 20.0, 5.0; t.cfg.anneal_samples = 1000
    core = mod.cDeepMimicCore(False)
    core.SeedRand(3); core.LoadTables(t, num_update_substeps=10); core.Init()
    seen = []
    for k in range(5):
        ep = int(core._env.get_state()["flags"][0][2])
        core.Reset()
        mt = float(core._env.get_state()["clocks"][0][4])
        assert abs(mt - model.draw_time_limit("exp", 0.5, 4.0, 1.0, streams.reset_rand01(core._seed, 0, ep, 1))) < 1e-12
        seen.append(mt)
    assert len(set(seen)) == 5 and min(seen) >= 0.5 and max(seen) <= 4.0
    core.SetSampleCount(1000)                                       # end of the annealing: Blend() reaches the end parameters, exp included
    ep = int(core._env.get_state()["flags"][0][2]); core.Reset()
    assert abs(float(core._env.get_state()["clocks"][0][4]) - model.draw_time_limit("exp", 0.5, 20.0, 5.0, streams.reset_rand01(core._seed, 0, ep, 1))) < 1e-12
    core.SetMode(core.eModeTest); core.Reset()
    assert float(core._env.get_state()["clocks"][0][4]) == 20.0


def test_imitate_amp_time_warp_multi_clip(emu_lib, monkeypatch):
    """round 4: the test-mode time-warp return with a multi-clip dataset (`--kin_ctrl clips`): the kinematic side of the alignment
    is the clip the env was reset to.  (a) a dataset of two copies of the walk clip scores exactly what the single-clip scene scores from the same
    start; (b) on a walk + spinkick dataset the host sampler of EVERY clip equals the oracle's kinematic character reset to that clip."""
    import copy
    from deepmimic_amd import model
    mod = _core_module()
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64"); monkeypatch.setenv("DM_RNG", "counter")
    walk, kick = model.load_asset("humanoid3d_walk"), model.load_asset("humanoid3d_spinkick")

    def dataset(a, b):
        t = copy.deepcopy(a)
        t.cfg.scene = "imitate_amp"
        t.cfg.time_lim_min = t.cfg.time_lim_max = t.cfg.time_end_lim_min = t.cfg.time_end_lim_max = 0.2
        t.frames = np.concatenate([a.frames, b.frames])
        t.clip_starts = np.array([0, a.frames.shape[0], a.frames.shape[0] + b.frames.shape[0]], np.int32)
        t.clip_weights = np.array([0.5, 0.5]); t.clip_loops = np.array([int(a.loop), int(b.loop)], np.int32)
        return t

    def episode(t, kin_time):
        core = mod.cDeepMimicCore(False)
        core.SeedRand(2); core.LoadTables(t, 10); core.Init()
        core.SetMode(core.eModeTest); core.Reset()
        clip = int(core._env.get_clips()[0]) if t.num_clips > 1 else 0
        core._env.reset(kin_times=[kin_time], max_times=[0.2]); core._after_reset.__self__._sync_clocks()      # same start for both runs
        core._tw["sim"], core._tw["kin"] = [], []; core._tw_sample()
        while True:
            if core.NeedNewAction(0):
                core.SetAction(0, [0.0] * core.GetActionSize(0))
            core.Update(1.0 / 600)
            if core.IsEpisodeEnd():
                break
        return core.CalcReward(0), clip, core

    single = copy.deepcopy(walk); single.cfg.scene = "imitate_amp"
    single.cfg.time_lim_min = single.cfg.time_lim_max = single.cfg.time_end_lim_min = single.cfg.time_end_lim_max = 0.2
    r1, _, _ = episode(single, 0.3)
    r2, clip, core = episode(dataset(walk, walk), 0.3)
    assert r1 == r2 and r1 > 0, (r1, r2, clip)
    # (b) the sampler of each clip against the oracle's kinematic character reset to that clip (oracle: pinned to the reference's cKinCharacter / cClipsController)
    from oracle_lib import Oracle
    t = dataset(walk, kick)
    o = Oracle(t)
    for clip, kt in ((0, 0.3), (1, 0.45), (1, 1.1), (0, 1.9)):
        o.reset_ex(kt, np.inf, clip, 0.0)
        kp, _, ko = o.kin_state()
        want = model.KinSampler(t, clip).pose(kt, ko[0:3], ko[3:7])
        assert np.abs(kp - want).max() < 1e-12, (clip, kt, np.abs(kp - want).max())
