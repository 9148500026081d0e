"""Goal-conditioned AMP task scenes: `--scene target_amp` (scenes/SceneTargetAMP.cpp),
`--scene heading_amp` (scenes/SceneHeadingAMP.cpp), `--scene heading_amp_getup` (scenes/SceneHeadingAMPGetup.cpp) and
`--scene strike_amp` (scenes/SceneStrikeAMP.cpp) with multi-clip datasets (anim/ClipsController.cpp) and
enable_rand_rot_reset (scenes/SceneImitate.cpp:331-349).  Device path (emulator build on CPU, HIP library marked gpu) vs
the oracle restatement, closed loop with seeded random actions through auto-resets; every random draw is mirrored."""
import numpy as np
import pytest

import parity_common as pc
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
from oracle_lib import Oracle

SCENES = ["amp_heading_zombie", "amp_target_zombie", "amp_heading_clips4"]


def test_assets_describe_the_scenes():
    h, tg, c4 = (model.load_asset(n) for n in SCENES)
    assert (h.goal_kind, tg.goal_kind, c4.goal_kind) == (2, 1, 2) and h.goal_dim == 3
    assert h.cfg.enable_rand_rot_reset and h.cfg.enable_amp_obs_local_root and not h.enable_phase_input
    assert (h.cfg.rand_target_time_min, h.cfg.rand_target_time_max) == (0.2, 0.5)            # args/train_amp_heading_humanoid3d_zombie_args.txt
    assert (tg.cfg.rand_target_time_min, tg.cfg.rand_target_time_max, tg.cfg.tar_fail_dist) == (5.0, 10.0, 15.0)
    assert c4.num_clips == 4 and list(c4.clip_loops) == [1, 1, 0, 0] and h.num_clips == 1


def test_goal_closed_form(oracle_built):
    """RecordGoal / CalcReward against the formulas of the scene sources evaluated by hand (numpy) on the oracle's state."""
    for name in ("amp_heading_zombie", "amp_target_zombie"):
        t = model.load_asset(name)
        o = Oracle(t); o.goal_rng(9, 0, 0); o.reset_ex(0.4, np.inf, 0, 0.7)
        a = np.zeros(o.A)
        o.set_action(a); o.control_step(20, pc.DT, end_early=False)
        gs = o.goal_state(); p, v = o.sim_state()
        links = o.links(); mass = t.body_defs[:, model.BD_MASS]
        com = (links[:, 0:3] * mass[:, None]).sum(0) / mass.sum()
        fwd = np.array([1 - 2 * (p[5] ** 2 + p[6] ** 2), 0, 2 * (p[4] * p[6] - p[3] * p[5])])     # q * e_x, x and z components
        heading = np.arctan2(-fwd[2], fwd[0])
        step_dur = o.lib.orc_time(o.h) + 0.4 - gs[10]
        dcom = com - gs[7:10]
        if t.goal_kind == 2:
            th = gs[3] - heading
            assert np.abs(o.record_goal() - [np.cos(th), -np.sin(th), gs[4]]).max() < 1e-12
            av = dcom / step_dur; av[1] = 0
            sp = np.cos(gs[3]) * av[0] - np.sin(gs[3]) * av[2]
            want = np.exp(-t.cfg.vel_reward_scale * (gs[4] - sp) ** 2) if sp > 0 else 0.0
        else:
            rel = np.array([gs[0] - p[0], 0, gs[2] - p[2]]); d = np.linalg.norm(rel)
            c, s = np.cos(-heading), np.sin(-heading)
            r = np.array([c * rel[0] + s * rel[2], 0, -s * rel[0] + c * rel[2]]) / d
            assert np.abs(o.record_goal() - [r[0], r[2], d]).max() < 1e-12
            ct = np.array([gs[0] - com[0], 0, gs[2] - com[2]]); cd = np.linalg.norm(ct)
            avg = ct.dot(dcom) / cd / step_dur
            ve = max(t.cfg.tar_speed - avg, 0.0) if t.cfg.enable_min_tar_vel else t.cfg.tar_speed - avg
            vr = 1.0 if d < t.cfg.target_succ_dist else (0.0 if avg < 0 else np.exp(-4 / t.cfg.tar_speed ** 2 * ve * ve))
            want = 0.6 * np.exp(-t.cfg.pos_reward_scale * d * d) + 0.4 * vr
        assert abs(o.calc_reward() - want) < 1e-12, (name, o.calc_reward(), want)
        assert abs(step_dur - 19 * pc.DT) < 1e-12        # HandleNewAction latches mTime after its first increment


@pytest.mark.parametrize("name,pack", [("amp_heading_zombie", 1), ("amp_target_zombie", 1), ("amp_heading_zombie", 2), ("amp_heading_clips4", 1), ("amp_heading_clips4", 2)])
def test_goal_scene_matches_oracle_emulator(emu_lib, name, pack):
    t = model.load_asset(name)
    w = pc.goal_rollout_compare(t, 64, emu_lib, steps=24, n=2, seed=5, wave_packing=pack)
    print(name, w)
    assert w["flags_ok"] and w["resets"] >= 1 and w["live"] >= 10
    tol = 1e-6 if name != "amp_heading_clips4" else 1e-3    # the 4-clip dataset starts some episodes lying on the ground (stiff contacts)
    assert w["reward"] < tol and w["goal"] < 10 * tol and w["goal_state"] < tol and w["state"] < max(1e-5, 50 * tol)
    assert w["kin"] < 50 * tol, w["kin"]      # the kinematic origin: cycle boundaries of the env's own clip (round 5: every env wrapped on clip 0's period before)


def test_kin_character_of_a_multi_clip_env_runs_on_its_own_clip(emu_lib):
    """long episodes over the 4-clip dataset: the origin of the kinematic character follows the simulated root at the cycle boundaries of the clip the env was
    reset to (two looping clips of 0.8 / 1.27 s, two non-looping get-up clips), as the oracle's (cKinCharacter on cClipsController's active motion)"""
    t = model.load_asset("amp_heading_clips4")
    t.cfg.time_lim_min = t.cfg.time_lim_max = 2.0; t.cfg.time_end_lim_min = t.cfg.time_end_lim_max = 2.0
    t.cfg.enable_fall_end = False          # (episodes run to the timer: several cycles of the looping clips)
    w = pc.goal_rollout_compare(t, 64, emu_lib, steps=80, n=3, seed=11, wave_packing=1)
    assert w["flags_ok"] and w["resets"] >= 3 and len(w["clips"]) >= 2, w
    assert w["kin"] < 5e-3, w["kin"]            # (inherits the simulated root's kernel-vs-oracle drift at each sync; a boundary taken on the wrong clip moves the origin by decimetres)


def test_target_distance_failure(emu_lib):
    """cSceneTargetAMP::CheckTerminate: Fail when the root is further than tar_fail_dist from the target (SceneTargetAMP.cpp:306-345)"""
    t = model.load_asset("amp_target_zombie")
    env = BatchEnv(t, 2, precision=64, lib_path=emu_lib, seed=2, wave_packing=1)
    env.reset()
    gs = env.get_goal_state()
    st = env.get_state()
    gs[1, 0] = st["pose"][1, 0] + 15.5           # env 1: target 15.5 m away (tar_fail_dist = 15)
    gs[1, 2] = st["pose"][1, 2]
    env.set_goal_state(gs)
    q = env.query()
    assert q["terminate"][0] == 0 and q["terminate"][1] == 1 and q["episode_end"][1] == 1 and q["reward"][1] == 0.0
    assert abs(q["goal"][1][2] - 15.5) < 1e-5
    out = env.step(np.zeros((2, env.A), np.float32), pc.DT, 20, auto_reset=True)
    assert out["terminate"][1] == 1 and out["goal"][1][2] <= t.cfg.max_target_dist + 1e-5     # reset drew a new target within max_target_dist


def test_rand_rot_reset_spreads_headings(emu_lib):
    t = model.load_asset("amp_heading_zombie")
    env = BatchEnv(t, 16, precision=64, lib_path=emu_lib, seed=4, wave_packing=1)
    env.reset()
    q = env.get_state()["pose"][:, 3:7]
    yaw = 2 * np.arctan2(q[:, 2], q[:, 0])
    assert yaw.std() > 0.8                        # uniform in [-pi, pi): std ~ 1.8; without the random rotation all 16 are equal
    assert (np.abs(env.get_state()["pose"][:, [0, 2]]) < 1e-12).all()     # SetCharRandPlacement keeps x, z at 0


def test_facade_goal_surface(emu_lib, monkeypatch):
    import os, sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepmimic_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from DeepMimicCore import DeepMimicCore
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    for name, scene_name in (("amp_heading_zombie", "Heading AMP"), ("amp_target_zombie", "Target AMP")):
        core = DeepMimicCore.cDeepMimicCore(False)
        core.SeedRand(3); core.LoadTables(model.load_asset(name), 10); core.Init()
        assert core.GetName() == scene_name and core.GetGoalSize(0) == 3 and core.EnableAMPTaskReward() is True
        assert core.BuildGoalOffset(0) == [0.0] * 3 and core.BuildGoalScale(0) == [1.0] * 3 and core.BuildGoalNormGroups(0) == [0] * 3
        g = core.RecordGoal(0)
        assert len(g) == 3 and np.isfinite(g).all() and core.GetAMPObsSize() > 0
        core.SetAction(0, [0.0] * core.GetActionSize(0))
        for _ in range(20):
            core.Update(1.0 / 600)
        assert 0.0 <= core.CalcReward(0) <= 1.0 and len(core.RecordAMPObsAgent(0)) == core.GetAMPObsSize()
        assert len(core.RecordAMPObsExpert(0)) == core.GetAMPObsSize()
        core.Shutdown()


def test_amp_expert_multi_clip(emu_lib, oracle_built):
    """RecordAMPObsExpert with a cClipsController: the sample's clip and time are given; device vs oracle"""
    t = model.load_asset("amp_heading_clips4")
    env = BatchEnv(t, 1, precision=64, lib_path=emu_lib, wave_packing=1)
    o = Oracle(t)
    clips = np.array([0, 1, 2, 3, 2], dtype=np.int32)
    times = np.array([0.3, 1.0, 2.5, 0.1, 0.02])
    got = env.amp_expert_clips(5, clips, times, 0.0)
    for i in range(5):
        want = o.amp_obs_expert_clip(int(clips[i]), float(times[i]), 0.0)
        assert np.abs(got[i] - want).max() < 2e-6, (i, np.abs(got[i] - want).max())


# ---- heading_amp_getup and strike_amp
def _getup_tables(recover=None, time_lim=None):
    t = model.load_asset("amp_heading_getup")
    if recover is not None:
        t.cfg.recover_episode_prob = recover
    if time_lim is not None:
        t.cfg.time_lim_min = t.cfg.time_lim_max = time_lim
    return t


def _strike_tables(variant):
    """the shipped strike keys with one knob turned so that random actions reach the branch under test within a few seconds"""
    t = model.load_asset("amp_strike_punch")
    t.cfg.time_lim_min = t.cfg.time_lim_max = 4.0
    if variant == "init_hit":            # episodes that start hit (ResetTargetHit) -> success 2 s after the drawn hit time
        t.cfg.init_hit_prob = 0.6
    elif variant == "hit":               # a large target sphere around a near target, any speed counts: CheckTargetHit fires
        t.cfg.target_radius = 1.0; t.cfg.hit_tar_speed = 0.0; t.cfg.fail_tar_contact_bodies = []; t.cfg.tar_far_prob = 0.0
    elif variant == "contact_fail":      # the same sphere with the forbidden bodies of the arg file: CheckTarContactFail
        t.cfg.target_radius = 1.0; t.cfg.tar_far_prob = 0.0
    elif variant == "test_succ":         # test mode: success pays the time left on the episode clock
        t.cfg.target_radius = 1.0; t.cfg.hit_tar_speed = 0.0; t.cfg.fail_tar_contact_bodies = []; t.cfg.tar_far_prob = 0.0
        t.cfg.time_end_lim_max = 4.0; t.cfg.target_hit_reset_time = 0.5
    return t


def test_new_scene_assets():
    g, k = model.load_asset("amp_heading_getup"), model.load_asset("amp_strike_punch")
    assert (g.goal_kind, g.goal_dim, k.goal_kind, k.goal_dim) == (3, 4, 4, 4)
    # args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt: getup_motion_ids 2 3, head_id 2, recover_episode_prob 0.2
    assert g.getup_clip_mask == 0b1100 and g.cfg.head_id == 2 and g.cfg.recover_episode_prob == 0.2 and g.cfg.getup_height_head == 1.3
    assert abs(g.getup_time - max(g.clip_duration(2), g.clip_duration(3))) < 1e-12 and g.getup_time > 3.0
    # args/train_amp_strike_humanoid3d_walk_punch_args.txt
    assert k.cfg.strike_bodies == [8] and k.cfg.fail_tar_contact_bodies == [0, 1, 2] and k.cfg.init_hit_prob == 0.1
    assert tuple(k.cfg.target_min) == (-0.5, 1.2, 0.6) and k.cfg.tar_near_dist == 1.4 and k.cfg.tar_fail_dist == 15.0


def test_getup_closed_form(oracle_built):
    """cSceneHeadingAMPGetup: an episode reset into a get-up clip is `getting up` from the clip time on (SyncGetupTimer), pays
    CalcRewardGetup = 0.2 clamp(root_h / 0.5) + 0.8 clamp(head_h / 1.3), reports phase 1 - t / getup_time, and cannot fall."""
    t = _getup_tables()
    o = Oracle(t); o.goal_rng(3, 0, 0); o.reset_ex(0.5, np.inf, 2, 0.0)              # clip 2 = getup_facedown: starts lying on the ground
    o.set_action(np.zeros(o.A)); o.control_step(20, pc.DT, end_early=False)
    gs = o.goal_state(full=True); p, _ = o.sim_state(); links = o.links()
    tm = 0.5 + 20 * pc.DT
    assert abs(gs[13] - tm) < 1e-12 and abs(o.record_goal()[3] - (1 - tm / t.getup_time)) < 1e-12
    want = 0.2 * np.clip(p[1] / 0.5, 0, 1) + 0.8 * np.clip(links[2, 1] / 1.3, 0, 1)
    assert abs(o.calc_reward() - want) < 1e-12 and 0 < want < 0.5
    assert any(c and f for c, f in zip(o.contacts(), t.fall_mask())) and o.check_terminate() == 0     # a fall body touches the ground, yet no fall while getting up
    o2 = Oracle(t); o2.goal_rng(3, 0, 0); o2.reset_ex(0.2, np.inf, 1, 0.0)           # clip 1 = walk: not getting up, phase 0
    assert o2.record_goal()[3] == 0.0 and o2.goal_state(full=True)[13] == t.getup_time


@pytest.mark.parametrize("pack", [1, 2])
def test_getup_recovery_episodes_emulator(emu_lib, pack):
    """train mode, recover_episode_prob 0.6, large action noise: falls end episodes, some of them continue as recovery episodes
    (timers + controller reset only, get-up timer restarted) -- device and oracle decide and continue identically"""
    w = pc.goal_rollout_compare(_getup_tables(recover=0.6, time_lim=4.0), 64, emu_lib, steps=45, n=2, seed=7, wave_packing=pack, action_sigma=0.6)
    print(pack, w)
    assert w["flags_ok"] and w["recoveries"] >= 1 and w["aux"] < 1e-9 and w["aux_steps"] >= 20
    assert w["reward_mean"] < 1e-4 and w["goal_state"] < 5e-3


def test_getup_test_mode_emulator(emu_lib):
    """test mode: a fall starts a get-up (UpdateTestGetup) instead of ending the episode"""
    w = pc.goal_rollout_compare(_getup_tables(), 64, emu_lib, steps=30, n=2, seed=5, wave_packing=2, test_mode=True)
    print(w)
    assert w["flags_ok"] and w["fail"] == 0 and w["resets"] == 0 and w["aux_steps"] >= 10 and w["aux"] < 1e-9 and w["reward_mean"] < 1e-4


@pytest.mark.parametrize("variant,pack,test_mode", [("init_hit", 1, False), ("hit", 2, False), ("contact_fail", 1, False), ("test_succ", 2, True)])
def test_strike_emulator(emu_lib, variant, pack, test_mode):
    w = pc.goal_rollout_compare(_strike_tables(variant), 64, emu_lib, steps=40, n=2, seed=3, wave_packing=pack, action_sigma=0.1, test_mode=test_mode)
    print(variant, w)
    assert w["flags_ok"] and w["aux"] < 1e-9 and w["reward_mean"] < 1e-4 and w["goal_state"] < 5e-3
    if variant in ("init_hit", "test_succ"):
        assert w["succ"] >= 1                      # eTerminateSucc reaches the caller
    if variant == "hit":
        assert w["aux_steps"] >= 20                # CheckTargetHit fired and the hit reward (1.0) was paid
    if variant == "contact_fail":
        assert w["fail"] >= 5


def test_facade_new_goal_scenes(emu_lib, monkeypatch):
    """GetGoalSize / BuildGoal* / RecordGoal / SetMode of the two scenes through the cDeepMimicCore facade"""
    import os, sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepmimic_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from DeepMimicCore import DeepMimicCore
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    for name, nm, groups in (("amp_heading_getup", "Heading AMP Getup", [0, 0, 0, -1]), ("amp_strike_punch", "Strike AMP", [-1] * 4)):
        core = DeepMimicCore.cDeepMimicCore(False)
        core.SeedRand(2); core.LoadTables(model.load_asset(name), 10); core.Init()
        assert core.GetName() == nm and core.GetGoalSize(0) == 4 and core.BuildGoalNormGroups(0) == groups
        if name == "amp_heading_getup":
            assert core.BuildGoalOffset(0) == [0.0, 0.0, 0.0, -0.5] and core.BuildGoalScale(0) == [1.0, 1.0, 1.0, 2.0]
        core.SetMode(core.eModeTest)
        g = core.RecordGoal(0)
        assert len(g) == 4 and np.isfinite(g).all() and 0.0 <= g[3] <= 1.0
        core.SetAction(0, [0.0] * core.GetActionSize(0))
        for _ in range(20):
            core.Update(1.0 / 600)
        assert np.isfinite(core.CalcReward(0)) and len(core.RecordGoal(0)) == 4
        core.Shutdown()


# ---- dribble_amp: a free rigid sphere next to the character
def _ball_setup(lib, prec, n=2, seed=4, pack=1):
    """device env + mirrored oracles after a reset at given clip times; the ball is then placed next to the character and kicked at it"""
    t = model.load_asset("amp_dribble_zombie")
    env = BatchEnv(t, n, precision=prec, lib_path=lib, wave_packing=pack, seed=seed)
    g0 = env.get_goal_state()
    kts = [0.2 + 0.3 * e for e in range(n)]
    env.reset(kin_times=kts)
    oracles = []
    for e in range(n):
        o = Oracle(t); o.goal_rng(seed, e, int(g0[e][11])); o.reset_ex(kts[e], np.inf, 0, 0.0); oracles.append(o)
    return t, env, oracles


def _kick_rollout(lib, prec, steps, pack=1):
    t, env, oracles = _ball_setup(lib, prec, pack=pack)
    st = env.get_state(); ob = env.get_obj_state()
    assert max(np.abs(ob[e] - o.ball_state()[:13]).max() for e, o in enumerate(oracles)) < 1e-6       # the ball reset draws agree
    for e, o in enumerate(oracles):
        root = st["pose"][e][:3]
        b = np.zeros(13); b[0:3] = [root[0] + 0.5, 0.25 + 0.1 * e, root[2] + 0.05]; b[3] = 1.0; b[7:10] = [-4.0, 0.5, 0.0]; b[10:13] = [0, 0, 3.0]
        ob[e] = b; o.set_ball(b)
    env.set_obj_state(ob)
    rng = np.random.default_rng(1)
    mx = dict(ball=0.0, reward=0.0, state=0.0, speed=0.0)
    for k in range(steps):
        acts = (0.05 * rng.normal(size=(len(oracles), env.A))).astype(np.float32)
        out = env.step(acts, pc.DT, 20)
        ob = env.get_obj_state()
        for e, o in enumerate(oracles):
            o.set_action(acts[e].astype(np.float64)); o.control_step(20, pc.DT, end_early=False)
            bs = o.ball_state()
            mx["ball"] = max(mx["ball"], np.abs(ob[e] - bs[:13]).max()); mx["reward"] = max(mx["reward"], abs(out["reward"][e] - o.calc_reward()))
            so = o.record_state(); mx["state"] = max(mx["state"], np.abs(out["state"][e] - so).max() / max(1.0, np.abs(so).max()))
            mx["speed"] = max(mx["speed"], float(np.abs(bs[7:10]).max()))
    return mx


def test_ball_physics_closed_form(oracle_built):
    """the free sphere of the oracle: free fall, rest on the ground at y = r, sliding turns into rolling without slipping
    (v = omega x r), damping (1 - 0.4)^t on both velocities"""
    t = model.load_asset("amp_dribble_zombie")
    assert (t.goal_kind, t.goal_dim, t.state_dim) == (5, 3, 226 + 15)
    o = Oracle(t); o.goal_rng(3, 0, 0); o.reset_ex(0.3, np.inf, 0, 0.0)
    b = o.ball_state()[:13].copy()
    assert abs(b[1] - 0.2) < 1e-12 and np.abs(b[7:13]).max() == 0 and abs(np.linalg.norm(b[3:7]) - 1) < 1e-12
    b[0] += 30.0; b[1] = 1.0; b[7:13] = [1.0, 0, 0, 0, 0, 0]; o.set_ball(b)                # far from the character
    o.set_action(np.zeros(o.A))
    for _ in range(120): o.update(pc.DT)                                               # 0.2 s of free fall
    bs = o.ball_state()
    assert abs(bs[8] - (-9.8 * 0.2)) < 0.25 and abs(bs[7] - 0.6 ** 0.2) < 1e-3       # v_y ~ -g t (with damping), v_x = 0.6^t
    for _ in range(1200): o.update(pc.DT)                                              # lands, slides, rolls
    bs = o.ball_state()
    assert abs(bs[1] - 0.2) < 2e-3 and abs(bs[8]) < 1e-3                               # rests on the ground
    assert bs[7] > 0.05 and abs(bs[12] + bs[7] / 0.2) < 1e-3 * abs(bs[12])              # rolling: omega_z = -v_x / r
    # the 15 task-state entries: position in the origin frame, unit normal / tangent, velocities
    ts = o.record_state()[-15:]
    assert abs(np.linalg.norm(ts[3:6]) - 1) < 1e-9 and abs(np.linalg.norm(ts[6:9]) - 1) < 1e-9 and abs(np.dot(ts[3:6], ts[6:9])) < 1e-9
    assert abs(np.linalg.norm(ts[9:12]) - np.linalg.norm(bs[7:10])) < 1e-9


@pytest.mark.parametrize("pack", [1, 2])
@pytest.mark.parametrize("test_mode", [False, True])
def test_dribble_scene_matches_oracle_emulator(emu_lib, test_mode, pack):
    t = model.load_asset("amp_dribble_zombie")
    w = pc.goal_rollout_compare(t, 64, emu_lib, steps=24, n=2 * pack, seed=5, wave_packing=pack, test_mode=test_mode)
    print(w)
    assert w["flags_ok"] and w["ball"] < 1e-9 and w["reward"] < 1e-6 and w["goal"] < 1e-5 and w["goal_state"] < 1e-6 and w["state"] < 1e-5
    assert w["resets"] >= 1 or test_mode


def test_dribble_scene_physics_2_matches_oracle_emulator(emu_lib):
    """round 5: dribble_amp under DM-physics v2 -- the links' ground contacts through the persistent manifolds (both limit rows, one new support
    point per narrowphase call), the ball's contacts single points as under v1 (a sphere's support point along -n is unique; scenes/SceneDribbleAMP.cpp:398-420)"""
    t = model.load_asset("amp_dribble_zombie")
    w = pc.goal_rollout_compare(t, 64, emu_lib, steps=24, n=2, seed=5, wave_packing=1, physics=2)
    print(w)
    assert w["flags_ok"] and w["ball"] < 1e-9 and w["reward"] < 1e-6 and w["goal"] < 1e-5 and w["goal_state"] < 1e-6 and w["state"] < 1e-5 and w["resets"] >= 1
    v1 = BatchEnv(t, 2, precision=64, lib_path=emu_lib, seed=5, physics=1); v2 = BatchEnv(t, 2, precision=64, lib_path=emu_lib, seed=5, physics=2)
    v1.reset(); v2.reset()
    for _ in range(10):
        a = v1.step(np.zeros((2, v1.A), np.float32), 1.0 / 600, 20); b = v2.step(np.zeros((2, v2.A), np.float32), 1.0 / 600, 20)
    assert v2.physics == 2 and v2.get_manifolds()[:, :, 0].sum() > 0 and not np.array_equal(a["state"], b["state"])      # it IS the other rigid-body step


@pytest.mark.parametrize("pack", [1, 2])
def test_ball_kick_matches_oracle_emulator(emu_lib, pack):
    """the ball thrown at the character's legs: contacts between the free body and the links, the shared constraint solve; one character per wavefront and (round 6)
    two -- the ball's contacts in the half's own slots, its velocity change by half-wave sums"""
    mx = _kick_rollout(emu_lib, 64, 8, pack=pack)
    print(mx)
    assert mx["speed"] > 5.0 and mx["ball"] < 1e-8 and mx["reward"] < 1e-6 and mx["state"] < 1e-5


def test_dribble_facade_and_refusals(emu_lib, monkeypatch):
    import os, sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepmimic_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from DeepMimicCore import DeepMimicCore
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    core = DeepMimicCore.cDeepMimicCore(False)
    core.SeedRand(2); core.LoadTables(model.load_asset("amp_dribble_zombie"), 10); core.Init()
    assert core.GetName() == "Dribble AMP" and core.GetGoalSize(0) == 3 and core.GetStateSize(0) == 241
    assert len(core.BuildStateOffset(0)) == 241 and core.BuildStateNormGroups(0)[-15:] == [0] * 15 and core.BuildStateScale(0)[-15:] == [1.0] * 15
    s0 = core.RecordState(0)
    assert len(s0) == 241 and np.isfinite(s0).all() and abs(np.linalg.norm(s0[-12:-9]) - 1) < 1e-5
    core.SetAction(0, [0.0] * core.GetActionSize(0))
    for _ in range(20):
        core.Update(1.0 / 600)
    assert np.isfinite(core.CalcReward(0)) and len(core.RecordGoal(0)) == 3
    core.Shutdown()
    with pytest.raises(RuntimeError, match="one character per wavefront"):      # (round 6: two per wave under DM-physics v1; v2 keeps the one-per-wave kernel)
        BatchEnv(model.load_asset("amp_dribble_zombie"), 2, lib_path=emu_lib, wave_packing=2, physics=2)


@pytest.mark.parametrize("asset", ["amp_dribble_zombie", "amp_heading_getup", "humanoid3d_walk"])
def test_snapshot_restore_reproduces_the_rollout(emu_lib, asset):
    """BatchEnv.snapshot / restore at an action boundary: the same actions give the same outputs bit for bit (goal row, scene block
    and the ball included)"""
    t = model.load_asset(asset)
    env = BatchEnv(t, 2, precision=64, lib_path=emu_lib, wave_packing=1, seed=9)
    env.reset()
    rng = np.random.default_rng(2)
    acts = [(0.2 * rng.normal(size=(2, env.A))).astype(np.float32) for _ in range(6)]
    for a in acts[:2]:
        env.step(a, pc.DT, 20, auto_reset=True)
    snap = env.snapshot()
    first = [env.step(a, pc.DT, 20, auto_reset=True, amp=bool(env.amp_size)) for a in acts[2:]]
    env.restore(snap)
    again = [env.step(a, pc.DT, 20, auto_reset=True, amp=bool(env.amp_size)) for a in acts[2:]]
    for x, y in zip(first, again):
        for k in x:
            assert np.array_equal(x[k], y[k]), (asset, k)
    env.close()


# ---- the HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("name,prec,pack", [("amp_heading_zombie", 64, 1), ("amp_target_zombie", 64, 2), ("amp_heading_zombie", 32, 2),
                                            ("amp_target_zombie", 32, 1), ("amp_heading_clips4", 64, 0)])
def test_goal_scene_matches_oracle_gpu(hip_lib, name, prec, pack):
    t = model.load_asset(name)
    w = pc.goal_rollout_compare(t, prec, hip_lib, steps=120, n=8, seed=6, wave_packing=pack)
    print(name, prec, pack, w)
    assert w["flags_ok"] or prec == 32
    assert w["resets"] >= 4 and w["live"] >= 100
    if name == "amp_heading_clips4":
        assert len(w["clips"]) >= 3 and w["goal_state"] < 1e-3
        assert w["kin"] < 5e-2, w["kin"]          # the kinematic origin (pack 0 = two per wavefront here): cycle boundaries of each env's own clip; follows the simulated root's drift
        return
    # free-running closed loop with random actions: the characters fall within a second, and two correct contact simulations separate
    # chaotically around a fall, so the MEAN errors carry the statement and the maxima are bounded loosely;
    # the goal generator itself (targets, headings, speeds, timers: goal_state) is exact up to the root position it samples around
    if prec == 64:
        assert w["reward_mean"] < 1e-5 and w["goal_mean"] < 1e-4 and w["reward"] < 5e-3 and w["goal"] < 5e-2 and w["goal_state"] < 5e-3
    else:
        assert w["reward_mean"] < 2e-4 and w["goal_mean"] < 2e-3 and w["reward"] < 5e-2 and w["goal_state"] < 5e-2


@pytest.mark.gpu
def test_goal_scenes_4096_auto_reset(hip_lib):
    """full-size property check: 4096 envs of each task scene, random actions, auto-reset; finite outputs, rewards in [0, 1],
    goal vectors well formed (heading: unit direction + speed in range; target: unit direction + distance <= tar_fail_dist)"""
    for name in ("amp_heading_zombie", "amp_target_zombie"):
        t = model.load_asset(name)
        env = BatchEnv(t, 4096, seed=8)
        env.reset()
        rng = np.random.default_rng(0)
        ends = 0
        for k in range(30):
            out = env.step((0.2 * rng.normal(size=(4096, env.A))).astype(np.float32), pc.DT, 20, auto_reset=True, amp=True)
            assert np.isfinite(out["state"]).all() and np.isfinite(out["goal"]).all() and np.isfinite(out["amp_obs"]).all()
            assert (out["reward"] >= 0).all() and (out["reward"] <= 1 + 1e-6).all()
            assert np.abs(np.linalg.norm(out["goal"][:, :2], axis=1) - 1).max() < 1e-4
            ends += int(out["episode_end"].sum())
        assert ends > 0 and out["reward"].mean() > 0.01
        if t.goal_kind == 2:
            assert (out["goal"][:, 2] >= t.cfg.tar_speed_min - 1e-6).all() and (out["goal"][:, 2] <= t.cfg.tar_speed_max + 1e-6).all()
        else:
            assert (out["goal"][:, 2] <= t.cfg.tar_fail_dist + 1e-3).all()


@pytest.mark.gpu
@pytest.mark.parametrize("prec,pack", [(64, 1), (32, 2)])
def test_getup_scene_gpu(hip_lib, prec, pack):
    """heading_amp_getup on the HIP kernels: recovery episodes in train mode, get-up on a fall in test mode"""
    # a recovery episode continues from wherever the fall left the character: unlike a scene reset it does not re-synchronise two
    # free-running simulations, so deviations accumulate over chained recovery episodes (large action noise, stiff ground contact of a
    # lying character); the decisions -- which falls recover, the get-up clock -- are what is asserted exactly, the reward by its mean
    w = pc.goal_rollout_compare(_getup_tables(recover=0.6, time_lim=4.0), prec, hip_lib, steps=60, n=8, seed=7, wave_packing=pack, action_sigma=0.6)
    print("train", prec, pack, w)
    assert (w["flags_ok"] or prec == 32) and w["recoveries"] >= 2 and w["aux_steps"] >= 100
    assert w["reward_mean"] < (5e-3 if prec == 64 else 2e-2) and (w["aux"] < 1e-9 or prec == 32)
    w = pc.goal_rollout_compare(_getup_tables(), prec, hip_lib, steps=100, n=8, seed=5, wave_packing=pack, test_mode=True)
    print("test", prec, pack, w)
    assert (w["flags_ok"] or prec == 32) and w["aux_steps"] >= 50 and w["reward_mean"] < (1e-4 if prec == 64 else 2e-3)
    if prec == 64:
        assert w["fail"] == 0 and w["resets"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("variant,prec,pack,test_mode", [("init_hit", 64, 2, False), ("hit", 32, 2, False), ("contact_fail", 32, 1, False), ("test_succ", 64, 1, True),
                                                         ("plain", 32, 2, False)])
def test_strike_scene_gpu(hip_lib, variant, prec, pack, test_mode):
    w = pc.goal_rollout_compare(_strike_tables(variant), prec, hip_lib, steps=120, n=8, seed=3, wave_packing=pack, action_sigma=0.1, test_mode=test_mode)
    print(variant, prec, pack, w)
    # fp32: an episode may end one update earlier / later than the oracle's (a borderline contact or hit test); such an env is not
    # scored from there on (parity_common), at most a quarter of the 8 x 120 transitions may be lost that way
    assert w["flags_ok"] or (prec == 32 and w["scored"] >= 720)
    assert w["reward_mean"] < (1e-4 if prec == 64 else 2e-3) and w["aux"] < (1e-6 if prec == 64 else 0.05)
    if variant in ("init_hit", "test_succ"):
        assert w["succ"] >= 2
    if variant == "hit":
        assert w["aux_steps"] >= 100
    if variant == "contact_fail":
        assert w["fail"] >= 10


@pytest.mark.gpu
def test_new_goal_scenes_4096(hip_lib):
    """4096 envs of heading_amp_getup and strike_amp, random actions, auto-reset: finite outputs, rewards in range, goal vectors well formed"""
    for name in ("amp_heading_getup", "amp_strike_punch"):
        t = model.load_asset(name)
        env = BatchEnv(t, 4096, seed=8)
        env.reset()
        rng = np.random.default_rng(0)
        ends = 0
        for k in range(30):
            out = env.step((0.2 * rng.normal(size=(4096, env.A))).astype(np.float32), pc.DT, 20, auto_reset=True, amp=True)
            assert out["goal"].shape == (4096, 4) and np.isfinite(out["state"]).all() and np.isfinite(out["goal"]).all() and np.isfinite(out["amp_obs"]).all()
            assert (out["reward"] >= 0).all() and (out["reward"] <= 1 + 1e-6).all()
            assert (out["goal"][:, 3] >= 0).all() and (out["goal"][:, 3] <= 1).all()
            ends += int(out["episode_end"].sum())
        assert ends > 0 and out["reward"].mean() > 0.01
        if t.goal_kind == 3:
            assert np.abs(np.linalg.norm(out["goal"][:, :2], axis=1) - 1).max() < 1e-4 and (out["goal"][:, 3] > 0).mean() > 0.2   # half the clips are get-ups


@pytest.mark.gpu
@pytest.mark.parametrize("pack", [1, 2])
@pytest.mark.parametrize("prec", [64, 32])
def test_dribble_scene_gpu(hip_lib, prec, pack):
    """dribble_amp on the HIP kernels: the scene through resets, and the ball kicked at the character; one character per wavefront (k_env_step<ClsBipedObj>) and two
    (round 6: k_env_step_duo<..., ClsBipedObj>, the default)"""
    t = model.load_asset("amp_dribble_zombie")
    w = pc.goal_rollout_compare(t, prec, hip_lib, steps=60, n=8, seed=5, wave_packing=pack)
    print(prec, pack, w)
    assert (w["flags_ok"] or (prec == 32 and w["scored"] >= 360)) and w["resets"] >= 2
    # (the ball is re-placed around the root of the previous episode's last state: its error follows the free-running root's)
    assert w["reward_mean"] < (1e-5 if prec == 64 else 2e-3) and w["ball"] < (1e-3 if prec == 64 else 2e-2)
    mx = _kick_rollout(hip_lib, prec, 8, pack=pack)
    print(prec, pack, mx)
    assert mx["speed"] > 5.0 and mx["ball"] < (1e-6 if prec == 64 else 5e-2) and mx["reward"] < (1e-5 if prec == 64 else 5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [64, 32])
def test_dribble_scene_physics_2_gpu(hip_lib, prec):
    """dribble_amp under DM-physics v2 on the HIP kernels (k_env_step<ClsBipedObj, ..., PHYS2>): bounds of test_dribble_scene_gpu (fp32: the v2 state caveat of
    tests/test_physics_v2.py applies -- rewards, goals and the ball are held)"""
    t = model.load_asset("amp_dribble_zombie")
    w = pc.goal_rollout_compare(t, prec, hip_lib, steps=60, n=8, seed=5, wave_packing=1, physics=2)
    print(prec, w)
    assert (w["flags_ok"] or (prec == 32 and w["scored"] >= 360)) and w["resets"] >= 2
    assert w["reward_mean"] < (1e-5 if prec == 64 else 2e-3) and w["ball"] < (1e-3 if prec == 64 else 5e-2)


@pytest.mark.gpu
def test_dribble_4096(hip_lib):
    """4096 envs of dribble_amp, random actions, auto-reset: finite outputs, rewards in [0, 1], unit goal directions, balls on or above the ground"""
    t = model.load_asset("amp_dribble_zombie")
    env = BatchEnv(t, 4096, seed=8)
    env.reset()
    rng = np.random.default_rng(0)
    ends = 0
    for k in range(30):
        out = env.step((0.2 * rng.normal(size=(4096, env.A))).astype(np.float32), pc.DT, 20, auto_reset=True, amp=True)
        assert out["state"].shape == (4096, 241) and np.isfinite(out["state"]).all() and np.isfinite(out["goal"]).all()
        assert (out["reward"] >= 0).all() and (out["reward"] <= 1 + 1e-6).all()
        assert np.abs(np.linalg.norm(out["goal"][:, :2], axis=1) - 1).max() < 1e-4
        ends += int(out["episode_end"].sum())
    ob = env.get_obj_state()
    assert ends > 0 and np.isfinite(ob).all() and (ob[:, 1] > 0.15).all() and np.abs(np.linalg.norm(ob[:, 3:7], axis=1) - 1).max() < 1e-4
    assert np.abs(ob[:, 7:10]).max() > 0.1           # some balls have been kicked
