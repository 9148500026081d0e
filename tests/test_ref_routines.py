"""The reference's OWN routines, compiled (not composed): sim/CtPDController.cpp / sim/ImpPDController.cpp (stable PD and action
mapping), sim/CtController.cpp (RecordState), scenes/SceneImitate.cpp (CalcRewardImitate), built unmodified against oracle/bullet_stub
and run on the stand-in character of oracle/ref_standins.cpp -- held against (a) the oracle restatement and (b) the compositions of
oracle/ref_glue.cpp that used to be the only witness ("next" 5).  CPU, this container only (needs the reference's
controller / character / motion files; the prebuilt library alone is not enough)."""
import os

import numpy as np
import pytest

import ref_lib
from deepmimic_amd import model
from oracle_lib import Oracle
from ref_lib import Components, RefRig, Skel, random_pose_vel

pytestmark = pytest.mark.skipif(not os.path.isdir(ref_lib.REF_DATA), reason="needs /root/reference/data (controller / character files)")

CASES = [("humanoid3d", "humanoid3d_walk"), ("dog3d", "dog3d_pace")]
N = 300


@pytest.fixture(scope="module")
def ref(oracle_built):
    return Components("ref")


def _close(a, b, rtol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert err <= rtol, "%s: %.3e > %.1e" % (what, err, rtol)
    return err


def _gains(t):
    P = t.pose_dim
    kp, kd = np.zeros(P), np.zeros(P)
    for j in range(1, t.num_joints):
        off = int(t.joint_mat[j, model.JD_PARAM_OFFSET]); sz = model.joint_param_size(int(t.joint_mat[j, model.JD_TYPE]), False)
        kp[off:off + sz] = t.pd_params[j, 0]; kd[off:off + sz] = t.pd_params[j, 1]
    return kp, kd


@pytest.mark.parametrize("char,asset", CASES)
def test_action_mapping_routine(ref, char, asset):
    """cCtPDController::ApplyAction -> ConvertActionToTargetPose -> SetTargetTheta (sim/CtPDController.cpp:97-166), incl. exp maps longer
    than 2 pi (clamped) and near 0, vs the oracle's set_action."""
    t = model.load_asset(asset); rig = RefRig(ref, char); o = Oracle(t); o.reset(0.0)
    assert (rig.P, rig.S, rig.A) == (t.pose_dim, o.S, o.A)
    rng = np.random.default_rng(31)
    for i in range(N):
        a = rng.normal(size=rig.A) * rng.choice([1e-9, 0.3, 2.0, 9.0])
        tar = rig.apply_action(a)
        o.set_action(a)
        _close(tar[7:], o.tar_pose()[7:], 1e-13, "%s PD targets %d" % (char, i))


@pytest.mark.parametrize("char,asset", CASES)
def test_spd_routine(ref, char, asset):
    """cCtPDController::UpdateBuildTau -> cImpPDController::UpdateControlForce (UpdateRBDModel + CalcControlForces,
    sim/ImpPDController.cpp:47-73,129-195) with the gains the reference itself loads from the controller file, vs the composition
    (same ingredients called in the builder's order) and, through it, the oracle (tests/test_oracle_vs_ref.py holds those two together)."""
    t = model.load_asset(asset); rig = RefRig(ref, char); sr = Skel(ref, t)
    kp, kd = _gains(t)
    rng = np.random.default_rng(32)
    worst = 0.0
    for i in range(N):
        p, v = random_pose_vel(t, rng)
        rig.set_state(p, v)
        tar = rig.apply_action(rng.normal(size=rig.A) * 0.7)
        tau = rig.spd_tau(1 / 600)
        worst = max(worst, _close(tau, sr.spd_tau(p, v, tar, kp, kd, 1 / 600), 1e-12, "%s SPD torque %d" % (char, i)))
        assert np.all(tau[:7] == 0)
    print("%s: worst |routine - composition| / scale = %.2e" % (char, worst))


@pytest.mark.parametrize("char,asset", CASES)
def test_spd_routine_vs_oracle(ref, char, asset):
    """the same routine, clamped per joint as cSimBodyJoint::ClampTotalTorque does (parity_common.clamp_tau), directly against the oracle's
    SPD torque on states near the clip"""
    import parity_common as pc
    t = model.load_asset(asset); rig = RefRig(ref, char); o = Oracle(t)
    rng = np.random.default_rng(33)
    for i in range(100):
        o.reset(rng.uniform(0, o.duration))
        p, v = o.sim_state(); v = v + 0.3 * rng.normal(size=v.shape) * (v != 0)
        a = o.pose_to_action(p) + 0.1 * rng.normal(size=o.A)
        o.set_sim_state(p, v); o.set_action(a)
        rig.set_state(p, v); rig.apply_action(a)
        tau_r, tau_o = pc.clamp_tau(t, rig.spd_tau(1 / 600)), o.spd_tau(1 / 600)
        _close(tau_r, tau_o, 1e-10, "%s SPD torque vs oracle %d" % (char, i))


@pytest.mark.parametrize("char,asset", CASES)
def test_record_state_routine(ref, char, asset):
    """cCtController::RecordState (BuildStatePhase / BuildStatePose / BuildStateVel, sim/CtController.cpp:281-293,373-478) with the flags
    of the shipped controller file, vs the oracle's record_state and the learner tables (offset / scale / norm groups, action bounds)
    vs the oracle's."""
    t = model.load_asset(asset); rig = RefRig(ref, char); o = Oracle(t)
    rng = np.random.default_rng(34)
    for i in range(N):
        tt = rng.uniform(0, 2 * o.duration)
        o.reset(tt)
        p, v = random_pose_vel(t, rng)
        o.set_sim_state(p, v)
        rig.set_state(p, v)
        s_r, s_o = rig.record_state(o.phase(), 0.0), o.record_state()
        _close(s_r, s_o, 1e-12, "%s state vector %d" % (char, i))


@pytest.mark.parametrize("char,asset", CASES)
def test_learner_tables_routine(ref, char, asset, emu_lib):
    """BuildStateOffsetScale / BuildStateNormGroups / BuildActionOffsetScale / BuildActionBounds of the reference's controller object
    vs what the PRODUCT's host code hands the learner (dm_build_offsets_scales)."""
    from deepmimic_amd.core import BatchEnv
    t = model.load_asset(asset); rig = RefRig(ref, char)
    tb = rig.tables()
    env = BatchEnv(t, 1, lib_path=emu_lib, precision=64)
    m = env.offsets_scales()
    for k_ref, k_env in (("s_off", "state_offset"), ("s_scale", "state_scale"), ("a_off", "action_offset"), ("a_scale", "action_scale"),
                         ("a_min", "action_min"), ("a_max", "action_max")):
        _close(tb[k_ref], m[k_env], 1e-14, "%s %s" % (char, k_ref))
    assert np.array_equal(tb["s_groups"], m["state_norm_groups"])


@pytest.mark.parametrize("char,asset", CASES + [("humanoid3d", "humanoid3d_spinkick")])
def test_reward_routine(ref, char, asset):
    """cSceneImitate::CalcRewardImitate (scenes/SceneImitate.cpp:7-127) as compiled, joint weights by CalcJointWeights (:236-248), the
    reference's own kinematic character on the clip -- vs the oracle scene's calc_reward on the same (sim state, clip time, kin origin)."""
    t = model.load_asset(asset); rig = RefRig(ref, char, motion=asset); o = Oracle(t)
    rng = np.random.default_rng(35)
    worst = 0.0
    for i in range(120):
        tt = rng.uniform(0, 2.5 * o.duration)
        o.reset(tt)
        p, v = o.sim_state()
        v = v + rng.normal(size=v.shape) * (v != 0) * 0.5
        for j in range(1, t.num_joints):
            off, ty = int(t.joint_mat[j, model.JD_PARAM_OFFSET]), int(t.joint_mat[j, model.JD_TYPE])
            if ty == model.JT_SPHERICAL:
                q = p[off:off + 4] + 0.15 * rng.normal(size=4); q /= np.linalg.norm(q); p[off:off + 4] = q if q[0] >= 0 else -q
            elif ty == model.JT_REVOLUTE:
                p[off] += 0.1 * rng.normal()
        p[0:3] += rng.normal(size=3) * 0.05; p[1] += 0.3
        o.set_sim_state(p, v)
        ps, vs = o.sim_state(); kp_, kv_, ko = o.kin_state()
        rig.set_state(ps, vs)
        rig.kin_set(o.kin_time(), ko[:3], ko[3:7])
        kpr, kvr = rig.kin_state()
        _close(kpr, kp_, 1e-12, "kin pose"); _close(kvr, kv_, 1e-11, "kin vel")
        r_ref, r_o = rig.reward_imitate(0.0), o.calc_reward()
        worst = max(worst, abs(r_ref - r_o))
        assert abs(r_ref - r_o) < 1e-12, (asset, i, r_ref, r_o)
    print("%s: worst |CalcRewardImitate - oracle| = %.2e" % (asset, worst))


@pytest.mark.parametrize("local_root", [False, True])
def test_amp_obs_routine(ref, local_root):
    """cSceneImitateAMP::RecordAMPObsAgent -> BuildAMPObs (scenes/SceneImitateAMP.cpp:101-113,279-396) as compiled, both
    enable_amp_obs_local_root modes, vs the oracle's agent observation through a rollout (history = state at the last action latch)."""
    import parity_common as pc
    t = model.load_asset("humanoid3d_walk"); t.cfg.scene_amp = True; t.cfg.enable_amp_obs_local_root = local_root
    rig = RefRig(ref, "humanoid3d"); o = Oracle(t)
    rng = np.random.default_rng(36)
    for i in range(40):
        o.reset(rng.uniform(0, o.duration))
        for _ in range(int(rng.integers(1, 4))):
            kp, _, _ = o.kin_state(); o.set_action(o.pose_to_action(kp) + 0.2 * rng.normal(size=o.A)); o.control_step(20, pc.DT, end_early=False)
        p, v = o.sim_state(); pp, pv = o.prev_state()
        rig.set_state(p, v)
        a_r, a_o = rig.amp_obs(pp, pv, local_root), o.amp_obs_agent()
        assert a_r.shape == a_o.shape == (226,)
        _close(a_r, a_o, 1e-12, "AMP observation %d" % i)


@pytest.mark.parametrize("name,kind", [("amp_target_zombie", 1), ("amp_heading_zombie", 2)])
def test_task_scene_routines(ref, name, kind):
    """cSceneTargetAMP / cSceneHeadingAMP CalcReward and RecordGoal (scenes/SceneTargetAMP.cpp:3-81,192-218; SceneHeadingAMP.cpp:3-43,
    134-149) as compiled, vs the oracle's task reward and goal vector at action boundaries of a closed-loop rollout (targets, headings and
    speeds drawn by the oracle's goal generator)."""
    import parity_common as pc
    t = model.load_asset(name); c = t.cfg
    rig = RefRig(ref, "humanoid3d"); o = Oracle(t)
    rng = np.random.default_rng(37)
    n = 0
    for ep in range(12):
        o.goal_rng(9, ep, 0); t0 = rng.uniform(0, 1.0); o.reset_ex(t0, np.inf, 0, rng.uniform(-3, 3))
        for k in range(8):
            o.set_action(0.3 * rng.normal(size=o.A)); o.control_step(20, pc.DT, end_early=False)
            if o.check_terminate() != 0:
                break
            gs = o.goal_state(); p, v = o.sim_state()
            rig.set_state(p, v)
            ctrl_time = gs[10] + 19 * pc.DT          # cDeepMimicCharController::GetTime() at the boundary: 19 updates after the latch
            par = [gs[0], gs[1], gs[2], gs[4], c.target_succ_dist, c.tar_fail_dist if np.isfinite(c.tar_fail_dist) else 1e30, float(c.enable_min_tar_vel), c.pos_reward_scale,
                   gs[3], c.vel_reward_scale, gs[10], gs[7], gs[8], gs[9], ctrl_time, 0.0]
            r_ref, g_ref = rig.task_scene(kind, par)
            assert abs(r_ref - o.calc_reward()) < 1e-12, (name, ep, k, r_ref, o.calc_reward())
            _close(g_ref, o.record_goal(), 1e-12, "%s goal %d/%d" % (name, ep, k))
            n += 1
    assert n >= 40
    par[15] = 1.0                                     # a fallen character: task reward 0 (SceneTargetAMP.cpp:22-24, SceneHeadingAMP.cpp:11-13)
    assert rig.task_scene(kind, par)[0] == 0.0
