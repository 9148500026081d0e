"""Checks against tests/golden/ref_vectors.npz -- outputs of the reference's OWN compiled sources (oracle/_ref, generator
tests/golden/make_ref_golden.py).  The file is committed, so these run where /root/reference does not exist:

* the oracle restatement vs the golden vectors (CPU),
* the product's device + host sources in the CPU emulator build vs the golden vectors (CPU; no oracle in the loop),
* marked gpu: the HIP kernels through the C-ABI vs the golden vectors, fp64 algorithm build and fp32 production build.
"""
import numpy as np
import pytest

import parity_common as pc
from deepmimic_amd import model
from oracle_lib import Oracle

SCENES = ["humanoid3d_walk", "dog3d_pace", "humanoid3d_spinkick"]


def _G(name, k):
    return pc.ref_golden()["%s/%s" % (name, k)]


def test_golden_file_is_reference_generated():
    g = pc.ref_golden()
    for name in SCENES:
        assert g["%s/pose" % name].shape[0] == 8 and g["%s/H" % name].shape[1] == model.load_asset(name).pose_dim
    assert g["math/op"].size == 640


@pytest.mark.parametrize("name", SCENES)
def test_oracle_vs_ref_golden(oracle_built, name):
    """Oracle Scene (the object the GPU parity tests compare with) vs the reference vectors: kin sample, H, C, SPD torque,
    state vector, reward terms, reward."""
    t = model.load_asset(name)
    o = Oracle(t)
    idx = pc.dof_index(t)
    n = _G(name, "pose").shape[0]
    import ctypes as C
    for e in range(n):
        tk, org = float(_G(name, "kin_time")[e]), _G(name, "kin_origin")[e]
        o.reset(tk)
        o.lib.orc_set_kin_origin(o.h, org[:3].ctypes.data_as(C.POINTER(C.c_double)), np.ascontiguousarray(org[3:]).ctypes.data_as(C.POINTER(C.c_double)))
        o.lib.orc_kin_set_time(o.h, C.c_double(tk))
        kp, kv, ko = o.kin_state()
        assert np.abs(kp - _G(name, "kin_pose")[e]).max() < 1e-12
        assert np.abs(kv - _G(name, "kin_vel")[e]).max() < 1e-11 * max(1.0, np.abs(kv).max())
        p, v = _G(name, "pose")[e], _G(name, "vel")[e]
        o.set_sim_state(p, v)
        ps, vs = o.sim_state()
        assert np.abs(ps - p).max() < 1e-15 and np.array_equal(vs, v)      # golden states are already in reported form (up to the renormalisation ulp)
        H, Cb = o.mass_bias(0, p, v)
        assert np.abs(H - _G(name, "H")[e]).max() < 1e-12 * np.abs(H).max()
        assert np.abs(Cb - _G(name, "C")[e]).max() < 1e-12 * max(1.0, np.abs(Cb).max())
        # SPD torque through the scene path (targets latched, clamp applied)
        o.lib.orc_set_tar_pose(o.h, np.ascontiguousarray(_G(name, "tar")[e]).ctypes.data_as(C.POINTER(C.c_double)))
        tau = o.spd_tau(pc.DT)
        tr = pc.clamp_tau(t, _G(name, "spd_tau")[e])
        assert np.abs(tau - tr).max() < 1e-10 * max(1.0, np.abs(tr).max())
        # state vector and reward (ctrl clock = kin time on a fresh reset)
        st = o.record_state()
        assert np.abs(st - _G(name, "state")[e]).max() < 1e-11 * max(1.0, np.abs(st).max())
        r, terms = o.calc_reward_terms()
        assert np.abs(terms - _G(name, "reward_terms")[e]).max() < 1e-11 * max(1.0, np.abs(terms).max())
        assert abs(r - float(_G(name, "reward")[e])) < 1e-12


def test_oracle_amp_obs_vs_ref_golden(oracle_built):
    """the oracle's agent observation vs cSceneImitateAMP::BuildAMPObs as compiled (golden `amp_obs`: pose / vel of the state, history one
    control period back on the clip)"""
    for name in SCENES:
        t = model.load_asset(name); t.cfg.scene_amp = True; t.cfg.enable_amp_obs_local_root = False
        o = Oracle(t)
        for e in range(_G(name, "pose").shape[0]):
            o.reset(0.0); o.set_sim_state(_G(name, "pose")[e], _G(name, "vel")[e]); o.set_prev_state(_G(name, "amp_prev_pose")[e], _G(name, "amp_prev_vel")[e])
            a = o.amp_obs_agent()
            assert a.shape == _G(name, "amp_obs")[e].shape and np.abs(a - _G(name, "amp_obs")[e]).max() < 1e-12 * max(1.0, np.abs(a).max())


def _task_device_check(lib_path, precision, tol_r, tol_g):
    """cSceneTargetAMP / cSceneHeadingAMP CalcReward + RecordGoal as compiled from the reference (golden `task/...`) vs the device path:
    state, clocks and the goal row are set through the C-ABI, the answers come from dm_query / dm_query_goal."""
    from deepmimic_amd.core import BatchEnv
    g = pc.ref_golden()
    worst = {}
    for name in ("amp_target_zombie", "amp_heading_zombie"):
        t = model.load_asset(name)
        P, V, par = g["task/%s/pose" % name], g["task/%s/vel" % name], g["task/%s/par" % name]
        n = P.shape[0]
        env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=1)
        env.reset(kin_times=np.zeros(n), max_times=np.inf)
        clocks = np.stack([np.zeros(n), par[:, 14], np.zeros(n), np.zeros(n), np.full(n, np.inf)], axis=1)
        flags = np.tile(np.array([[1, 0, 1, 1]], dtype=np.int32), (n, 1))
        st = env.get_state()
        env.set_state(pose=P, vel=V, tar=st["tar"], kin=st["kin"], clocks=clocks, flags=flags)
        gs = env.get_goal_state()
        gs[:, 0:3] = par[:, 0:3]; gs[:, 3] = par[:, 8]; gs[:, 4] = par[:, 3]; gs[:, 5] = 0.0; gs[:, 6] = 1e9
        gs[:, 7:10] = par[:, 11:14]; gs[:, 10] = par[:, 10]
        env.set_goal_state(gs)
        q = env.query(); goals = env.query_goal()
        dr = np.abs(q["reward"] - g["task/%s/reward" % name]).max(); dg = np.abs(goals - g["task/%s/goal" % name]).max()
        worst[name] = (float(dr), float(dg))
        assert dr < tol_r and dg < tol_g, (name, dr, dg)
        assert (g["task/%s/reward" % name] > 0).sum() >= n // 2       # informative: most cases carry a nonzero task reward
    return worst


def _strike_dribble_device_check(lib_path, precision, tol_r, tol_g, tol_s):
    """cSceneStrikeAMP / cSceneDribbleAMP as compiled from the reference (golden `task/amp_strike_punch`, `task/amp_dribble_zombie`: CalcReward in
    train mode, RecordGoal, CheckTargetHit / CheckTarContactFail / CheckTarHitSucc, CheckTargetSucc / the two distance failures / HasFallen, and
    the 15 task entries of the dribble state vector) vs the device path: pose, clocks, goal row, hit record and ball through the C-ABI, the
    answers from dm_query / dm_query_goal -- and, for the hit test, which the device runs inside an update, from the hit record after an update
    of 1e-7 s."""
    from deepmimic_amd.core import BatchEnv
    g = pc.ref_golden()
    worst = {}
    for name in ("amp_strike_punch", "amp_dribble_zombie"):
        t = model.load_asset(name)
        P, V, par, extra = (g["task/%s/%s" % (name, k)] for k in ("pose", "vel", "par", "extra"))
        n = P.shape[0]
        strike = name == "amp_strike_punch"
        env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=1)
        env.reset(kin_times=np.zeros(n), max_times=np.inf)
        clocks = np.stack([np.zeros(n), par[:, 14], np.zeros(n), par[:, 24] if strike else np.zeros(n), np.full(n, np.inf)], axis=1)
        flags = np.tile(np.array([[1, 0, 1, 1]], dtype=np.int32), (n, 1))
        fall_bit = int(np.flatnonzero(t.fall_mask())[0])
        flags[:, 1] = np.where(par[:, 15] != 0, 1 << fall_bit, 0)                  # "fallen" = a fall-contact body touches the ground
        st = env.get_state()
        env.set_state(pose=P, vel=V, tar=st["tar"], kin=st["kin"], clocks=clocks, flags=flags)
        gs = env.get_goal_state()
        gs[:, 0:3] = par[:, 0:3]; gs[:, 3] = 0.0; gs[:, 4] = par[:, 3]; gs[:, 5] = 0.0; gs[:, 6] = 1e9
        gs[:, 7:10] = par[:, 11:14]; gs[:, 10] = par[:, 10]
        env.set_goal_state(gs)
        aux = env.get_goal_aux()
        if strike: aux[:, 0] = par[:, 22]; aux[:, 1] = par[:, 23]
        else:
            aux[:, 2:5] = par[:, 29:32]; aux[:, 5] = 0.0; aux[:, 6] = 1e9
            env.set_obj_state(par[:, 16:29])
        env.set_goal_aux(aux)
        q = env.query(); goals = env.query_goal()
        dr = np.abs(q["reward"] - g["task/%s/reward" % name]).max(); dg = np.abs(goals - g["task/%s/goal" % name]).max()
        assert dr < tol_r and dg < tol_g, (name, dr, dg)
        if strike:
            check_hit, contact_fail, hit_succ = extra[:, 0] != 0, extra[:, 1] != 0, extra[:, 2] != 0
            # the fall test of cRLSceneSimChar::CheckTerminate, then cSceneStrikeAMP::CheckTerminateTarget (:527-545); no distance failure here
            want = np.where((par[:, 15] != 0) | contact_fail, 1, np.where(hit_succ, 2, 0))
            assert np.array_equal(q["terminate"], want), (q["terminate"], want)
            assert contact_fail.sum() >= 2 and hit_succ.sum() >= 2 and check_hit.sum() >= 1
            env.update(1e-7, 1)
            hit_after = env.get_goal_aux()[:, 0] != 0
            assert np.array_equal(hit_after, (par[:, 22] != 0) | check_hit), (hit_after, check_hit)
            worst[name] = (float(dr), float(dg))
        else:
            succ, tar_fail, char_fail, fallen = (extra[:, 15 + k] != 0 for k in range(4))
            ds = np.abs(q["state"][:, -15:] - extra[:, :15]).max()
            assert ds < tol_s, ds
            fell = par[:, 15] != 0
            want = np.where(fell | tar_fail | char_fail, 1, np.where(succ, 2, 0))  # fall test on cSceneDribbleAMP::HasFallen, then CheckTerminateTarget (:343-349, 468-477)
            assert np.array_equal(fallen, fell | tar_fail | char_fail)
            assert np.array_equal(q["terminate"], want), (q["terminate"], want)
            assert succ.sum() >= 2 and tar_fail.sum() >= 1 and char_fail.sum() >= 1
            worst[name] = (float(dr), float(dg), float(ds))
    return worst


def _getup_device_check(lib_path, precision, tol_r, tol_g):
    """cSceneHeadingAMPGetup as compiled from the reference (golden `task/amp_heading_getup`): the get-up reward while the get-up timer runs and the
    heading reward otherwise, the goal with the get-up phase, no contact fall while getting up -- vs dm_query / dm_query_goal"""
    from deepmimic_amd.core import BatchEnv
    g = pc.ref_golden()
    name = "amp_heading_getup"
    t = model.load_asset(name)
    P, V, par, extra = (g["task/%s/%s" % (name, k)] for k in ("pose", "vel", "par", "extra"))
    n = P.shape[0]
    assert abs(t.getup_time - par[0, 16]) < 1e-12
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=1)
    env.reset(kin_times=np.zeros(n), max_times=np.inf)
    clocks = np.stack([np.zeros(n), par[:, 14], np.zeros(n), np.zeros(n), np.full(n, np.inf)], axis=1)
    flags = np.tile(np.array([[1, 0, 1, 1]], dtype=np.int32), (n, 1))
    flags[:, 1] = np.where(par[:, 15] != 0, 1 << int(np.flatnonzero(t.fall_mask())[0]), 0)
    st = env.get_state()
    env.set_state(pose=P, vel=V, tar=st["tar"], kin=st["kin"], clocks=clocks, flags=flags)
    gs = env.get_goal_state()
    gs[:, 0:3] = 0.0; gs[:, 3] = par[:, 8]; gs[:, 4] = par[:, 3]; gs[:, 5] = 0.0; gs[:, 6] = 1e9
    gs[:, 7:10] = par[:, 11:14]; gs[:, 10] = par[:, 10]
    env.set_goal_state(gs)
    aux = env.get_goal_aux(); aux[:, 0] = par[:, 17]; env.set_goal_aux(aux)
    q = env.query(); goals = env.query_goal()
    getting_up, fallen_contact = extra[:, 0] != 0, extra[:, 1] != 0
    dr = np.abs(q["reward"] - g["task/%s/reward" % name]).max(); dg = np.abs(goals - g["task/%s/goal" % name]).max()
    assert dr < tol_r and dg < tol_g, (dr, dg)
    assert np.array_equal(q["terminate"], fallen_contact.astype(np.int32)), (q["terminate"], fallen_contact)
    assert getting_up.sum() >= 4 and fallen_contact.sum() >= 1 and ((par[:, 15] != 0) & ~fallen_contact).sum() >= 1     # a contact while getting up is no fall
    return float(dr), float(dg)


def test_emulated_device_getup_vs_ref_golden(emu_lib):
    print(_getup_device_check(emu_lib, 64, 1e-6, 1e-6))


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [(64, 2e-6), (32, 5e-5)])
def test_hip_getup_vs_ref_golden(hip_lib, prec, tol):
    print(_getup_device_check(hip_lib, prec, tol, tol))


def test_emulated_device_strike_dribble_vs_ref_golden(emu_lib):
    print(_strike_dribble_device_check(emu_lib, 64, 1e-6, 1e-6, 2e-6))        # rewards / goals / states cross the boundary as float32


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [(64, 2e-6), (32, 5e-5)])
def test_hip_strike_dribble_vs_ref_golden(hip_lib, prec, tol):
    print(_strike_dribble_device_check(hip_lib, prec, tol, tol, tol))


def test_emulated_device_task_scenes_vs_ref_golden(emu_lib):
    print(_task_device_check(emu_lib, 64, 1e-6, 1e-6))        # rewards / goals cross the boundary as float32


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [(64, 1e-6), (32, 2e-5)])
def test_hip_task_scenes_vs_ref_golden(hip_lib, prec, tol):
    print(_task_device_check(hip_lib, prec, tol, tol))


def test_oracle_math_vs_ref_golden(oracle_built):
    import ref_lib
    orc = ref_lib.Components("orc")
    g = pc.ref_golden()
    for op, inp, out in zip(g["math/op"], g["math/in"], g["math/out"]):
        got = orc.math_op(int(op), inp)
        tol = 1e-10 if op in (3, 4) else 1e-12
        assert np.abs(got - out[:len(got)]).max() < tol * max(1.0, np.abs(out).max()), op


@pytest.mark.parametrize("name", SCENES)
def test_emulated_device_vs_ref_golden_fp64(emu_lib, name):
    w = pc.check_device_vs_ref_golden(name, 64, emu_lib, rtol_dyn=1e-11, rtol_tau=1e-9, tol_kin=1e-11, tol_state=2e-6,
                                      tol_terms=1e-9, tol_reward=1e-6)   # states / rewards cross the boundary as float32
    print(name, w)


def test_emulated_device_vs_ref_golden_fp32(emu_lib):
    w = pc.check_device_vs_ref_golden("humanoid3d_walk", 32, emu_lib, rtol_dyn=2e-5, rtol_tau=2e-3, tol_kin=5e-6, tol_state=2e-5,
                                      tol_terms=2e-4, tol_reward=1e-4)
    print(w)


# ---- the HIP kernels (C-ABI -> libdm_hip.so) vs the reference vectors ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_hip_vs_ref_golden_fp64(hip_lib, name):
    w = pc.check_device_vs_ref_golden(name, 64, hip_lib, rtol_dyn=1e-11, rtol_tau=1e-9, tol_kin=1e-11, tol_state=2e-6,
                                      tol_terms=1e-9, tol_reward=1e-6)
    print(name, w)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_hip_vs_ref_golden_fp32(hip_lib, name):
    """The production (fp32) kernels against reference fp64 vectors: fixed bounds, no oracle-relative escape hatch.
    Reward within 1e-4 (BASELINE.json north_star), state vector within 2e-5 relative, SPD torque within 2e-3 relative."""
    w = pc.check_device_vs_ref_golden(name, 32, hip_lib, rtol_dyn=2e-5, rtol_tau=2e-3, tol_kin=5e-6, tol_state=2e-5,
                                      tol_terms=2e-4, tol_reward=1e-4)
    print(name, w)
