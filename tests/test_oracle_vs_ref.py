"""The oracle restatement (oracle/orc_*.h) held against the reference's OWN compiled sources (oracle/_ref/libdm_ref.so =
unmodified util/MathUtil.cpp, sim/SpAlg.cpp, anim/KinTree.cpp, anim/Motion.cpp, anim/KinCharacter.cpp,
anim/MotionController.cpp, sim/RBDUtil.cpp, sim/RBDModel.cpp, sim/CtCtrlUtil.cpp ... built by oracle/build_ref.sh against
the Eigen-API shim).  This is what pins SURVEY.md 8(a) rows a5-a7, a10-a12 and a18 to the reference itself.

Tolerances: both sides are IEEE double; they differ in operation order only (the reference works in 6-D spatial algebra on
Eigen types, the oracle on its own V3/M3/Q4).  1e-12 relative to the magnitude of the compared quantity.

Runs on CPU (`-m "not gpu"`).  Needs oracle/_ref/libdm_ref.so (prebuilt, or buildable because /root/reference is present);
skipped with a reason otherwise."""
import os

import numpy as np
import pytest

import ref_lib
from deepmimic_amd import model
from oracle_lib import Oracle
from ref_lib import Components, RefKinChar, Skel, random_pose_vel

pytestmark = pytest.mark.skipif(not ref_lib.ref_available(), reason="oracle/_ref not built and /root/reference absent")

TOL = 1e-12
N_STATES = 1000
CHARS = ["humanoid3d_walk", "dog3d_pace"]


@pytest.fixture(scope="module")
def libs():
    return Components("ref"), Components("orc")


def _close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.abs(a).max()) if a.size else 1.0)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    assert err <= tol * scale, "%s: |ref - oracle| = %.3e (scale %.3g)" % (what, err, scale)
    return err


def _rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


# ---------------------------------------------------------------------------------------------------- cMathUtil
def test_math_ops_random(libs):
    ref, orc = libs
    rng = np.random.default_rng(0)
    for _ in range(2000):
        q0, q1 = _rand_quat(rng), _rand_quat(rng)
        e = rng.normal(size=3) * rng.choice([1e-8, 0.1, 1.0, 3.0, 7.0])
        dt = rng.uniform(1e-3, 0.1)
        t = rng.uniform(0, 1)
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        th = rng.uniform(-7, 7)
        cases = [
            (0, e), (1, q0), (2, np.r_[q0, q1]), (3, np.r_[q0, q1, dt]), (4, np.r_[q0, q1, dt]), (5, q0),
            (6, [rng.uniform(-20, 20)]), (7, np.r_[q0, q1, t]), (9, q0), (10, q0), (11, np.r_[axis, th]),
            (12, np.r_[q0, rng.normal(size=3)]), (13, q0), (14, rng.uniform(-3, 3, size=3)), (16, q0), (17, np.r_[q0, q1]),
            (18, np.r_[axis, th]), (20, q0), (21, [rng.uniform(-5, 5), rng.uniform(0.2, 3), rng.uniform(0, 1), rng.integers(0, 2)]),
        ]
        for op, inp in cases:
            a, b = ref.math_op(op, inp), orc.math_op(op, inp)
            tol = TOL
            if op in (3, 4):
                tol = 1e-10   # angle / dt with dt down to 1e-3; acos near w = 1 amplifies the last bit
            _close(a, b, tol, "math op %d" % op)


def test_math_quaternion_sign_conventions(libs):
    """EulerToQuaternion / RotMatToQuaternion may differ by the overall sign of q between two correct implementations; the
    oracle must reproduce the reference's sign because StandardizeQuat is NOT applied on every path."""
    ref, orc = libs
    rng = np.random.default_rng(1)
    for _ in range(500):
        eul = rng.uniform(-3.1, 3.1, size=3)
        a, b = ref.math_op(15, eul), orc.math_op(15, eul)
        # same rotation always; same sign whenever w is not ~0
        if abs(a[0]) > 1e-9:
            _close(a, b, 1e-11, "EulerToQuaternion")
        else:
            assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-9
        q = _rand_quat(rng)
        a, b = ref.math_op(19, q), orc.math_op(19, q)
        assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-12


def test_slerp_edge_cases(libs):
    ref, orc = libs
    q = np.array([0.5, 0.5, -0.5, 0.5])
    for q1 in (q, -q, q + np.array([1e-17, 0, 0, 0]), np.array([0.5, -0.5, 0.5, 0.5])):
        for t in (0.0, 0.25, 1.0):
            _close(ref.math_op(7, np.r_[q, q1, t]), orc.math_op(7, np.r_[q, q1, t]), TOL, "slerp edge")


def test_check_next_interval_grid(libs):
    """cMathUtil::CheckNextInterval over the 600 Hz update grid: the 30 Hz latch must fire on identical updates."""
    ref, orc = libs
    dt, period = 1.0 / 600, 1.0 / 30
    for t0 in (0.0, 0.0123, 0.37, 1.9):
        t = t0
        fire_r, fire_o = [], []
        for k in range(600):
            t += dt
            fire_r.append(ref.math_op(8, [dt, t, period])[0])
            fire_o.append(orc.math_op(8, [dt, t, period])[0])
        assert fire_r == fire_o
        assert 29 <= sum(fire_r) <= 31


# ---------------------------------------------------------------------------------------------------- cKinTree / cRBDUtil
@pytest.fixture(scope="module", params=CHARS)
def skels(request, libs):
    ref, orc = libs
    t = model.load_asset(request.param)
    return t, Skel(ref, t), Skel(orc, t)


def test_loaded_tables_match_reference_loader(libs):
    """cKinTree::Load / LoadBodyDefs on the reference's character files == the model compiler's tables in assets/."""
    ref, _ = libs
    if not os.path.isdir(ref_lib.REF_DATA):
        pytest.skip("reference data files absent")
    for name in CHARS:
        t = model.load_asset(name)
        jm, bd = ref_lib.ref_load_char(ref, os.path.join("/root/reference", t.cfg.character_file))
        assert jm.shape == t.joint_mat.shape and bd.shape == t.body_defs.shape
        assert np.array_equal(jm, t.joint_mat), name
        assert np.array_equal(bd, t.body_defs), name


def test_dims_and_mass(skels):
    t, sr, so = skels
    assert sr.P == so.P == t.pose_dim
    _close(sr.total_mass(), so.total_mass(), 1e-14, "total mass")
    for j in range(t.num_joints):
        if int(t.body_defs[j, model.BD_SHAPE]) != 0:
            _close(sr.inertia(j), so.inertia(j), TOL, "moment of inertia link %d" % j)


def test_pose_functions_1000_states(skels):
    t, sr, so = skels
    rng = np.random.default_rng(10)
    worst = {}
    for i in range(N_STATES):
        p0, v0 = random_pose_vel(t, rng)
        p1, v1 = random_pose_vel(t, rng)
        lerp = rng.uniform(0, 1)
        dt = rng.uniform(1 / 600, 1 / 30)
        checks = {
            "LerpPoses": (sr.lerp_poses(p0, p1, lerp), so.lerp_poses(p0, p1, lerp), TOL),
            "CalcVel": (sr.calc_vel(p0, p1, dt), so.calc_vel(p0, p1, dt), 1e-10),
            "VelToPoseDiff": (sr.vel_to_pose_diff(p0, v0), so.vel_to_pose_diff(p0, v0), TOL),
            "PostProcessPose": (sr.post_process_pose(p0 * 1.3), so.post_process_pose(p0 * 1.3), TOL),
            "CalcPoseErr/VelErr/RootRotErr": (sr.pose_errs(p0, p1, v0, v1), so.pose_errs(p0, p1, v0, v1), TOL),
            "BuildOriginTrans": (sr.origin_trans(p0), so.origin_trans(p0), TOL),
        }
        for k, (a, b, tol) in checks.items():
            worst[k] = max(worst.get(k, 0.0), _close(a, b, tol, "%s state %d" % (k, i)))
    print("worst |ref - oracle|:", worst)


def test_forward_kinematics_1000_states(skels):
    t, sr, so = skels
    rng = np.random.default_rng(11)
    for i in range(N_STATES):
        p, v = random_pose_vel(t, rng)
        (jr, br), (jo, bo) = sr.world_trans(p), so.world_trans(p)
        _close(jr, jo, TOL, "JointWorldTrans state %d" % i)
        _close(br, bo, TOL, "BodyWorldTrans state %d" % i)
        _close(sr.link_vel(p, v), so.link_vel(p, v), TOL, "link velocities state %d" % i)


def test_mass_matrix_bias_force_1000_states(skels):
    """cRBDModel::Update: BuildMassMat (CRBA, RBDUtil.cpp:123-195) and BuildBiasForce (RNEA, :4-97), incl. the reference's
    BuildCjRoot (which the oracle reproduces on purpose)."""
    t, sr, so = skels
    rng = np.random.default_rng(12)
    wh = wc = 0.0
    for i in range(N_STATES):
        p, v = random_pose_vel(t, rng, big=(i % 4 == 0))
        (Hr, Cr), (Ho, Co) = sr.mass_bias(p, v), so.mass_bias(p, v)
        assert np.array_equal(Hr, Hr.T) or np.abs(Hr - Hr.T).max() < 1e-12
        wh = max(wh, _close(Hr, Ho, TOL, "mass matrix state %d" % i))
        wc = max(wc, _close(Cr, Co, TOL, "bias force state %d" % i))
    print("worst: H %.2e C %.2e" % (wh, wc))


def test_inverse_dynamics_and_com(skels):
    t, sr, so = skels
    rng = np.random.default_rng(13)
    for i in range(200):
        p, v = random_pose_vel(t, rng)
        acc = rng.normal(size=t.pose_dim) * 5
        _close(sr.inv_dyna(p, v, acc), so.inv_dyna(p, v, acc), TOL, "SolveInvDyna state %d" % i)
        (cr, cvr), (co, cvo) = sr.com(p, v), so.com(p, v)
        _close(cr, co, TOL, "CalcCoM pos"); _close(cvr, cvo, TOL, "CalcCoM vel")


def _gains(t):
    """pose-layout Kp/Kd vectors as cImpPDController::Init builds them (zeros on the root)."""
    P = t.pose_dim
    kp, kd = np.zeros(P), np.zeros(P)
    for j in range(1, t.num_joints):
        off = int(t.joint_mat[j, model.JD_PARAM_OFFSET])
        sz = model.joint_param_size(int(t.joint_mat[j, model.JD_TYPE]), False)
        kp[off:off + sz] = t.pd_params[j, 0]
        kd[off:off + sz] = t.pd_params[j, 1]
    return kp, kd


def test_spd_torque_1000_states(skels):
    """Stable-PD torque: the oracle's arithmetic vs the reference ingredients composed as ImpPDController.cpp:136-188."""
    t, sr, so = skels
    kp, kd = _gains(t)
    rng = np.random.default_rng(14)
    worst = 0.0
    for i in range(N_STATES):
        p, v = random_pose_vel(t, rng)
        tar, _ = random_pose_vel(t, rng)
        tar[:7] = 0
        a, b = sr.spd_tau(p, v, tar, kp, kd, 1 / 600), so.spd_tau(p, v, tar, kp, kd, 1 / 600)
        # torques reach 1e3..1e4; the 43x43 pivoted LDLT vs the oracle's reduced LDLT differ in elimination order
        worst = max(worst, _close(a, b, 1e-11, "SPD torque state %d" % i))
    print("worst SPD |ref - oracle| / scale: %.2e" % worst)


def test_reward_terms_vs_scene_oracle(libs):
    """cSceneImitate::CalcRewardImitate composed from reference functions (ref_glue.cpp) vs the oracle Scene's calc_reward on
    the same (sim pose/vel, kin pose/vel, kin origin)."""
    ref, orc = libs
    rng = np.random.default_rng(15)
    for name in CHARS + ["humanoid3d_spinkick"]:
        t = model.load_asset(name)
        o = Oracle(t)
        sr = Skel(ref, t)
        w = t.joint_mat[:, model.JD_DIFF_W].copy()
        w = w / np.abs(w).sum()    # cSceneImitate::CalcJointWeights (SceneImitate.cpp:236-248)
        for i in range(100):
            tt = rng.uniform(0, 2.5 * o.duration)
            o.reset(tt)
            p, v = o.sim_state()
            v = v + rng.normal(size=v.shape) * (v != 0) * 0.5
            for j in range(1, t.num_joints):
                off, ty = int(t.joint_mat[j, model.JD_PARAM_OFFSET]), int(t.joint_mat[j, model.JD_TYPE])
                if ty == model.JT_SPHERICAL:
                    q = p[off:off + 4] + 0.15 * rng.normal(size=4); q /= np.linalg.norm(q)
                    p[off:off + 4] = q if q[0] >= 0 else -q
                elif ty == model.JT_REVOLUTE:
                    p[off] += 0.1 * rng.normal()
            p[0:3] += rng.normal(size=3) * 0.05
            p[1] += 0.3   # keep every link off the ground: a fallen character has reward 0 by definition
            o.set_sim_state(p, v)
            ps, vs = o.sim_state()
            kp_, kv_, ko = o.kin_state()
            r_o, terms_o = o.calc_reward_terms()
            out = ref_lib.ref_reward_terms(ref, sr, ps, vs, kp_, kv_, w, 0.0, ko[1])
            _close(out[:5], terms_o, 1e-11, "%s reward terms %d" % (name, i))
            _close(out[5], r_o, 1e-12, "%s reward %d" % (name, i))


def test_action_meta_vs_product_tables(libs, emu_lib):
    """cCtCtrlUtil::BuildBoundsPD / BuildOffsetScalePD (compiled reference) vs what the PRODUCT's host code reports through
    dm_build_offsets_scales (facade BuildAction{Offset,Scale,BoundMin,BoundMax}); the host code runs here in the CPU
    emulator build of the same sources."""
    from deepmimic_amd.core import BatchEnv
    ref, _ = libs
    for name in CHARS:
        t = model.load_asset(name)
        sr = Skel(ref, t)
        n, lo, hi, off, sc = ref_lib.ref_action_meta(ref, sr, t.action_dim)
        assert n == t.action_dim
        m = BatchEnv(t, 1, precision=64, lib_path=emu_lib).offsets_scales()
        _close(lo, m["action_min"], 1e-14, "action bound min"); _close(hi, m["action_max"], 1e-14, "action bound max")
        _close(off, m["action_offset"], 1e-14, "action offset"); _close(sc, m["action_scale"], 1e-14, "action scale")


# ---------------------------------------------------------------------------------------------------- cMotion / cKinCharacter
@pytest.fixture(scope="module", params=["humanoid3d_walk", "humanoid3d_spinkick", "dog3d_pace", "humanoid3d_backflip"])
def kin_pair(request, libs):
    ref, orc = libs
    if not os.path.isdir(ref_lib.REF_DATA):
        pytest.skip("reference data files absent (the kinematic character is loaded from them)")
    t = model.load_asset(request.param)
    kc = RefKinChar(ref, os.path.join("/root/reference", t.cfg.character_file), os.path.join("/root/reference", t.cfg.motion_file))
    o = Oracle(t)
    return t, kc, o


def test_motion_frames_and_frame_velocities(kin_pair):
    """cMotion::LoadJson + PostProcessFrames + BuildFrameVel (Motion.cpp:170-191,302-430) vs the oracle's Motion::load."""
    t, kc, o = kin_pair
    assert kc.F == o.F and kc.P == o.P and kc.loop == bool(t.loop)
    _close(kc.duration, o.duration, 1e-14, "duration")
    for f in range(kc.F):
        fr, fv, ft = kc.frame(f)
        fo, vo, to = o.motion_frame(f)
        _close(fr, fo, 1e-14, "frame %d" % f)
        _close(fv, vo, 1e-10, "frame vel %d" % f)
        _close(ft, to, 1e-14, "frame time %d" % f)


def test_kin_character_pose_vel_over_time(kin_pair, libs):
    """cKinCharacter::CalcPose / CalcVel (origin transform, cycle root offset, slerp blending) at 400 times spanning
    several cycles, for three origins."""
    t, kc, o = kin_pair
    ref, orc = libs
    import ctypes as C
    rng = np.random.default_rng(20)
    origins = [(np.zeros(3), np.array([1.0, 0, 0, 0])),
               (np.array([0.7, 0.013, -1.9]), np.array([np.cos(0.4), 0, np.sin(0.4), 0])),
               (np.array([-3.0, 0.0, 2.5]), np.array([np.cos(-1.3), 0, np.sin(-1.3), 0]))]
    hi = 3.2 * o.duration if t.loop else 1.2 * o.duration
    for pos, rot in origins:
        kc.set_origin(pos, rot)
        o.reset(0.0)
        orc.lib.orc_set_kin_origin(o.h, ref_lib._d(ref_lib._arr(pos)), ref_lib._d(ref_lib._arr(rot)))
        for tt in np.r_[0.0, o.duration, rng.uniform(-0.2 * o.duration if t.loop else 0.0, hi, size=400)]:
            pr, vr = kc.eval(tt)
            po, vo = o.kin_eval(tt)
            _close(pr, po, TOL, "CalcPose t=%.4f" % tt)
            _close(vr, vo, 1e-11, "CalcVel t=%.4f" % tt)
    kc.set_origin(np.zeros(3), np.array([1.0, 0, 0, 0]))


def test_raw_motion_eval(kin_pair, libs):
    """cMotion::CalcFrame / CalcFrameVel without origin or cycle offset (what RecordAMPObsExpert samples)."""
    t, kc, o = kin_pair
    _, orc = libs
    rng = np.random.default_rng(21)
    for tt in rng.uniform(0, 2 * o.duration if t.loop else o.duration, size=200):
        fr, vr = kc.motion_eval(tt)
        fo, vo = np.zeros(o.P), np.zeros(o.P)
        orc.lib.orc_motion_eval(o.h, C_double(tt), ref_lib._d(fo), ref_lib._d(vo))
        _close(fr, fo, TOL, "CalcFrame t=%.4f" % tt)
        _close(vr, vo, 1e-11, "CalcFrameVel t=%.4f" % tt)


def C_double(x):
    import ctypes
    return ctypes.c_double(x)


def test_kin_character_stateful_update_and_sync(kin_pair, libs):
    """The stateful path of cSceneImitate::UpdateKinChar: SetTime, 600 Hz Update over two cycles, with the root moved /
    rotated mid-way the way SyncKinCharNewCycle does (SetRootPos / RotateRoot -> origin changes)."""
    t, kc, o = kin_pair
    _, orc = libs
    t0 = 0.31 * o.duration
    kc.set_origin(np.zeros(3), np.array([1.0, 0, 0, 0]))
    kc.set_time(t0)
    o.reset(0.0)
    orc.lib.orc_set_kin_origin(o.h, ref_lib._d(np.zeros(3)), ref_lib._d(np.array([1.0, 0, 0, 0])))
    orc.lib.orc_kin_set_time(o.h, C_double(t0))
    n = int(2.2 * o.duration * 600) if t.loop else int(0.6 * o.duration * 600)
    for k in range(n):
        kc.update(1 / 600)
        orc.lib.orc_kin_set_time(o.h, C_double(kc.time()))   # same clock value on both sides; Update itself is time += dt; Pose()
        if k == n // 3:
            pr, _ = kc.state()
            newp = pr[:3] + np.array([0.4, 0.02, -0.3])
            kc.set_root_pos(newp)
            orc.lib.orc_kin_set_root_pos(o.h, ref_lib._d(ref_lib._arr(newp)))
        if k == n // 2:
            dq = np.array([np.cos(0.35), 0, np.sin(0.35), 0])
            kc.rotate_root(dq)
            orc.lib.orc_kin_rotate_root(o.h, ref_lib._d(dq))
        if k % 7 == 0 or k in (n // 3, n // 2):
            pr, vr = kc.state()
            po, vo, oo = o.kin_state()
            _close(pr, po, 1e-11, "kin pose update %d" % k)
            _close(vr, vo, 1e-10, "kin vel update %d" % k)
            op, orot = kc.get_origin()
            _close(np.r_[op, orot], oo, 1e-11, "kin origin update %d" % k)
    if t.loop:
        assert kc.cycle() == orc.lib.orc_kin_cycle(o.h, C_double(kc.time()))
    assert kc.phase() == pytest.approx(o_phase(o, kc.time()), abs=1e-12)


def o_phase(o, t):
    ph = t / o.duration
    return ph - np.floor(ph) if o.t.loop else min(max(ph, 0.0), 1.0)


def test_timer_end(libs):
    """cTimer::Update / IsEnd (util/Timer.cpp:55-83) vs the oracle scene's episode clock."""
    ref, _ = libs
    ref.lib.ref_timer_first_end.argtypes = None
    t = model.load_asset("humanoid3d_walk")
    for max_time in (0.05, 0.5, 1.0 / 3):
        o = Oracle(t)
        o.reset(0.0, max_time)
        first = -1
        for k in range(400):
            o.update(1 / 600)
            if o.time() >= max_time and first < 0:
                first = k
        import ctypes
        got = ref.lib.ref_timer_first_end(ctypes.c_double(max_time), ctypes.c_double(1 / 600), 400)
        assert got == first, (max_time, got, first)


def test_timer_draws_and_annealing_vs_reference(libs):
    """cTimer::Reset for both timer types (util/Timer.cpp:55-73) and the annealed parameters (cTimer::tParams::Blend at cAnnealer's pow-4 lerp,
    scenes/RLSceneSimChar.cpp:330-347), compiled from the reference's own sources: model.draw_time_limit maps uniform numbers to the same
    distribution (the reference draws from std::default_random_engine, this path from its counter generator: distributions, not sequences, can
    agree), model.timer_limits / timer_exp give the same blended parameters"""
    import ctypes
    ref, _ = libs
    n = 40000
    u = (np.arange(n) + 0.5) / n
    for typ, name, tmin, tmax, texp in ((0, "uniform", 0.5, 3.0, 1.0), (1, "exp", 0.5, 3.0, 0.8), (1, "exp", 2.0, 50.0, 5.0)):
        out = np.zeros(n)
        ref.lib.ref_timer_draws(typ, ctypes.c_double(tmin), ctypes.c_double(tmax), ctypes.c_double(texp), ctypes.c_ulong(7), n,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        mine = np.sort([model.draw_time_limit(name, tmin, tmax, texp, x) for x in u])
        assert out.min() >= tmin and out.max() <= tmax
        # two-sample Kolmogorov distance between the reference's draws and the quantiles of this path's map
        grid = np.linspace(tmin, tmax, 400)
        ks = np.abs(np.searchsorted(np.sort(out), grid, side="right") / n - np.searchsorted(mine, grid, side="right") / n).max()
        assert ks < 0.012, (name, ks)
        assert abs((out == tmax).mean() - (mine == tmax).mean()) < 0.005                     # the mass the exp timer clips at time_lim_max
    cfg = model.SceneConfig()
    cfg.timer_type = "exp"; cfg.time_lim_min, cfg.time_lim_max, cfg.time_lim_exp = 0.5, 0.5, 0.2
    cfg.time_end_lim_min, cfg.time_end_lim_max, cfg.time_end_lim_exp = 20.0, 40.0, 5.0; cfg.anneal_samples = 1000
    for count in (0, 100, 500, 900, 1000, 5000):
        out = np.zeros(4)
        p0 = np.array([0.5, 0.5, 0.2]); p1 = np.array([20.0, 40.0, 5.0])
        ref.lib.ref_timer_anneal(p0.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), p1.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                 ctypes.c_double(count / 1000.0), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        lo, hi = model.timer_limits(cfg, False, count)
        assert abs(lo - out[1]) < 1e-12 and abs(hi - out[2]) < 1e-12 and abs(model.timer_exp(cfg, False, count) - out[3]) < 1e-12, (count, out)


def test_time_warp_alignment_vs_reference_dtw(libs):
    """model.time_warp_cost (the facade's test-mode score of imitate_amp) vs the reference's cDynamicTimeWarper, compiled"""
    import ctypes
    ref, _ = libs
    ref.lib.ref_time_warp.restype = ctypes.c_double
    rng = np.random.default_rng(30)
    for n, m in ((5, 5), (17, 17), (9, 12), (2, 2)):
        d0, d1 = rng.normal(size=(n, 45)), rng.normal(size=(m, 45)) * 0.7 + 0.1
        want = ref.lib.ref_time_warp(ref_lib._d(ref_lib._arr(d0)), n, ref_lib._d(ref_lib._arr(d1)), m, 45, 40)
        assert abs(model.time_warp_cost(d0, d1) - want) < 1e-12 * max(1.0, abs(want))


def test_amp_observation_vs_reference_composition(libs):
    """RecordAMPObsAgent of the oracle Scene (what the device AMP path is compared with) vs cSceneImitateAMP::BuildAMPObs composed
    from the compiled reference functions (ref_glue.cpp ref_amp_obs): humanoid and dog, with and without
    --enable_amp_obs_local_root, states in the middle of a rollout (history = state at the last action latch)."""
    import copy, ctypes
    ref, _ = libs
    rng = np.random.default_rng(40)
    for name in CHARS:
        for local_root in (False, True):
            t = copy.deepcopy(model.load_asset(name))
            t.cfg.scene = "imitate_amp"; t.cfg.enable_amp_obs_local_root = local_root
            o = Oracle(t)
            sr = Skel(ref, t)
            for trial in range(4):
                o.reset(float(rng.uniform(0, o.duration)))
                for k in range(int(rng.integers(1, 4))):
                    o.set_action(0.2 * rng.normal(size=o.A))
                    for u in range(int(rng.integers(3, 20))):
                        o.update(1.0 / 600)
                pp, pv = o.prev_state(); p, v = o.sim_state()
                got = o.amp_obs_agent()
                want = np.zeros(len(got) + 8)
                n = ref.lib.ref_amp_obs(sr.h, ref_lib._d(ref_lib._arr(pp)), ref_lib._d(ref_lib._arr(pv)), ref_lib._d(ref_lib._arr(p)),
                                        ref_lib._d(ref_lib._arr(v)), ctypes.c_double(0.0), int(local_root), ref_lib._d(want))
                assert n == len(got), (n, len(got))
                _close(want[:n], got, 1e-12, "%s AMP obs local_root=%s" % (name, local_root))
