"""Shared parity checks: device path (real GPU library or the CPU emulator build of the same sources) vs the oracle."""
import numpy as np

from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
from oracle_lib import Oracle

DT = 1.0 / 600


def dof_index(t):
    """pose-layout index of every generalized velocity (root lin, root ang, joints)."""
    idx = [0, 1, 2, 3, 4, 5]
    for j in range(1, t.num_joints):
        off, ty = int(t.joint_mat[j, model.JD_PARAM_OFFSET]), int(t.joint_mat[j, model.JD_TYPE])
        if ty == model.JT_SPHERICAL:
            idx += [off, off + 1, off + 2]
        elif ty == model.JT_REVOLUTE:
            idx += [off]
    return np.array(idx)


def random_states(t, o, n, seed, vel_scale=1.0, lift=0.0):
    """n plausible sim states: reference pose at a random phase + perturbed velocities."""
    rng = np.random.default_rng(seed)
    poses, vels, times = [], [], []
    for _ in range(n):
        tt = rng.uniform(0, o.duration)
        o.reset(tt)
        p, v = o.sim_state()
        p[1] += lift
        v = v + vel_scale * rng.normal(size=v.shape) * (v != 0)
        # perturb joint rotations a little
        for j in range(1, t.num_joints):
            off, ty = int(t.joint_mat[j, model.JD_PARAM_OFFSET]), int(t.joint_mat[j, model.JD_TYPE])
            if ty == model.JT_SPHERICAL:
                q = p[off:off + 4] + 0.1 * rng.normal(size=4)
                q /= np.linalg.norm(q)
                p[off:off + 4] = q if q[0] >= 0 else -q
                v[off:off + 3] += vel_scale * rng.normal(size=3)
            elif ty == model.JT_REVOLUTE:
                v[off] += vel_scale * rng.normal()
        v[:6] += vel_scale * 0.3 * rng.normal(size=6)
        poses.append(p); vels.append(v); times.append(tt)
    return np.array(poses), np.array(vels), np.array(times)


def make_pair(name, n, precision, lib_path, **kw):
    t = model.load_asset(name)
    o = Oracle(t)
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, **kw)
    return t, o, env


def check_reset_and_query(name, precision, lib_path, tol_state, tol_reward):
    t, o, env = make_pair(name, 4, precision, lib_path)
    times = np.array([0.0, 0.21, 0.8 * o.duration, 2.3 * o.duration])
    env.reset(kin_times=times, max_times=np.inf)
    st = env.get_state(); q = env.query()
    for e, tt in enumerate(times):
        o.reset(tt)
        p, v = o.sim_state(); kp, kv, ko = o.kin_state()
        assert np.abs(st["pose"][e] - p).max() < tol_state
        assert np.abs(st["vel"][e] - v).max() < 50 * tol_state * max(1.0, np.abs(v).max()), np.abs(st["vel"][e] - v).max()
        assert np.abs(st["kin"][e] - ko).max() < tol_state
        assert np.abs(q["state"][e] - o.record_state()).max() < max(50 * tol_state, 2e-6)
        assert abs(q["reward"][e] - o.calc_reward()) < tol_reward
        assert q["terminate"][e] == o.check_terminate() and q["need_new_action"][e] == 1
        assert st["clocks"][e][0] == tt and st["clocks"][e][2] == -tt


def check_dynamics(name, precision, lib_path, rtol):
    t, o, env = make_pair(name, 8, precision, lib_path)
    idx = dof_index(t)
    P, V, _ = random_states(t, o, 8, seed=1)
    env.set_state(pose=P, vel=V)
    env.probe(2, DT)
    H, C = env.debug("H"), env.debug("C")
    for e in range(8):
        Ho, Co = o.mass_bias(0, P[e], V[e])
        Ho = Ho[np.ix_(idx, idx)]; Co = Co[idx]
        assert np.abs(H[e] - Ho).max() < rtol * np.abs(Ho).max()
        assert np.abs(C[e] - Co).max() < rtol * max(1.0, np.abs(Co).max())


def check_spd(name, precision, lib_path, rtol):
    t, o, env = make_pair(name, 8, precision, lib_path)
    idx = dof_index(t)
    P, V, T = random_states(t, o, 8, seed=2, vel_scale=0.5)
    tars = []
    for e in range(8):
        kp, _ = o.kin_eval(T[e] + 0.05)
        o.set_action(o.pose_to_action(kp)); tars.append(o.tar_pose())
    env.set_state(pose=P, vel=V, tar=np.array(tars))
    env.probe(0, DT)
    tau = env.debug("tau")
    for e in range(8):
        kp, _ = o.kin_eval(T[e] + 0.05)
        o.set_action(o.pose_to_action(kp))
        o.set_sim_state(P[e], V[e])
        tau_o = o.spd_tau(DT)[idx]
        assert np.abs(tau[e] - tau_o).max() < rtol * max(1.0, np.abs(tau_o).max()), (e, np.abs(tau[e] - tau_o).max())


def check_substep(name, precision, lib_path, tol_vel, tol_pose, lift=0.0, n=8):
    t, o, env = make_pair(name, n, precision, lib_path)
    idx = dof_index(t)
    P, V, T = random_states(t, o, n, seed=3, vel_scale=0.3, lift=lift)
    rng = np.random.default_rng(7)
    tau = 20.0 * rng.normal(size=(n, len(idx))); tau[:, :6] = 0
    env.set_state(pose=P, vel=V)
    # latch torques through the tau tap: run SPD once is not needed, write via state (tau lives in EnvState)
    env.set_tau(tau)
    env.probe(1, DT / 2)
    st = env.get_state(); rows = env.debug("rows")
    tot_contacts = 0
    check_substep.self_contacts = 0
    for e in range(n):
        o.set_sim_state(P[e], V[e])
        tp = np.zeros(o.P); tp[idx] = tau[e]; o.set_tau(tp)
        o.substep(DT / 2)
        p2, v2 = o.sim_state()
        assert int(rows[e][0]) == o.num_rows() and int(rows[e][1]) == o.num_contacts()
        tot_contacts += o.num_contacts()
        check_substep.self_contacts += o.num_self_contacts()
        vtol = tol_vel * max(1.0, np.abs(v2).max())          # relative: deep penetration yields large push-out velocities
        assert np.abs(st["vel"][e] - v2).max() < vtol, (e, np.abs(st["vel"][e] - v2).max(), np.abs(v2).max())
        assert np.abs(st["pose"][e] - p2).max() < tol_pose * max(1.0, np.abs(v2).max()), np.abs(st["pose"][e] - p2).max()
        assert st["flags"][e][1] == int(sum(int(c) << j for j, c in enumerate(o.contacts())))
    return tot_contacts


def fp32_step_sensitivity(name, steps, t0=0.0):
    """Per-control-step |reward(fp64 oracle) - reward(fp32 oracle)| with the fp32 oracle teacher-forced from the fp64 one.

    The fp32 oracle is the same restatement with every real narrowed to float (oracle/Makefile).  A step on which it
    already disagrees with the fp64 oracle is ill-conditioned in single precision (a contact candidate sits on its
    activation threshold and rounding decides the manifold): no fp32 implementation can be held to 1e-4 there."""
    t = model.load_asset(name)
    o, f = Oracle(t), Oracle(t, variant="f32")
    o.reset(t0); f.reset(t0)
    dr = []
    for k in range(steps):
        kp, _, _ = o.kin_state()
        a = o.pose_to_action(kp)
        o.set_action(a); f.set_action(a)
        p, v = o.sim_state(); f.set_sim_state(p, v)
        for u in range(20):
            o.update(DT); f.update(DT)
        dr.append(abs(o.calc_reward() - f.calc_reward()))
    return np.array(dr)


def fp32_free_running_sensitivity(name, steps, t0=0.0):
    """Per-step |reward| gap between the fp64 oracle and its own fp32 build, both free-running on stream A1."""
    t = model.load_asset(name)
    o, f = Oracle(t), Oracle(t, variant="f32")
    o.reset(t0); f.reset(t0)
    dr = []
    for k in range(steps):
        for x in (o, f):
            kp, _, _ = x.kin_state()
            x.set_action(x.pose_to_action(kp))
            for u in range(20):
                x.update(DT)
        dr.append(abs(o.calc_reward() - f.calc_reward()))
    return np.array(dr)


def rollout_compare(name, precision, lib_path, steps, t0=0.0, open_loop_on_device=False, resync=False):
    """Open-loop mocap-tracking rollout (stream A1); returns per-step |reward diff|, max state diff, flags equal.

    resync=True is the teacher-forced variant: before every control step the device env is set to the oracle's
    state, so each of the `steps` comparisons checks one control step (20 updates, 40 substeps) from identical
    inputs -- independent of the chaotic divergence a free-running contact simulation shows on some clips."""
    t, o, env = make_pair(name, 1, precision, lib_path)
    o.reset(t0); env.reset(kin_times=[t0], max_times=np.inf)
    dr, ds, flags_ok = [], [], True
    for k in range(steps):
        kp, kv, ko = o.kin_state()
        o.set_action(o.pose_to_action(kp))
        if resync:
            p, v = o.sim_state()
            cm = int(sum(int(c) << j for j, c in enumerate(o.contacts())))
            env.set_state(pose=p[None], vel=v[None], tar=o.tar_pose()[None], kin=ko[None],
                          clocks=np.array([[o.kin_time(), o.kin_time(), -t0, o.time(), np.inf]]),
                          flags=np.array([[int(o.need_new_action()), cm, 1, 1]], dtype=np.int32))
        if open_loop_on_device:
            out = env.step(None, DT, 20, open_loop=True)
        else:
            if not resync:
                env.set_state(tar=o.tar_pose()[None])
            out = env.step(None, DT, 20)
        for u in range(20):
            o.update(DT)
        dr.append(abs(float(out["reward"][0]) - o.calc_reward()))
        ds.append(np.abs(out["state"][0] - o.record_state()).max())
        flags_ok &= (int(out["terminate"][0]) == o.check_terminate()) and (int(out["valid"][0]) == int(o.check_valid_episode()))
    return np.array(dr), np.array(ds), flags_ok


def batch_rollout_compare(name, precision, lib_path, steps, t0s, wave_packing=0, lifts=None, stats=None):
    """Free-running open-loop rollout of len(t0s) envs in one batch (no debug taps armed, so the production step kernel of
    the requested wave packing runs); every env is compared with its own oracle.  Returns per-env max |reward diff|, max
    |state diff| and whether every terminate / valid flag agreed."""
    t = model.load_asset(name)
    n = len(t0s)
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing)
    env.reset(kin_times=np.array(t0s, dtype=np.float64), max_times=np.inf)
    oracles = []
    for e, t0 in enumerate(t0s):
        o = Oracle(t); o.reset(t0); oracles.append(o)
    if lifts is not None:                      # push some characters into the ground: many contacts (> 32 rows)
        st = env.get_state()
        pose = st["pose"].copy()
        for e, o in enumerate(oracles):
            pose[e, 1] += lifts[e]
            p, v = o.sim_state(); p[1] += lifts[e]; o.set_sim_state(p, v)
        env.set_state(pose=pose)
    dr, ds, ok = np.zeros(n), np.zeros(n), True
    for k in range(steps):
        out = env.step(None, DT, 20, open_loop=True)
        for e, o in enumerate(oracles):
            kp, _, _ = o.kin_state()
            o.set_action(o.pose_to_action(kp))
            for u in range(20):
                o.update(DT)
            dr[e] = max(dr[e], abs(float(out["reward"][e]) - o.calc_reward()))
            ds[e] = max(ds[e], np.abs(out["state"][e] - o.record_state()).max())
            ok &= int(out["terminate"][e]) == o.check_terminate() and int(out["valid"][e]) == int(o.check_valid_episode())
    if stats is not None:          # per env: substeps on the 64-lane fallback | on borrowed lanes (two-per-wave kernel)
        stats["fallback"] = env.debug("fallback"); stats["borrowed"] = env.debug("borrowed")
    return dr, ds, ok


def action_rollout_compare(name, precision, lib_path, steps, stream, t0s, wave_packing=0):
    """Explicit-action rollout (the a10 action path: exp-map / angle -> PD target on the device) of len(t0s) envs.

    stream "A0" = zeros, "A1" = encoding of the oracle's kinematic pose, "A2" = A1 + Philox N(0, 0.05^2) noise keyed by
    env id and control step (deepmimic_amd/streams.py).  Actions cross the boundary as float32, so the oracle is fed
    the float32-rounded values.  Returns per-env max |reward diff|, max |state diff|, flags equal, #terminated."""
    from deepmimic_amd import streams
    t = model.load_asset(name)
    n = len(t0s)
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing)
    env.reset(kin_times=np.array(t0s, dtype=np.float64), max_times=np.inf)
    oracles = []
    for t0 in t0s:
        o = Oracle(t); o.reset(t0); oracles.append(o)
    A = env.A
    dr, ds, ok, fallen = np.zeros(n), np.zeros(n), True, 0
    for k in range(steps):
        if stream == "A0":
            acts = streams.stream_a0(n, A)
        else:
            acts = np.array([o.pose_to_action(o.kin_state()[0]) for o in oracles])
            if stream == "A2":
                acts = streams.stream_a2(acts, np.arange(n), k)
        acts = acts.astype(np.float32)
        out = env.step(acts, DT, 20)
        for e, o in enumerate(oracles):
            o.set_action(acts[e].astype(np.float64))
            for u in range(20):
                o.update(DT)
            dr[e] = max(dr[e], abs(float(out["reward"][e]) - o.calc_reward()))
            ds[e] = max(ds[e], np.abs(out["state"][e] - o.record_state()).max())
            ok &= int(out["terminate"][e]) == o.check_terminate() and int(out["valid"][e]) == int(o.check_valid_episode())
    fallen = sum(o.check_terminate() == 1 for o in oracles)
    return dr, ds, ok, fallen


# ---------------------------------------------------------------------------------------------------------------------------
# Device path vs vectors produced by the REFERENCE's own compiled sources (tests/golden/ref_vectors.npz, generator
# tests/golden/make_ref_golden.py, library oracle/_ref/libdm_ref.so).  No oracle in the loop.
_REF_GOLDEN = None


def ref_golden():
    global _REF_GOLDEN
    if _REF_GOLDEN is None:
        import os
        _REF_GOLDEN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))
    return _REF_GOLDEN


def clamp_tau(t, tau_pose):
    """cSimBodyJoint::ClampTotalTorque on the per-joint vector norm (SimBodyJoint.cpp:299-307); root torque is not applied."""
    out = tau_pose.copy()
    out[:7] = 0
    for j in range(1, t.num_joints):
        off, ty = int(t.joint_mat[j, model.JD_PARAM_OFFSET]), int(t.joint_mat[j, model.JD_TYPE])
        lim = t.joint_mat[j, model.JD_TORQUE_LIM]
        if ty == model.JT_SPHERICAL:
            mag = np.linalg.norm(out[off:off + 3])
            if mag > lim:
                out[off:off + 3] *= lim / mag
            out[off + 3] = 0
        elif ty == model.JT_REVOLUTE:
            if abs(out[off]) > lim:
                out[off] *= lim / abs(out[off])
    return out


def check_device_vs_ref_golden(name, precision, lib_path, rtol_dyn, rtol_tau, tol_kin, tol_state, tol_terms, tol_reward):
    """Mass matrix, bias force, SPD torque, kinematic-character sample, link kinematics, state vector, reward terms and reward
    of the device path against reference-generated vectors.  Returns the worst deviations (for the log)."""
    g = ref_golden()
    t = model.load_asset(name)
    G = lambda k: g["%s/%s" % (name, k)]
    n = G("pose").shape[0]
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path)
    idx = dof_index(t)
    P, V, T = G("pose"), G("vel"), G("tar")
    tk = G("kin_time")
    clocks = np.stack([tk, tk, np.zeros(n), np.zeros(n), np.full(n, np.inf)], axis=1)
    flags = np.tile(np.array([[1, 0, 1, 1]], dtype=np.int32), (n, 1))
    env.reset(kin_times=tk, max_times=np.inf)
    env.set_state(pose=P, vel=V, tar=T, kin=G("kin_origin"), clocks=clocks, flags=flags)
    worst = {}
    # a12: H, C of the SPD model
    env.probe(2, DT)
    H, C = env.debug("H"), env.debug("C")
    for e in range(n):
        Hr, Cr = G("H")[e][np.ix_(idx, idx)], G("C")[e][idx]
        dh = np.abs(H[e] - Hr).max() / np.abs(Hr).max(); dc = np.abs(C[e] - Cr).max() / max(1.0, np.abs(Cr).max())
        worst["H"] = max(worst.get("H", 0), dh); worst["C"] = max(worst.get("C", 0), dc)
        assert dh < rtol_dyn and dc < rtol_dyn, (e, dh, dc)
    # a11 + a13: SPD torque after the clamp
    env.set_state(pose=P, vel=V, tar=T)
    env.probe(0, DT)
    tau = env.debug("tau")
    for e in range(n):
        tr = clamp_tau(t, G("spd_tau")[e])[idx]
        d = np.abs(tau[e] - tr).max() / max(1.0, np.abs(tr).max())
        worst["tau"] = max(worst.get("tau", 0), d)
        assert d < rtol_tau, (e, d)
    # a5/a6 kin sample, a17 links, a19 state, a18 reward
    env.set_state(pose=P, vel=V, tar=T, kin=G("kin_origin"), clocks=clocks, flags=flags)
    q = env.query()
    kp, kv, links, terms = env.debug("kin_pose"), env.debug("kin_vel"), env.debug("links"), env.debug("reward_terms")
    for e in range(n):
        dk = max(np.abs(kp[e] - G("kin_pose")[e]).max(), np.abs(kv[e] - G("kin_vel")[e]).max() / max(1.0, np.abs(G("kin_vel")[e]).max()))
        worst["kin"] = max(worst.get("kin", 0), dk)
        assert dk < tol_kin, (e, dk)
        bw, lv, jw = G("body_world")[e], G("link_vel")[e], G("joint_world")[e]
        dl = max(np.abs(links[e][:, 0:3] - bw[:, 9:12]).max(), np.abs(links[e][:, 3:12] - bw[:, 0:9]).max(),
                 np.abs(links[e][:, 18:21] - jw[:, 9:12]).max(),
                 np.abs(links[e][:, 12:18] - lv).max() / max(1.0, np.abs(lv).max()))
        worst["links"] = max(worst.get("links", 0), dl)
        assert dl < tol_kin, (e, dl)
        ds = np.abs(q["state"][e] - G("state")[e]).max() / max(1.0, np.abs(G("state")[e]).max())
        worst["state"] = max(worst.get("state", 0), ds)
        assert ds < tol_state, (e, ds)
        dt_ = np.abs(terms[e] - G("reward_terms")[e]).max() / max(1.0, np.abs(G("reward_terms")[e]).max())
        dr = abs(float(q["reward"][e]) - float(G("reward")[e]))
        worst["terms"] = max(worst.get("terms", 0), dt_); worst["reward"] = max(worst.get("reward", 0), dr)
        assert dt_ < tol_terms and dr < tol_reward, (e, dt_, dr)
        assert 0.0 < G("reward")[e] < 1.0
    return worst


def auto_reset_rollout_compare(name, precision, lib_path, steps, n, seed, wave_packing=0, time_lim=np.inf, physics=1):
    """Free-running open-loop rollout THROUGH auto-resets: the oracle mirrors every reset of the device with the same
    counter-based draw (streams.reset_rand01), so all `steps` control steps of every env are live transitions (a fallen
    character is reset instead of lying on the ground with reward 0).  Returns per (step, env) arrays: |reward diff|,
    max |state diff| (relative to max(1, |state|)), alive mask (the oracle's reward is non-zero), number of resets, flags ok."""
    from deepmimic_amd import streams
    t = model.load_asset(name)
    if np.isfinite(time_lim):
        t.cfg.time_lim_min = t.cfg.time_lim_max = float(time_lim)
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing, seed=seed, physics=physics)
    env.reset()
    ep = env.get_state()["flags"][:, 2].astype(np.int64)        # episode counter the NEXT reset will draw with
    tmin, tmax = float(t.cfg.time_lim_min), float(t.cfg.time_lim_max)

    def draw(e, episode):     # (clip time, episode-timer limit) of a reset, as k_env_step / k_env_reset draw them
        mt = tmin + (tmax - tmin) * streams.reset_rand01(seed, e, episode, 1) if tmax > tmin else tmax
        return streams.reset_rand01(seed, e, episode, 0), mt

    oracles = []
    for e in range(n):
        o = Oracle(t) if physics == 1 else Oracle(t, physics=physics, max_contacts=env.max_contacts)
        u, mt = draw(e, int(ep[e]) - 1)
        o.reset(o.duration * u, mt)
        oracles.append(o)
    st0 = env.get_state()
    for e, o in enumerate(oracles):
        assert abs(st0["clocks"][e][0] - o.kin_time()) < 1e-12, "reset draw mismatch"
    dr, ds, alive = np.zeros((steps, n)), np.zeros((steps, n)), np.zeros((steps, n), dtype=bool)
    resets, ok = 0, True
    for k in range(steps):
        out = env.step(None, DT, 20, open_loop=True, auto_reset=True)
        for e, o in enumerate(oracles):
            kp, _, _ = o.kin_state()
            o.set_action(o.pose_to_action(kp))
            o.control_step(20, DT)            # ends at the update where the episode is over, like the device with auto-reset
            r = o.calc_reward()
            dr[k, e] = abs(float(out["reward"][e]) - r)
            alive[k, e] = r != 0.0
            term, end = o.check_terminate(), o.is_episode_end()
            ok &= int(out["terminate"][e]) == term and bool(out["episode_end"][e]) == end and int(out["valid"][e]) == int(o.check_valid_episode())
            if end or not o.check_valid_episode():       # the device resets after writing reward / flags (an invalid episode is reset too); its observation is the first of the new episode
                u, mt = draw(e, int(ep[e]))
                o.reset(o.duration * u, mt)
                ep[e] += 1; resets += 1
            so = o.record_state()
            ds[k, e] = np.abs(out["state"][e] - so).max() / max(1.0, np.abs(so).max())
    return dr, ds, alive, resets, ok


def stepwise_live_compare(name, precision, lib_path, steps, n, seed, wave_packing=0):
    """Teacher-forced, live steps only: `n` oracles free-run through their own resets (clip time drawn with
    streams.reset_rand01; an episode ends on fall / motion end / timer); before every control step the device envs are set to the
    oracles' states, so each of the steps x n comparisons checks ONE control step (20 updates, 40 substeps) from identical inputs,
    and none of them is the vacuous reward-0 of a character lying on the ground.
    Returns |reward diff|, relative max |state diff|, alive mask (steps x n) and whether all flags agreed."""
    from deepmimic_amd import streams
    t = model.load_asset(name)
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing, seed=seed)
    oracles, ep = [], np.zeros(n, dtype=np.int64)
    for e in range(n):
        o = Oracle(t); o.reset(o.duration * streams.reset_rand01(seed, e, 0, 0)); oracles.append(o)
    env.reset(kin_times=[o.kin_time() for o in oracles], max_times=np.inf)
    dr, ds, alive, ok = np.zeros((steps, n)), np.zeros((steps, n)), np.zeros((steps, n), dtype=bool), True
    for k in range(steps):
        P, V, T, K, CL, FL = [], [], [], [], [], []
        for o in oracles:
            kp, kv, ko = o.kin_state()
            o.set_action(o.pose_to_action(kp))
            p, v = o.sim_state()
            cm = int(sum(int(c) << j for j, c in enumerate(o.contacts())))
            P.append(p); V.append(v); T.append(o.tar_pose()); K.append(ko)
            CL.append([o.kin_time(), o.kin_time(), 0.0, o.time(), np.inf]); FL.append([int(o.need_new_action()), cm, 1, 1])
        env.set_state(pose=np.array(P), vel=np.array(V), tar=np.array(T), kin=np.array(K), clocks=np.array(CL), flags=np.array(FL, dtype=np.int32))
        out = env.step(None, DT, 20)
        for e, o in enumerate(oracles):
            for u in range(20):
                o.update(DT)
            r = o.calc_reward()
            dr[k, e] = abs(float(out["reward"][e]) - r); alive[k, e] = r != 0.0
            so = o.record_state()
            ds[k, e] = np.abs(out["state"][e] - so).max() / max(1.0, np.abs(so).max())
            ok &= int(out["terminate"][e]) == o.check_terminate() and int(out["valid"][e]) == int(o.check_valid_episode())
            if o.check_terminate() != 0:
                ep[e] += 1
                o.reset(o.duration * streams.reset_rand01(seed, e, int(ep[e]), 0))
    return dr, ds, alive, ok


def goal_rollout_compare(t, precision, lib_path, steps, n, seed, wave_packing=0, action_sigma=0.15, test_mode=False, physics=1):
    """Goal-conditioned task scenes (target_amp / heading_amp; multi-clip datasets, enable_rand_rot_reset): closed-loop rollout
    with seeded random actions THROUGH auto-resets.  The oracle mirrors every device draw: the reset generator's streams 0 (clip
    time), 1 (episode timer), 3 (clip by weight), 4 (yaw) keyed by the episode counter, and the goal generator (stream 2, draw
    counter kept in the goal row).  heading_amp_getup: recovery episodes (train mode) and falls that start a get-up (test mode) are
    mirrored too; strike_amp: hits, success / target-contact termination.  Returns dict of worst deviations + counts."""
    from deepmimic_amd import streams
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing, seed=seed, test_mode=test_mode, physics=physics)
    g0 = env.get_goal_state()                      # draws consumed by the reset inside dm_create
    env.reset()
    ep = env.get_state()["flags"][:, 2].astype(np.int64)
    tmin, tmax = float(t.cfg.time_lim_min), float(t.cfg.time_lim_max)
    if test_mode and t.cfg.time_end_lim_max is not None:
        tmin = tmax = float(t.cfg.time_end_lim_max)
    has_aux = t.goal_kind in (3, 4)
    has_ball = t.goal_kind == 5

    def clip_of(o, e, episode):     # the clip the reset of `episode` selects; episode -1: the one cClipsController::Init selected (stream 6)
        if t.num_clips <= 1:
            return 0
        return o.draw_clip(streams.reset_rand01(seed, e, 0, 6) if episode < 0 else streams.reset_rand01(seed, e, episode, 3))

    def draw(o, e, episode, prev=None):
        clip = clip_of(o, e, episode)
        # the reference draws the clip time over the duration of the clip that was active BEFORE the reset (scenes/SceneImitate.cpp:331-335, 494-500), then selects the
        # new one.  prev: that clip; None = the clip the previous episode's reset selected (no recovery episode in between)
        prev = clip_of(o, e, episode - 1) if prev is None else prev
        kt = o.clip_duration(prev) * streams.reset_rand01(seed, e, episode, 0)
        mt = tmin + (tmax - tmin) * streams.reset_rand01(seed, e, episode, 1) if tmax > tmin else tmax
        yaw = (-np.pi + 2 * np.pi * streams.reset_rand01(seed, e, episode, 4)) if t.cfg.enable_rand_rot_reset else 0.0
        return clip, kt, mt, yaw

    oracles = []
    for e in range(n):
        o = Oracle(t, mode_test=int(test_mode)) if physics == 1 else Oracle(t, mode_test=int(test_mode), physics=physics, max_contacts=env.max_contacts)
        o.goal_rng(seed, e, int(g0[e][11]))
        o.reset_ex(*[(lambda c, k, m, y: (k, m, c, y))(*draw(o, e, int(ep[e]) - 1))][0])
        oracles.append(o)
    active = {e: draw(o, e, int(ep[e]) - 1)[0] for e, o in enumerate(oracles)}      # clip each env's kinematic controller is on
    w = dict(reward=0.0, state=0.0, goal=0.0, goal_state=0.0, resets=0, live=0, flags_ok=True, clips=set(), dist_fail=0, reward_errs=[], goal_errs=[],
             aux=0.0, recoveries=0, succ=0, fail=0, aux_steps=0, desynced=0, scored=0, ball=0.0, ball_moved=0.0, kin=0.0)
    dead = np.zeros(n, bool)       # fp32 only: an env whose episode ended at a different update than the oracle's is not scored from there on
    gs = env.get_goal_state(); clips = env.get_clips(); q = env.query(); qg = env.query_goal()
    for e, o in enumerate(oracles):
        assert clips[e] == o.lib.orc_num_clips(o.h) * 0 + (draw(o, e, int(ep[e]) - 1)[0]), "clip draw mismatch"
        w["goal_state"] = max(w["goal_state"], np.abs(gs[e] - o.goal_state()).max())
        w["goal"] = max(w["goal"], np.abs(qg[e] - o.record_goal()).max())
        w["state"] = max(w["state"], np.abs(q["state"][e] - o.record_state()).max())
    rng = np.random.default_rng(seed + 100)
    for k in range(steps):
        acts = (action_sigma * rng.normal(size=(n, env.A))).astype(np.float32)
        out = env.step(acts, DT, 20, auto_reset=True)
        gs = env.get_goal_state(); clips = env.get_clips(); kin_dev = env.get_state()["kin"]
        aux = env.get_goal_aux() if has_aux else None
        ball = env.get_obj_state() if has_ball else None
        baux = env.get_goal_aux() if has_ball else None
        for e, o in enumerate(oracles):
            if dead[e]:
                continue
            o.set_action(acts[e].astype(np.float64))
            o.control_step(20, DT)
            r = o.calc_reward(); term, end = o.check_terminate(), o.is_episode_end()
            if precision == 32 and not (int(out["terminate"][e]) == term and bool(out["episode_end"][e]) == end):
                dead[e] = True; w["desynced"] += 1; w["flags_ok"] = False
                continue
            w["scored"] += 1
            w["reward"] = max(w["reward"], abs(float(out["reward"][e]) - r)); w["live"] += int(r != 0.0)
            w["reward_errs"].append(abs(float(out["reward"][e]) - r))
            valid = o.check_valid_episode()
            w["flags_ok"] &= int(out["terminate"][e]) == term and bool(out["episode_end"][e]) == end and int(out["valid"][e]) == int(valid)
            w["succ"] += int(term == 2); w["fail"] += int(term == 1)
            if end or not valid:                                    # (the driver resets after an invalid episode as after an ended one)
                c, kt, mt, yaw = draw(o, e, int(ep[e]), active[e])
                if o.maybe_recovery_reset(mt):                      # heading_amp_getup, train mode: the episode goes on as a recovery episode
                    w["recoveries"] += 1; c = int(clips[e])
                else:
                    o.reset_ex(kt, mt, c, yaw); active[e] = c
                ep[e] += 1; w["resets"] += 1
                assert clips[e] == c, "clip draw mismatch after reset"
            w["clips"].add(int(clips[e]))
            so = o.record_state()
            w["state"] = max(w["state"], np.abs(out["state"][e] - so).max() / max(1.0, np.abs(so).max()))
            w["goal"] = max(w["goal"], np.abs(out["goal"][e] - o.record_goal()).max())
            w["goal_errs"].append(np.abs(out["goal"][e] - o.record_goal()).max())
            w["goal_state"] = max(w["goal_state"], np.abs(gs[e] - o.goal_state()).max())
            ko = o.kin_state()[2]               # the kinematic character's origin: follows the simulated root at every cycle boundary of the env's OWN clip
            dk = kin_dev[e]
            w["kin"] = max(w["kin"], float(np.abs(dk[:3] - ko[:3]).max()), 1.0 - abs(float(np.dot(dk[3:7], ko[3:7]))))
            if has_ball:
                ob = o.ball_state()
                w["ball"] = max(w["ball"], np.abs(ball[e] - ob[:13]).max(), np.abs(baux[e][2:7] - ob[13:18]).max())
                w["ball_moved"] = max(w["ball_moved"], float(np.abs(ob[7:13]).max()))
            if has_aux:
                oa = o.goal_state(full=True)[13:15]
                w["aux"] = max(w["aux"], np.abs(aux[e][:2] - oa).max())
                w["aux_steps"] += int((t.goal_kind == 3 and out["goal"][e][3] > 0) or (t.goal_kind == 4 and oa[0] != 0))
    w["reward_mean"] = float(np.mean(w.pop("reward_errs"))); w["goal_mean"] = float(np.mean(w.pop("goal_errs")))
    return w


# ---------------------------------------------------------------------------------------------------------------------------
# Oracles shadowing a SAMPLE of a large device batch (round 5: the 4096-env configurations bench.py times).  Before every control step the
# oracles are put where the device envs ARE (one row each of BatchEnv.get_state: pose, velocity, PD targets, kinematic origin, clocks, flags),
# so each comparison checks one control step (20 updates, 40 substeps) of the production launch -- every wave slot of the chip occupied, high
# block indices, two contexts on two streams -- from identical inputs, on states the device itself reached.  Nothing is written to the device.
class OracleSample:
    def __init__(self, tables, ids, variant="", physics=1, max_contacts=None):
        self.ids = np.asarray(ids, dtype=np.int64)
        self.physics = int(physics)
        # DM-physics v2 (round 6): the oracles run the v2 restatement too and carry the device's persistent ground manifolds (sync(st, manif))
        kw = {} if self.physics == 1 else dict(physics=self.physics, **({} if max_contacts is None else dict(max_contacts=int(max_contacts))))
        self.oracles = [Oracle(tables, variant=variant, **kw) for _ in self.ids]

    def sync(self, st, manif=None):
        """st = get_state() of the WHOLE batch (rows = envs); manif = get_manifolds() of the whole batch (physics 2: without it the oracle's
        manifolds would be those its own trajectory left, not the device's)"""
        if self.physics == 2 and manif is None:
            raise ValueError("physics 2: the oracle needs the device's manifolds (BatchEnv.get_manifolds) next to its state")
        for o, i in zip(self.oracles, self.ids):
            o.set_full_state(st["pose"][i], st["vel"][i], st["tar"][i], st["kin"][i], st["clocks"][i], st["flags"][i])
            if manif is not None:
                o.set_manifolds(manif[i])

    def control_step(self, n_updates=20, dt=DT, actions=None):
        """actions None: open-loop tracking (stream A1); else [len(ids), A] explicit actions (float32-rounded, as they cross the boundary), set before the first
        update of the step like cDeepMimicCore::SetAction at the action boundary (env/deepmimic_env.py:88).  The driver's end-of-episode rule either way; returns per
        sampled env reward, terminate, valid, episode_end, state vector (of the env as the step left it: BEFORE any reset)"""
        n = len(self.oracles)
        r, tm, vd, en, st = np.zeros(n), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), []
        for k, o in enumerate(self.oracles):
            if actions is not None:
                o.set_action(np.asarray(actions[k], dtype=np.float32).astype(np.float64))
            for u in range(n_updates):
                if actions is None and o.need_new_action():   # the device's open-loop stream encodes the kinematic pose at whichever update latches an action
                    kp, _, _ = o.kin_state()
                    o.set_action(o.pose_to_action(kp))
                o.update(dt)
                if o.is_episode_end() or not o.check_valid_episode():      # the driver's rule (DeepMimic.py:62-80), the device's DM_END_EPISODE_EARLY
                    break
            r[k] = o.calc_reward(); tm[k] = o.check_terminate(); vd[k] = int(o.check_valid_episode()); en[k] = int(o.is_episode_end())
            st.append(o.record_state())
        return r, tm, vd, en, np.array(st)


def sampled_compare(get_state, step, tables, ids, steps, conditioning=False, actions=None, physics=1, get_manifolds=None, max_contacts=None, on_step=None):
    """`steps` control steps of a device batch (step() -> dict of whole-batch host arrays, auto-reset on) with the sampled envs `ids`
    compared against oracles re-synchronised from the device before each step.  Returns |reward diff|, relative max |state diff| (NaN on
    steps whose env was auto-reset: the device's observation is then the first of the new episode), live mask (steps x len(ids)), flags ok, #episode ends seen.
    conditioning=True adds a sixth array: |reward(fp64 oracle) - reward(the oracle's own fp32 build)| from the SAME synchronised state -- a control step on
    which the restatement itself, narrowed to float, misses its fp64 self is ill-conditioned in single precision (fp32_step_sensitivity above).
    actions (round 6: the learner's entry, cDeepMimicCore::SetAction -> dm_step_batch(actions)): a callable (k, st0) -> [N, A] float32 actions of control step k
    for the WHOLE batch (st0 = the get_state() the oracles are synchronised from); step is then called as step(acts) and the oracles are fed rows `ids`.
    physics 2: get_manifolds() of the whole batch travels with the state.  on_step(k, st0, out): a hook behind every step (e.g. fallback counters)."""
    smp = OracleSample(tables, ids, physics=physics, max_contacts=max_contacts)
    s32 = OracleSample(tables, ids, variant="f32", physics=physics, max_contacts=max_contacts) if conditioning else None
    n = len(smp.ids)
    dr, ds, alive, ok, ends = np.zeros((steps, n)), np.full((steps, n), np.nan), np.zeros((steps, n), dtype=bool), True, 0
    d32 = np.zeros((steps, n))
    for k in range(steps):
        st0 = get_state()
        mf0 = get_manifolds() if (physics == 2 and get_manifolds is not None) else None
        smp.sync(st0, mf0)
        acts = None if actions is None else np.ascontiguousarray(actions(k, st0), dtype=np.float32)
        out = step() if acts is None else step(acts)
        a_s = None if acts is None else acts[smp.ids]
        r, tm, vd, en, so = smp.control_step(actions=a_s)
        if s32 is not None:
            s32.sync(st0, mf0)
            d32[k] = np.abs(s32.control_step(actions=a_s)[0] - r)
        for j, i in enumerate(smp.ids):
            dr[k, j] = abs(float(out["reward"][i]) - r[j]); alive[k, j] = r[j] != 0.0
            ok &= int(out["terminate"][i]) == tm[j] and int(out["valid"][i]) == vd[j] and int(bool(out["episode_end"][i])) == en[j]
            if en[j] or not vd[j]:
                ends += 1
            else:
                ds[k, j] = np.abs(out["state"][i] - so[j]).max() / max(1.0, np.abs(so[j]).max())
        if on_step is not None:
            on_step(k, st0, out)
    return (dr, ds, alive, ok, ends, d32) if conditioning else (dr, ds, alive, ok, ends)


def tracking_actions(tables, kin_times, clips=None, oracle=None):
    """stream A1 for a whole batch on the host: the action encoding of the reference motion at every env's kinematic clip time (what DM_OPEN_LOOP computes on
    the device), from ONE oracle's motion evaluation (the root block of the pose does not enter an action)."""
    o = oracle if oracle is not None else Oracle(tables)
    return np.array([o.pose_to_action(o.kin_eval(float(t))[0]) for t in np.asarray(kin_times, dtype=np.float64)])
