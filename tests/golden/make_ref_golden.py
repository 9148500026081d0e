"""Generate tests/golden/ref_vectors.npz: outputs of the REFERENCE's own compiled sources (oracle/_ref/libdm_ref.so, built
by oracle/build_ref.sh from the unmodified files under /root/reference/DeepMimicCore) on seeded inputs.

Unlike oracle_rollouts.json (the oracle's own outputs), every array written here is computed by reference code:
cKinTree / cRBDModel / cRBDUtil / cMathUtil / cKinCharacter / cMotionController / cMotion, and -- since round 3 -- the reference's
compiled ROUTINES (oracle/ref_standins.cpp: sim/ImpPDController.cpp CalcControlForces, sim/CtController.cpp RecordState,
scenes/SceneImitate.cpp CalcRewardImitate, scenes/SceneImitateAMP.cpp BuildAMPObs, scenes/SceneTargetAMP.cpp / SceneHeadingAMP.cpp
CalcReward + RecordGoal, scenes/SceneStrikeAMP.cpp / SceneDribbleAMP.cpp CalcReward + RecordGoal + their hit / contact / success / distance checks and
the dribble task state) for the SPD torque, the state vector, the reward, the AMP observation and the task rewards / goal vectors;
the compositions of oracle/ref_glue.cpp are evaluated next to them and must agree (only the five reward error TERMS, which the
routine does not hand out, are still taken from the composition).

The file travels with the repository, so the checks in tests/test_ref_golden.py (oracle vs golden, emulator build of the
device code vs golden, and -- marked gpu -- the HIP kernels vs golden) run where /root/reference does not exist.

Run from the repo root in a container that has /root/reference:  python tests/golden/make_ref_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_lib  # noqa: E402
from deepmimic_amd import model  # noqa: E402
from ref_lib import Components, RefKinChar, RefRig, Skel, random_pose_vel  # noqa: E402

N_STATES = 8
N_KIN = 16
SCENES = ["humanoid3d_walk", "dog3d_pace", "humanoid3d_spinkick"]


def gains(t):
    P = t.pose_dim
    kp, kd = np.zeros(P), np.zeros(P)
    for j in range(1, t.num_joints):
        off = int(t.joint_mat[j, model.JD_PARAM_OFFSET])
        sz = model.joint_param_size(int(t.joint_mat[j, model.JD_TYPE]), False)
        kp[off:off + sz] = t.pd_params[j, 0]
        kd[off:off + sz] = t.pd_params[j, 1]
    return kp, kd


def main():
    ref = Components("ref")
    out = {}
    for si, name in enumerate(SCENES):
        t = model.load_asset(name)
        sk = Skel(ref, t)
        kc = RefKinChar(ref, os.path.join("/root/reference", t.cfg.character_file), os.path.join("/root/reference", t.cfg.motion_file))
        char = os.path.splitext(os.path.basename(t.cfg.character_file))[0]
        rig = RefRig(ref, char, motion=os.path.splitext(os.path.basename(t.cfg.motion_file))[0])
        rng = np.random.default_rng(1000 + si)
        kp, kd = gains(t)
        w = t.joint_mat[:, model.JD_DIFF_W].copy(); w = w / np.abs(w).sum()
        P, J = t.pose_dim, t.num_joints
        flags = int(t.enable_phase_input) | (int(t.record_world_root_pos) << 1) | (int(t.record_world_root_rot) << 2)
        g = {k: [] for k in ("pose", "vel", "tar", "H", "C", "spd_tau", "joint_world", "body_world", "link_vel", "com", "com_vel",
                             "kin_time", "kin_origin", "kin_pose", "kin_vel", "reward_terms", "reward", "state", "phase", "amp_obs", "amp_prev_pose", "amp_prev_vel")}
        for i in range(N_STATES):
            # kinematic character: a clip time (beyond one cycle for looping clips) and an origin (translation + yaw)
            tk = rng.uniform(0, 2.6 * kc.duration if kc.loop else 0.95 * kc.duration)
            yaw = rng.uniform(-np.pi, np.pi) if i % 2 else 0.0
            opos = np.array([rng.normal() * 2, 0.01 * (i % 3), rng.normal() * 2]) if i else np.zeros(3)
            orot = np.array([np.cos(yaw / 2), 0.0, np.sin(yaw / 2), 0.0])
            kc.set_origin(opos, orot)
            kpose, kvel = kc.eval(tk)
            # sim state: the kin pose perturbed (so that the reward is informative, not ~0), lifted clear of the ground
            p, v = kpose.copy(), kvel.copy()
            p[0:3] += rng.normal(size=3) * 0.04
            p[1] += 0.25
            for j in range(0, J):
                off, ty = int(t.joint_mat[j, model.JD_PARAM_OFFSET]), int(t.joint_mat[j, model.JD_TYPE])
                if j == 0:
                    q = p[3:7] + 0.05 * rng.normal(size=4); q /= np.linalg.norm(q); p[3:7] = q if q[0] >= 0 else -q
                    v[0:6] += rng.normal(size=6) * 0.2
                elif ty == model.JT_SPHERICAL:
                    q = p[off:off + 4] + 0.12 * rng.normal(size=4); q /= np.linalg.norm(q); p[off:off + 4] = q if q[0] >= 0 else -q
                    v[off:off + 3] += rng.normal(size=3) * 0.8
                elif ty == model.JT_REVOLUTE:
                    p[off] += 0.1 * rng.normal(); v[off] += rng.normal() * 0.8
            tar, _ = random_pose_vel(t, rng)
            tar[:7] = 0
            H, C = sk.mass_bias(p, v)
            jw, bw = sk.world_trans(p)
            com, comv = sk.com(p, v)
            phase = tk / kc.duration - np.floor(tk / kc.duration)
            st = np.zeros(t.state_dim)
            n = ref.lib.ref_record_state(sk.h, ref_lib._d(p), ref_lib._d(v), ref_lib.C.c_double(phase), ref_lib.C.c_double(0.0), flags, ref_lib._d(st))
            assert n == t.state_dim, (n, t.state_dim)
            rt = ref_lib.ref_reward_terms(ref, sk, p, v, kpose, kvel, w, 0.0, opos[1])
            # the compiled routines on the same inputs; the compositions must agree with them
            rig.set_state(p, v); rig.set_targets(tar)
            tau_routine = rig.spd_tau(1 / 600)
            assert np.abs(tau_routine - sk.spd_tau(p, v, tar, kp, kd, 1 / 600)).max() < 1e-11 * max(1.0, np.abs(tau_routine).max())
            st_routine = rig.record_state(phase, 0.0)
            assert np.abs(st_routine - st).max() < 1e-12 * max(1.0, np.abs(st).max())
            rig.kin_set(tk, opos, orot)
            rew_routine = rig.reward_imitate(0.0)
            assert abs(rew_routine - rt[5]) < 1e-12, (name, i, rew_routine, rt[5])
            st, rt[5] = st_routine, rew_routine
            # AMP observation (imitate_amp, enable_amp_obs_local_root as the 34 shipped arg files have it: false) with a history one control period back
            pp, pv = kc.eval(tk - 1 / 30)
            amp = rig.amp_obs(pp, pv, False)
            for k, val in (("pose", p), ("vel", v), ("tar", tar), ("H", H), ("C", C), ("spd_tau", tau_routine), ("amp_obs", amp), ("amp_prev_pose", pp), ("amp_prev_vel", pv),
                           ("joint_world", jw), ("body_world", bw), ("link_vel", sk.link_vel(p, v)), ("com", com), ("com_vel", comv),
                           ("kin_time", tk), ("kin_origin", np.r_[opos, orot]), ("kin_pose", kpose), ("kin_vel", kvel),
                           ("reward_terms", rt[:5]), ("reward", rt[5]), ("state", st), ("phase", phase)):
                g[k].append(np.array(val))
        for k, val in g.items():
            out["%s/%s" % (name, k)] = np.array(val)
        # kinematic character alone: N_KIN times, identity origin -- frames, frame velocities, duration
        kc.set_origin(np.zeros(3), np.array([1.0, 0, 0, 0]))
        times = np.r_[0.0, kc.duration, rng.uniform(0, 3.0 * kc.duration if kc.loop else kc.duration, size=N_KIN - 2)]
        ev = [kc.eval(tt) for tt in times]
        out["%s/kin_eval_times" % name] = times
        out["%s/kin_eval_pose" % name] = np.array([e[0] for e in ev])
        out["%s/kin_eval_vel" % name] = np.array([e[1] for e in ev])
        fr = [kc.frame(f) for f in range(kc.F)]
        out["%s/frame_vel" % name] = np.array([f[1] for f in fr])
        out["%s/frame_time" % name] = np.array([f[2] for f in fr])
        out["%s/duration" % name] = np.array(kc.duration)
        out["%s/cycle_root_delta" % name] = kc.cycle_root_delta()
    # task scenes (scenes/SceneTargetAMP.cpp, SceneHeadingAMP.cpp as compiled): reward and goal vector on scripted goal states.
    # par layout = oracle/ref_standins.cpp ref2_task_scene
    for name, kind in (("amp_target_zombie", 1), ("amp_heading_zombie", 2)):
        t = model.load_asset(name); c = t.cfg
        char = os.path.splitext(os.path.basename(c.character_file))[0]
        kc = RefKinChar(ref, os.path.join("/root/reference", c.character_file), "/root/reference/data/motions/humanoid3d_walk.txt")     # (poses to stand in: any clip)
        rig = RefRig(ref, char)
        sk = Skel(ref, t)
        rng = np.random.default_rng(2000 + kind)
        g = {k: [] for k in ("pose", "vel", "par", "reward", "goal")}
        for i in range(12):
            tk = rng.uniform(0, kc.duration)
            kc.set_origin(np.array([rng.normal(), 0.0, rng.normal()]), np.array([np.cos(0.4 * i), 0.0, np.sin(0.4 * i), 0.0]))
            p, v = kc.eval(tk)
            p[1] += 0.05; v[0:3] += rng.normal(size=3) * 0.3
            com, _ = sk.com(p, v)
            speed = rng.uniform(0.5, 2.0)
            tar = np.array([p[0], 0.0, p[2]]) + rng.normal(size=3) * np.array([2.0, 0.0, 2.0]) * (0.1 if i % 5 == 4 else 1.0)    # (some targets inside the success radius)
            prev_t = rng.uniform(0.1, 3.0)
            prev_com = com - np.array([rng.normal() * 0.03, rng.normal() * 0.01, rng.normal() * 0.03]) - 19 / 600 * speed * np.array([np.cos(0.4 * i), 0, -np.sin(0.4 * i)])
            par = [tar[0], 0.0, tar[2], speed, c.target_succ_dist, c.tar_fail_dist if np.isfinite(c.tar_fail_dist) else 1e30, float(c.enable_min_tar_vel), c.pos_reward_scale,
                   rng.uniform(-np.pi, np.pi), c.vel_reward_scale, prev_t, prev_com[0], prev_com[1], prev_com[2], prev_t + 19 / 600, 0.0]
            rig.set_state(p, v)
            r, goal = rig.task_scene(kind, par)
            for k, val in (("pose", p), ("vel", v), ("par", par), ("reward", r), ("goal", goal)):
                g[k].append(np.array(val))
        for k, val in g.items():
            out["task/%s/%s" % (name, k)] = np.array(val)
    # heading_amp_getup (scenes/SceneHeadingAMPGetup.cpp as compiled): heading reward / get-up reward by the get-up timer, goal with the get-up phase,
    # CheckGettingUp, HasFallenContact (no contact fall while getting up)
    if True:
        name, kind = "amp_heading_getup", 3
        t = model.load_asset(name); c = t.cfg
        char = os.path.splitext(os.path.basename(c.character_file))[0]
        kc = RefKinChar(ref, os.path.join("/root/reference", c.character_file), "/root/reference/data/motions/humanoid3d_walk.txt")
        rig = RefRig(ref, char)
        sk = Skel(ref, t)
        rng = np.random.default_rng(3003)
        g = {k: [] for k in ("pose", "vel", "par", "reward", "goal", "extra")}
        for i in range(12):
            tk = rng.uniform(0, kc.duration)
            kc.set_origin(np.array([rng.normal(), 0.0, rng.normal()]), np.array([np.cos(0.4 * i), 0.0, np.sin(0.4 * i), 0.0]))
            p, v = kc.eval(tk)
            p[1] += 0.05 - (0.5 if i % 3 == 1 else 0.0); v[0:3] += rng.normal(size=3) * 0.3          # (some roots low: the get-up reward is not saturated)
            com, _ = sk.com(p, v)
            speed = rng.uniform(0.5, 2.0)
            prev_t = rng.uniform(0.1, 3.0)
            prev_com = com - np.array([rng.normal() * 0.03, rng.normal() * 0.01, rng.normal() * 0.03]) - 19 / 600 * speed * np.array([np.cos(0.4 * i), 0, -np.sin(0.4 * i)])
            timer = t.getup_time * (rng.uniform(0.05, 0.95) if i % 2 else rng.uniform(1.0, 1.5))           # odd cases are getting up
            fallen = 1.0 if i in (4, 5) else 0.0
            par = [0.0, 0.0, 0.0, speed, c.target_succ_dist, 1e30, float(c.enable_min_tar_vel), c.pos_reward_scale, (0.4 * i + rng.normal() * 0.4 + np.pi) % (2 * np.pi) - np.pi, c.vel_reward_scale,
                   prev_t, prev_com[0], prev_com[1], prev_com[2], prev_t + 19 / 600, fallen, t.getup_time, timer, c.getup_height_root, c.getup_height_head, float(c.head_id), 0.0]
            rig.set_state(p, v)
            r, goal, extra = rig.task_scene(kind, par, extras=2)
            for k, val in (("pose", p), ("vel", v), ("par", par), ("reward", r), ("goal", goal), ("extra", extra)):
                g[k].append(np.array(val))
        for k, val in g.items():
            out["task/%s/%s" % (name, k)] = np.array(val)
    # strike_amp / dribble_amp (scenes/SceneStrikeAMP.cpp, SceneDribbleAMP.cpp as compiled): reward, goal vector, the scenes' own checks and
    # -- dribble -- the 15 task entries of the state vector, on scripted goal states that reach every branch
    for name, kind in (("amp_strike_punch", 4), ("amp_dribble_zombie", 5)):
        t = model.load_asset(name); c = t.cfg
        char = os.path.splitext(os.path.basename(c.character_file))[0]
        kc = RefKinChar(ref, os.path.join("/root/reference", c.character_file), "/root/reference/data/motions/humanoid3d_walk.txt")
        rig = RefRig(ref, char)
        sk = Skel(ref, t)
        rng = np.random.default_rng(3000 + kind)
        g = {k: [] for k in ("pose", "vel", "par", "reward", "goal", "extra")}
        for i in range(16):
            tk = rng.uniform(0, kc.duration)
            kc.set_origin(np.array([rng.normal(), 0.0, rng.normal()]), np.array([np.cos(0.4 * i), 0.0, np.sin(0.4 * i), 0.0]))
            p, v = kc.eval(tk)
            p[1] += 0.05; v[0:3] += rng.normal(size=3) * 0.3
            com, _ = sk.com(p, v)
            body = sk.world_trans(p)[1][:, 9:12]
            speed = float(c.tar_speed)
            prev_t = rng.uniform(0.1, 3.0)
            prev_com = com - np.array([rng.normal() * 0.03, rng.normal() * 0.01, rng.normal() * 0.03]) - 19 / 600 * speed * np.array([np.cos(0.4 * i), 0, -np.sin(0.4 * i)])
            fallen = 1.0 if i == 13 else 0.0
            if kind == 4:
                sb, fb = int(c.strike_bodies[0]), [int(b) for b in c.fail_tar_contact_bodies]
                mode = i % 4                                   # 0 far, 1 near (target by the strike body), 2 hit, 3 far / forbidden-body contact / success by turns
                if mode == 1: tar = body[sb] + rng.normal(size=3) * np.array([0.25, 0.1, 0.25])
                elif mode == 3 and i % 8 == 3: tar = body[fb[i % len(fb)]] + rng.normal(size=3) * 0.05
                else: tar = np.array([p[0], 0.0, p[2]]) + np.array([rng.normal() * 2.5, rng.uniform(c.target_min[1], c.target_max[1]), rng.normal() * 2.5])
                hit = 1.0 if mode == 2 or i == 15 else 0.0
                scene_t = rng.uniform(3.0, 8.0)
                hit_t = scene_t - (rng.uniform(2.05, 3.0) if i == 15 or i == 10 else rng.uniform(0.0, 1.9))
                par = [tar[0], tar[1], tar[2], speed, c.target_succ_dist, c.tar_fail_dist, float(c.enable_min_tar_vel), c.pos_reward_scale, 0.0, c.vel_reward_scale,
                       prev_t, prev_com[0], prev_com[1], prev_com[2], prev_t + 19 / 600, fallen,
                       c.tar_near_dist, c.target_radius, c.tar_reward_scale, c.hit_tar_speed, float(1 << sb), float(sum(1 << b for b in fb)), hit, hit_t, scene_t, c.target_hit_reset_time]
                if mode == 1: v[3:6] += rng.normal(size=3) * 2.0                  # some spin: the strike body moves
                if i == 5:                                                         # a clean hit: strike body inside the sphere, moving at the target
                    tar = body[sb] + rng.normal(size=3) * 0.03; d = tar - p[0:3]; d[1] = 0.0
                    v[0:3] = 3.0 * d / np.linalg.norm(d); par[0:3] = tar
                rig.set_state(p, v)
                r, goal, extra = rig.task_scene(kind, par, extras=3)
            else:
                ball = np.array([p[0], c.ball_radius, p[2]]) + np.array([rng.normal(), 0.0, rng.normal()]) * (0.8 if i != 11 else 0.0) + (np.array([25.0, 0.0, 5.0]) if i == 11 else 0.0)
                q = rng.normal(size=4); q /= np.linalg.norm(q)
                bv, bw = rng.normal(size=3) * 0.8, rng.normal(size=3) * 2.0
                tar = ball + np.array([rng.normal(), 0.0, rng.normal()]) * (0.2 if i % 5 == 4 else 3.0) + (np.array([-4.0, 0.0, 24.0]) if i == 8 else 0.0); tar[1] = 0.0
                prev_ball = ball - 19 / 600 * bv + rng.normal(size=3) * 0.01
                par = [tar[0], 0.0, tar[2], speed, c.target_succ_dist, 1e30, float(c.enable_min_tar_vel), c.pos_reward_scale, 0.0, c.vel_reward_scale,
                       prev_t, prev_com[0], prev_com[1], prev_com[2], prev_t + 19 / 600, fallen,
                       ball[0], ball[1], ball[2], q[0], q[1], q[2], q[3], bv[0], bv[1], bv[2], bw[0], bw[1], bw[2], prev_ball[0], prev_ball[1], prev_ball[2],
                       c.max_target_dist, c.max_tar_obj_dist]
                rig.set_state(p, v)
                r, goal, extra = rig.task_scene(kind, par, extras=19)
            for k, val in (("pose", p), ("vel", v), ("par", par), ("reward", r), ("goal", goal), ("extra", extra)):
                g[k].append(np.array(val))
        for k, val in g.items():
            out["task/%s/%s" % (name, k)] = np.array(val)
    # scalar / quaternion functions of cMathUtil on a fixed input set
    rng = np.random.default_rng(7)
    ops, ins, outs = [], [], []
    for _ in range(64):
        q0 = rng.normal(size=4); q0 /= np.linalg.norm(q0)
        q1 = rng.normal(size=4); q1 /= np.linalg.norm(q1)
        e = rng.normal(size=3) * rng.choice([0.1, 1.0, 3.0, 7.0])
        for op, inp in ((0, e), (1, q0), (2, np.r_[q0, q1]), (3, np.r_[q0, q1, 1 / 30]), (4, np.r_[q0, q1, 1 / 30]), (5, q0),
                        (7, np.r_[q0, q1, rng.uniform()]), (9, q0), (13, q0), (15, rng.uniform(-3, 3, size=3))):
            pad = np.zeros(9); pad[:len(inp)] = inp
            o = ref.math_op(op, inp)
            po = np.zeros(9); po[:len(o)] = o
            ops.append(op); ins.append(pad); outs.append(po)
    out["math/op"] = np.array(ops); out["math/in"] = np.array(ins); out["math/out"] = np.array(outs)
    path = os.path.join(HERE, "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
