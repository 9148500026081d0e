"""ctypes binding to the CPU oracle (oracle/libdm_oracle*.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd import model  # noqa: E402  (scene constants only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

CFG_KEYS = ["num_sim_substeps", "world_scale", "grav_x", "grav_y", "grav_z", "sync_root_pos", "sync_root_rot",
            "enable_fall_end", "enable_contact_fall", "enable_root_rot_fail", "enable_rand_placement",
            "enable_phase_input", "record_world_root_pos", "record_world_root_rot", "query_rate",
            "friction", "erp", "solver_iters", "max_contacts", "self_collision", "scene_amp", "amp_local_root",
            "scene_goal", "rand_rot_reset", "tar_time_min", "tar_time_max", "max_tar_dist", "tar_succ_dist", "tar_fail_dist", "tar_speed",
            "pos_reward_scale", "min_tar_vel", "max_turn_rate", "sharp_turn_prob", "speed_change_prob", "tar_speed_min", "tar_speed_max",
            "vel_reward_scale",
            "mode_test", "getup_time", "getup_height_root", "getup_height_head", "head_id", "recover_prob", "getup_clip_mask",
            "tar_near_dist", "tar_far_prob", "target_radius", "hit_reset_time", "init_hit_prob", "hit_tar_speed", "tar_reward_scale",
            "tmin_x", "tmin_y", "tmin_z", "tmax_x", "tmax_y", "tmax_z", "strike_mask", "fail_tar_mask",
            "obj_time_min", "obj_time_max", "min_obj_dist", "max_obj_dist", "ball_radius", "ball_mass", "ball_friction", "ball_lin_damp", "ball_ang_damp",
            "perturb_on", "perturb_time_min", "perturb_time_max", "perturb_min", "perturb_max", "perturb_dur_min", "perturb_dur_max", "perturb_part_mask", "physics"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def build(target="all"):
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, target])


def load_lib(variant=""):
    name = {"": "libdm_oracle.so", "f32": "libdm_oracle_f32.so", "native": "libdm_oracle_native.so"}[variant]
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build("native" if variant == "native" else "all")
    lib = C.CDLL(path)
    lib.orc_create.restype = C.c_void_p
    for f in ("orc_motion_duration", "orc_calc_reward", "orc_calc_reward_terms", "orc_time", "orc_kin_time",
              "orc_phase", "orc_rollout", "orc_rollout_auto_reset"):
        getattr(lib, f).restype = C.c_double
    return lib


class Oracle:
    """One imitate scene on the CPU oracle."""

    def __init__(self, tables, variant="", **cfg_overrides):
        self.lib = lib = load_lib(variant)
        self.t = tables
        n = lib.orc_cfg_count()
        assert n == len(CFG_KEYS)
        cfg = np.zeros(n)
        lib.orc_cfg_default(_d(cfg))
        c = tables.cfg
        vals = dict(num_sim_substeps=c.num_sim_substeps, world_scale=c.world_scale,
                    grav_x=c.gravity[0], grav_y=c.gravity[1], grav_z=c.gravity[2],
                    sync_root_pos=c.sync_char_root_pos, sync_root_rot=c.sync_char_root_rot,
                    enable_fall_end=c.enable_fall_end, enable_contact_fall=c.enable_char_contact_fall,
                    enable_root_rot_fail=c.enable_root_rot_fail, enable_rand_placement=c.enable_rand_char_placement,
                    enable_phase_input=tables.enable_phase_input, record_world_root_pos=tables.record_world_root_pos,
                    record_world_root_rot=tables.record_world_root_rot, query_rate=tables.query_rate,
                    scene_amp=(c.scene != "imitate"), amp_local_root=model.amp_local_root(c),
                    scene_goal=tables.goal_kind, rand_rot_reset=c.enable_rand_rot_reset, tar_time_min=c.rand_target_time_min,
                    tar_time_max=c.rand_target_time_max, max_tar_dist=c.max_target_dist, tar_succ_dist=c.target_succ_dist,
                    tar_fail_dist=c.tar_fail_dist, tar_speed=c.tar_speed, pos_reward_scale=c.pos_reward_scale, min_tar_vel=c.enable_min_tar_vel,
                    max_turn_rate=c.max_heading_turn_rate, sharp_turn_prob=c.sharp_turn_prob, speed_change_prob=c.speed_change_prob,
                    tar_speed_min=(c.tar_speed if c.tar_speed_min is None else c.tar_speed_min),
                    tar_speed_max=(c.tar_speed if c.tar_speed_max is None else c.tar_speed_max), vel_reward_scale=c.vel_reward_scale,
                    getup_time=tables.getup_time, getup_height_root=c.getup_height_root, getup_height_head=c.getup_height_head, head_id=c.head_id,
                    recover_prob=c.recover_episode_prob, getup_clip_mask=tables.getup_clip_mask,
                    tar_near_dist=c.tar_near_dist, tar_far_prob=c.tar_far_prob, target_radius=c.target_radius, hit_reset_time=c.target_hit_reset_time,
                    init_hit_prob=c.init_hit_prob, hit_tar_speed=c.hit_tar_speed, tar_reward_scale=c.tar_reward_scale,
                    tmin_x=c.target_min[0], tmin_y=c.target_min[1], tmin_z=c.target_min[2], tmax_x=c.target_max[0], tmax_y=c.target_max[1], tmax_z=c.target_max[2],
                    strike_mask=sum(1 << int(b) for b in (c.strike_bodies or [])), fail_tar_mask=sum(1 << int(b) for b in (c.fail_tar_contact_bodies or [])),
                    obj_time_min=c.rand_tar_obj_time_min, obj_time_max=c.rand_tar_obj_time_max, min_obj_dist=c.min_tar_obj_dist, max_obj_dist=c.max_tar_obj_dist,
                    ball_radius=c.ball_radius, ball_mass=model.BALL_MASS, ball_friction=model.BALL_FRICTION * 0.9,
                    ball_lin_damp=model.BALL_LIN_DAMPING, ball_ang_damp=model.BALL_ANG_DAMPING,
                    perturb_on=bool(c.enable_rand_perturbs) and np.isfinite(c.perturb_time_min), perturb_time_min=c.perturb_time_min, perturb_time_max=c.perturb_time_max,
                    perturb_min=c.min_perturb, perturb_max=c.max_perturb, perturb_dur_min=c.min_pertrub_duration, perturb_dur_max=c.max_perturb_duration,
                    perturb_part_mask=sum(1 << int(b) for b in set(c.perturb_part_ids or [])))
        vals.update(cfg_overrides)
        for k, v in vals.items():
            cfg[CFG_KEYS.index(k)] = float(v)
        self.cfg = cfg
        jm = np.ascontiguousarray(tables.joint_mat, dtype=np.float64)
        bd = np.ascontiguousarray(tables.body_defs, dtype=np.float64)
        pd = np.ascontiguousarray(tables.pd_params, dtype=np.float64)
        fr = np.ascontiguousarray(tables.frames, dtype=np.float64)
        fall = np.ascontiguousarray(tables.fall_mask(), dtype=np.int32)
        n0 = fr.shape[0] if tables.clip_starts is None else int(tables.clip_starts[1])
        self.h = C.c_void_p(lib.orc_create(_d(jm), _d(bd), jm.shape[0], _d(pd), _d(fr), n0, int(tables.loop if tables.clip_starts is None else tables.clip_loops[0]),
                                           fall.ctypes.data_as(_ip), _d(cfg)))
        if tables.clip_starts is not None:
            cs = np.ascontiguousarray(tables.clip_starts, dtype=np.int32); cl = np.ascontiguousarray(tables.clip_loops, dtype=np.int32)
            cw = np.ascontiguousarray(tables.clip_weights, dtype=np.float64)
            lib.orc_set_clips(self.h, _d(fr), cs.ctypes.data_as(_ip), cl.ctypes.data_as(_ip), _d(cw), len(cw))
        lib.orc_clip_duration.restype = C.c_double
        dims = np.zeros(5, dtype=np.int32)
        lib.orc_dims(self.h, dims.ctypes.data_as(_ip))
        self.J, self.P, self.A, self.S, self.F = [int(x) for x in dims]
        self.duration = lib.orc_motion_duration(self.h)

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    def reset(self, kin_time=0.0, max_time=np.inf):
        self.lib.orc_reset(self.h, C.c_double(kin_time), C.c_double(max_time))

    def set_action(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.A
        self.lib.orc_set_action(self.h, _d(a))

    def update(self, dt):
        self.lib.orc_update(self.h, C.c_double(dt))

    def need_new_action(self):
        return bool(self.lib.orc_need_new_action(self.h))

    def record_state(self):
        out = np.zeros(self.S)
        self.lib.orc_record_state(self.h, _d(out))
        return out

    def amp_obs_size(self):
        return int(self.lib.orc_amp_obs_size(self.h))

    def amp_obs_agent(self):
        out = np.zeros(self.amp_obs_size())
        self.lib.orc_amp_obs_agent(self.h, _d(out))
        return out

    def amp_obs_expert(self, t):
        out = np.zeros(self.amp_obs_size())
        self.lib.orc_amp_obs_expert(self.h, C.c_double(t), _d(out))
        return out

    def prev_state(self):
        p, v = np.zeros(self.P), np.zeros(self.P)
        self.lib.orc_prev_state(self.h, _d(p), _d(v))
        return p, v

    def set_prev_state(self, pose, vel):
        p = np.ascontiguousarray(pose, dtype=np.float64); v = np.ascontiguousarray(vel, dtype=np.float64)
        self.lib.orc_set_prev_state(self.h, _d(p), _d(v))

    def calc_reward(self):
        return self.lib.orc_calc_reward(self.h)

    def calc_reward_terms(self):
        t = np.zeros(5)
        r = self.lib.orc_calc_reward_terms(self.h, _d(t))
        return r, t

    def check_terminate(self):
        return int(self.lib.orc_check_terminate(self.h))

    def is_episode_end(self):
        return bool(self.lib.orc_is_episode_end(self.h))

    def check_valid_episode(self):
        return bool(self.lib.orc_check_valid_episode(self.h))

    def time(self):
        return self.lib.orc_time(self.h)

    def kin_time(self):
        return self.lib.orc_kin_time(self.h)

    def phase(self):
        return self.lib.orc_phase(self.h)

    def sim_state(self):
        p, v = np.zeros(self.P), np.zeros(self.P)
        self.lib.orc_get_sim_state(self.h, _d(p), _d(v))
        return p, v

    def set_sim_state(self, p, v):
        p = np.ascontiguousarray(p, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        self.lib.orc_set_sim_state(self.h, _d(p), _d(v))

    def set_full_state(self, pose, vel, tar, kin_origin, clocks, flags):
        """put the oracle where a DEVICE env is: one row of every array of `BatchEnv.get_state()` (the kinematic pose is re-posed from time + origin)"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        pose, vel, tar, kin_origin, clocks = f(pose), f(vel), f(tar), f(kin_origin), f(clocks)
        flags = np.ascontiguousarray(flags, dtype=np.int32)
        self.lib.orc_set_full_state(self.h, _d(pose), _d(vel), _d(tar), _d(kin_origin), _d(clocks), flags.ctypes.data_as(_ip))

    def manifolds(self):
        """physics 2: the persistent ground manifolds, J x 25 (the layout of BatchEnv.get_manifolds)"""
        m = np.zeros((self.J, 25))
        self.lib.orc_get_manifolds(self.h, _d(m))
        return m

    def set_manifolds(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64).reshape(self.J, 25)
        self.lib.orc_set_manifolds(self.h, _d(m))

    def kin_state(self):
        p, v, o = np.zeros(self.P), np.zeros(self.P), np.zeros(7)
        self.lib.orc_get_kin_state(self.h, _d(p), _d(v), _d(o))
        return p, v, o

    def tar_pose(self):
        out = np.zeros(self.P)
        self.lib.orc_get_tar_pose(self.h, _d(out))
        return out

    def tau(self):
        out = np.zeros(self.P)
        self.lib.orc_get_tau(self.h, _d(out))
        return out

    def contacts(self):
        out = np.zeros(self.J, dtype=np.int32)
        self.lib.orc_get_contacts(self.h, out.ctypes.data_as(_ip))
        return out

    def kin_eval(self, t):
        p, v = np.zeros(self.P), np.zeros(self.P)
        self.lib.orc_kin_eval(self.h, C.c_double(t), _d(p), _d(v))
        return p, v

    def motion_frame(self, f):
        fr, fv, t = np.zeros(self.P), np.zeros(self.P), C.c_double(0)
        self.lib.orc_motion_frame(self.h, f, _d(fr), _d(fv), C.byref(t))
        return fr, fv, t.value

    def mass_bias(self, which, pose, vel):
        H, Cb = np.zeros((self.P, self.P)), np.zeros(self.P)
        pose = np.ascontiguousarray(pose, dtype=np.float64)
        vel = np.ascontiguousarray(vel, dtype=np.float64)
        self.lib.orc_mass_bias(self.h, which, _d(pose), _d(vel), _d(H), _d(Cb))
        return H, Cb

    def spd_tau(self, dt):
        out = np.zeros(self.P)
        self.lib.orc_spd_tau(self.h, C.c_double(dt), _d(out))
        return out

    def set_tau(self, tau):
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        self.lib.orc_set_tau(self.h, _d(tau))

    def substep(self, h):
        self.lib.orc_substep(self.h, C.c_double(h))

    def vstar(self):
        out = np.zeros(self.P)
        self.lib.orc_get_vstar(self.h, _d(out))
        return out

    def num_rows(self):
        return int(self.lib.orc_dbg_num_rows(self.h))

    def num_contacts(self):
        return int(self.lib.orc_dbg_num_contacts(self.h))

    def contact_list(self):
        n = self.num_contacts()
        out = np.zeros((max(n, 1), 9))
        self.lib.orc_dbg_contacts(self.h, _d(out))
        return out[:n]

    def num_self_contacts(self):
        return int(self.lib.orc_dbg_num_self_contacts(self.h))

    def links(self):
        out = np.zeros((self.J, 21))
        self.lib.orc_get_links(self.h, _d(out))
        return out

    def calc_com(self, pose, vel):
        c, v = np.zeros(3), np.zeros(3)
        pose = np.ascontiguousarray(pose, dtype=np.float64)
        vel = np.ascontiguousarray(vel, dtype=np.float64)
        self.lib.orc_calc_com(self.h, _d(pose), _d(vel), _d(c), _d(v))
        return c, v

    def pose_to_action(self, pose):
        a = np.zeros(self.A)
        pose = np.ascontiguousarray(pose, dtype=np.float64)
        self.lib.orc_pose_to_action(self.h, _d(pose), _d(a))
        return a

    def rollout(self, steps, updates_per_step=20, dt=1.0 / 600, actions=None, want_states=False):
        rewards = np.zeros(steps)
        states = np.zeros((steps, self.S)) if want_states else None
        ap = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.float64)
            assert actions.shape == (steps, self.A)
            ap = _d(actions)
        secs = self.lib.orc_rollout(self.h, steps, updates_per_step, C.c_double(dt), ap, _d(rewards),
                                    _d(states) if want_states else None)
        return secs, rewards, states

    def rollout_auto_reset(self, steps, seed, env_id, t0, time_lim_min=np.inf, time_lim_max=np.inf, updates_per_step=20, dt=1.0 / 600):
        """open-loop tracking with auto-reset (bench.py's workload); returns (seconds, resets, reward sum, live steps)"""
        stats = np.zeros(3)
        secs = self.lib.orc_rollout_auto_reset(self.h, int(steps), int(updates_per_step), C.c_double(dt), C.c_uint64(int(seed)), int(env_id),
                                               C.c_double(t0), C.c_double(time_lim_min), C.c_double(time_lim_max), _d(stats))
        return secs, int(stats[0]), float(stats[1]), int(stats[2])

    def control_step(self, n_updates=20, dt=1.0 / 600, end_early=True):
        """n_updates scene updates; with end_early the loop stops after the update at which the episode is over, the way the
        reference's driver does (DeepMimic.py:62-80).  Criteria as on the device (EnvSim::episode_over_now, kin_pre): terminate != Null,
        the episode timer, or an invalid episode (a link velocity beyond 100).  Returns the number of updates run."""
        for u in range(n_updates):
            self.update(dt)
            if end_early and (self.is_episode_end() or not self.check_valid_episode()):      # the driver tests both after every update
                return u + 1
        return n_updates

    # ---- goal scenes / multi-clip datasets
    def goal_rng(self, seed, env_id, draws=0):
        self.lib.orc_goal_rng(self.h, C.c_uint64(int(seed)), C.c_uint64(int(env_id)), C.c_uint64(int(draws)))

    def reset_ex(self, kin_time, max_time=np.inf, clip=0, yaw=0.0):
        self.lib.orc_reset_ex(self.h, C.c_double(kin_time), C.c_double(max_time), int(clip), C.c_double(yaw))

    def draw_clip(self, u):
        return int(self.lib.orc_draw_clip(self.h, C.c_double(u)))

    def clip_duration(self, c):
        return float(self.lib.orc_clip_duration(self.h, int(c)))

    def record_goal(self):
        out = np.zeros(int(self.lib.orc_goal_dim(self.h)))
        self.lib.orc_record_goal(self.h, _d(out))
        return out

    def goal_state(self, full=False):
        """the device's goal row: target, heading, speed, target timer, action bookkeeping, draw counter (12) [+ clip, aux0, aux1]"""
        out = np.zeros(15)
        self.lib.orc_goal_state(self.h, _d(out))
        return out if full else out[:12]

    def perturb_state(self):
        """the 16-double row of include/dm_hip.h dm_get_perturb_state"""
        out = np.zeros(16)
        self.lib.orc_perturb_state(self.h, _d(out))
        return out

    def set_perturb_state(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(16)
        self.lib.orc_set_perturb_state(self.h, _d(p))

    def num_perturbs(self):
        return int(self.lib.orc_num_perturbs(self.h))

    def ball_state(self):
        """dribble_amp: ball pos(3), rot wxyz(4), vel(3), ang vel(3), ball pos at the last action(3), target-object timer time / max"""
        out = np.zeros(18)
        self.lib.orc_ball_state(self.h, _d(out))
        return out

    def set_ball(self, b13):
        b = np.ascontiguousarray(b13, dtype=np.float64)
        self.lib.orc_set_ball(self.h, _d(b))

    def maybe_recovery_reset(self, max_time=np.inf):
        return bool(self.lib.orc_maybe_recovery_reset(self.h, C.c_double(max_time)))

    def amp_obs_expert_clip(self, clip, t, ground_h=0.0):
        out = np.zeros(self.amp_obs_size())
        self.lib.orc_amp_obs_expert_clip(self.h, int(clip), C.c_double(t), C.c_double(ground_h), _d(out))
        return out
