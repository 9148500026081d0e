"""ctypes binding of libdm_hip.so (include/dm_hip.h) and the batched environment handle.

The HIP library is the only compute path: if it is missing or no GPU is visible the constructor
raises -- there is no CPU fallback in the product.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from .model import AMP_SCENES, BALL_ANG_DAMPING, BALL_FRICTION, BALL_LIN_DAMPING, BALL_MASS, SceneTables
from .model import amp_local_root as _model_amp_local_root

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libdm_hip.so")

DM_DEVICE_PTRS, DM_AUTO_RESET, DM_OPEN_LOOP, DM_NO_EMIT, DM_END_EPISODE_EARLY = 1, 2, 4, 8, 16


class _CreateInfo(C.Structure):
    _fields_ = [("num_envs", C.c_int), ("device_id", C.c_int), ("seed", C.c_uint64), ("precision", C.c_int),
                ("max_contacts", C.c_int), ("env_id_offset", C.c_int), ("wave_packing", C.c_int), ("physics", C.c_int)]


class _SceneTables(C.Structure):
    _fields_ = [
        ("num_joints", C.c_int), ("joint_mat", C.POINTER(C.c_double)), ("body_defs", C.POINTER(C.c_double)),
        ("pd_params", C.POINTER(C.c_double)), ("num_frames", C.c_int), ("frames", C.POINTER(C.c_double)),
        ("loop", C.c_int), ("fall_mask", C.POINTER(C.c_int32)),
        ("num_sim_substeps", C.c_int), ("world_scale", C.c_double), ("gravity", C.c_double * 3),
        ("sync_char_root_pos", C.c_int), ("sync_char_root_rot", C.c_int), ("enable_fall_end", C.c_int),
        ("enable_char_contact_fall", C.c_int), ("enable_root_rot_fail", C.c_int), ("enable_rand_char_placement", C.c_int),
        ("enable_rand_rot_reset", C.c_int), ("time_lim_min", C.c_double), ("time_lim_max", C.c_double),
        ("enable_phase_input", C.c_int), ("record_world_root_pos", C.c_int), ("record_world_root_rot", C.c_int),
        ("query_rate", C.c_double), ("friction", C.c_double), ("erp", C.c_double), ("solver_iters", C.c_int),
        ("disable_self_collision", C.c_int), ("scene_amp", C.c_int), ("enable_amp_obs_local_root", C.c_int),
        ("scene_goal", C.c_int), ("rand_target_time_min", C.c_double), ("rand_target_time_max", C.c_double), ("max_target_dist", C.c_double),
        ("target_succ_dist", C.c_double), ("tar_fail_dist", C.c_double), ("tar_speed", C.c_double), ("enable_min_tar_vel", C.c_int),
        ("pos_reward_scale", C.c_double), ("max_heading_turn_rate", C.c_double), ("sharp_turn_prob", C.c_double), ("speed_change_prob", C.c_double),
        ("tar_speed_min", C.c_double), ("tar_speed_max", C.c_double), ("vel_reward_scale", C.c_double),
        ("num_clips", C.c_int), ("clip_starts", C.POINTER(C.c_int32)), ("clip_weights", C.POINTER(C.c_double)), ("clip_loops", C.POINTER(C.c_int32)),
        ("mode_test", C.c_int), ("getup_time", C.c_double), ("getup_height_root", C.c_double), ("getup_height_head", C.c_double),
        ("recover_episode_prob", C.c_double), ("head_id", C.c_int), ("getup_clip_mask", C.c_int),
        ("tar_near_dist", C.c_double), ("tar_far_prob", C.c_double), ("target_radius", C.c_double), ("target_hit_reset_time", C.c_double),
        ("init_hit_prob", C.c_double), ("hit_tar_speed", C.c_double), ("tar_reward_scale", C.c_double),
        ("target_min", C.c_double * 3), ("target_max", C.c_double * 3), ("strike_mask", C.c_int), ("fail_tar_mask", C.c_int),
        ("rand_tar_obj_time_min", C.c_double), ("rand_tar_obj_time_max", C.c_double), ("min_tar_obj_dist", C.c_double), ("max_tar_obj_dist", C.c_double),
        ("ball_radius", C.c_double), ("ball_mass", C.c_double), ("ball_friction", C.c_double), ("ball_lin_damping", C.c_double), ("ball_ang_damping", C.c_double),
        ("enable_rand_perturbs", C.c_int), ("perturb_time_min", C.c_double), ("perturb_time_max", C.c_double), ("min_perturb", C.c_double),
        ("max_perturb", C.c_double), ("min_perturb_duration", C.c_double), ("max_perturb_duration", C.c_double), ("perturb_part_mask", C.c_int),
    ]


ABI_VERSION = 5        # include/dm_hip.h DM_ABI_VERSION
TAPE_K, TAPE_HDR = 96, 16                 # include/dm_hip.h DM_TAPE_*: the draw tape (dm_set_draw_tape)
TAPE_STRIDE = TAPE_HDR + 8 * TAPE_K
_libs = {}


def load_library(path: Optional[str] = None) -> C.CDLL:
    path = path or os.environ.get("DM_HIP_LIB") or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            "HIP extension %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(deepmimic_amd has no CPU fallback)" % path)
    lib = C.CDLL(path)
    # the CPU fiber-emulator build of the same sources (tests/emu) is test infrastructure: the product refuses it unless the test
    # harness says so, so that no environment variable alone can turn the shipped path into a CPU path
    if not hasattr(lib, "dm_is_emulator"):
        raise RuntimeError("%s does not export dm_is_emulator: stale build, rebuild with __graft_entry__.build()" % path)
    # the structs are mirrored by hand above: refuse a library whose layout is not the one this binding was written against
    sizes = (C.c_int32 * 2)()
    if (not hasattr(lib, "dm_abi_version") or lib.dm_abi_version() != ABI_VERSION or lib.dm_struct_sizes(sizes) != 0
            or (sizes[0], sizes[1]) != (C.sizeof(_CreateInfo), C.sizeof(_SceneTables))):
        raise RuntimeError("%s does not match this binding (ABI version / struct layout): rebuild with __graft_entry__.build()" % path)
    if lib.dm_is_emulator() and os.environ.get("DM_ALLOW_EMULATOR") != "1":
        raise RuntimeError("%s is the CPU emulator build (test infrastructure); deepmimic_amd runs on the HIP library only" % path)
    lib.dm_last_error.restype = C.c_char_p
    lib.dm_motion_duration.restype = C.c_double
    lib.dm_motion_duration.argtypes = [C.c_void_p]
    _libs[path] = lib
    return lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def fill_scene_tables(tables: SceneTables, test_mode: bool = False, self_collision: bool = True, erp: float = 0.0, solver_iters: int = 0):
    """`dm_scene_tables` (include/dm_hip.h) of a parsed scene: (the ctypes struct, the arrays its pointers refer to -- keep them alive for as long as the
    struct is used).  BatchEnv hands it to dm_create; tools/dump_tables.py serialises it for a native caller (tests/native/smoke.c)."""
    c = tables.cfg
    keep = []

    def arr(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a

    st = _SceneTables()
    st.num_joints = tables.num_joints
    st.joint_mat = _dp(arr(tables.joint_mat)); st.body_defs = _dp(arr(tables.body_defs)); st.pd_params = _dp(arr(tables.pd_params))
    st.num_frames = tables.frames.shape[0]; st.frames = _dp(arr(tables.frames)); st.loop = int(tables.loop)
    st.fall_mask = _ip(arr(tables.fall_mask(), np.int32))
    st.num_sim_substeps = int(c.num_sim_substeps); st.world_scale = float(c.world_scale)
    st.gravity = (C.c_double * 3)(*[float(g) for g in c.gravity])
    st.sync_char_root_pos = int(c.sync_char_root_pos); st.sync_char_root_rot = int(c.sync_char_root_rot)
    st.enable_fall_end = int(c.enable_fall_end); st.enable_char_contact_fall = int(c.enable_char_contact_fall)
    st.enable_root_rot_fail = int(c.enable_root_rot_fail); st.enable_rand_char_placement = int(c.enable_rand_char_placement)
    st.enable_rand_rot_reset = int(c.enable_rand_rot_reset)
    # episode timer: test mode uses time_end_lim_max (scenes/RLSceneSimChar.cpp:277-284)
    tmin, tmax = float(c.time_lim_min), float(c.time_lim_max)
    if test_mode and c.time_end_lim_max is not None:
        tmin = tmax = float(c.time_end_lim_max)
    st.time_lim_min, st.time_lim_max = tmin, tmax
    st.enable_phase_input = int(tables.enable_phase_input); st.record_world_root_pos = int(tables.record_world_root_pos)
    st.record_world_root_rot = int(tables.record_world_root_rot); st.query_rate = float(tables.query_rate)
    st.friction = 0.0; st.erp = float(erp); st.solver_iters = int(solver_iters)      # 0: the library's default (10, btContactSolverInfo::m_numIterations); other values are for measurements
    st.disable_self_collision = 0 if self_collision else 1
    st.scene_amp = int(c.scene in AMP_SCENES); st.enable_amp_obs_local_root = int(_model_amp_local_root(c))
    # goal-conditioned AMP task scenes and multi-clip datasets
    st.scene_goal = int(tables.goal_kind)
    for k in ("rand_target_time_min", "rand_target_time_max", "max_target_dist", "target_succ_dist", "tar_fail_dist", "tar_speed",
              "pos_reward_scale", "max_heading_turn_rate", "sharp_turn_prob", "speed_change_prob", "vel_reward_scale"):
        setattr(st, k, float(getattr(c, k)))
    st.enable_min_tar_vel = int(c.enable_min_tar_vel)
    st.tar_speed_min = float(c.tar_speed if c.tar_speed_min is None else c.tar_speed_min)
    st.tar_speed_max = float(c.tar_speed if c.tar_speed_max is None else c.tar_speed_max)
    if tables.clip_starts is not None:
        st.num_clips = int(tables.num_clips)
        st.clip_starts = _ip(arr(tables.clip_starts, np.int32)); st.clip_weights = _dp(arr(tables.clip_weights))
        st.clip_loops = _ip(arr(tables.clip_loops, np.int32))
    else:
        st.num_clips = 0
    # heading_amp_getup / strike_amp
    st.mode_test = int(bool(test_mode))
    st.getup_time = float(tables.getup_time); st.getup_clip_mask = int(tables.getup_clip_mask); st.head_id = int(c.head_id)
    for k in ("getup_height_root", "getup_height_head", "recover_episode_prob", "tar_near_dist", "tar_far_prob", "target_radius",
              "target_hit_reset_time", "init_hit_prob", "hit_tar_speed", "tar_reward_scale"):
        setattr(st, k, float(getattr(c, k)))
    st.target_min = (C.c_double * 3)(*[float(x) for x in c.target_min]); st.target_max = (C.c_double * 3)(*[float(x) for x in c.target_max])
    # body / clip ids become bits of 32-bit masks (c_int): an id that does not name a body part must raise here, not wrap silently
    def _mask(ids, limit, what):
        ids = [int(b) for b in (ids or [])]
        if any(b < 0 or b >= min(int(limit), 31) for b in ids):
            raise ValueError("%s out of range [0, %d): %s" % (what, min(int(limit), 31), sorted(ids)))
        return sum(1 << b for b in set(ids))
    st.strike_mask = _mask(c.strike_bodies, st.num_joints, "strike_bodies"); st.fail_tar_mask = _mask(c.fail_tar_contact_bodies, st.num_joints, "fail_tar_contact_bodies")
    if not 0 <= int(tables.getup_clip_mask) < (1 << 31):
        raise ValueError("getup_motion_ids out of range")
    # dribble_amp: the ball (constants of cSceneDribbleAMP::BuildTarObjs; friction combined with the 0.9 of links and ground)
    for k in ("rand_tar_obj_time_min", "rand_tar_obj_time_max", "min_tar_obj_dist", "max_tar_obj_dist", "ball_radius"):
        setattr(st, k, float(getattr(c, k)))
    st.ball_mass = BALL_MASS; st.ball_friction = BALL_FRICTION * 0.9; st.ball_lin_damping = BALL_LIN_DAMPING; st.ball_ang_damping = BALL_ANG_DAMPING
    # random perturbations: with the default (infinite) interval none ever fires -- the path stays off, as in the reference
    st.enable_rand_perturbs = int(bool(c.enable_rand_perturbs) and np.isfinite(c.perturb_time_min))
    if st.enable_rand_perturbs:
        st.perturb_time_min, st.perturb_time_max = float(c.perturb_time_min), float(c.perturb_time_max)
        st.min_perturb, st.max_perturb = float(c.min_perturb), float(c.max_perturb)
        st.min_perturb_duration, st.max_perturb_duration = float(c.min_pertrub_duration), float(c.max_perturb_duration)
        plist = [int(b) for b in (c.perturb_part_ids or [])]
        if plist != sorted(set(plist)):
            # the reference indexes the list as written (mPerturbPartIDs[RandInt(0, n)], scenes/SceneSimChar.cpp:244-252); the device keeps the ids as a bit mask and
            # takes the idx-th set bit: the same part for the same draw only for an ascending list without repeats
            raise ValueError("perturb_part_ids must be ascending and without repeats (got %s): the device draws the idx-th id of the SET" % plist)
        parts = set(plist)
        if any(b < 0 or b >= min(int(st.num_joints), 31) for b in parts):       # bits of a 32-bit mask, like strike_bodies
            raise ValueError("perturb_part_ids names a body part the character does not have: %s" % sorted(parts))
        st.perturb_part_mask = sum(1 << b for b in parts)
    return st, keep


class BatchEnv:
    """N independent imitate scenes on one GPU (one `dm_ctx`)."""

    def __init__(self, tables: SceneTables, num_envs: int = 1, device_id: int = 0, seed: int = 0,
                 precision: int = 32, max_contacts: int = 20, env_id_offset: int = 0,
                 test_mode: bool = False, lib_path: Optional[str] = None, wave_packing: int = 0, self_collision: bool = True,
                 erp: float = 0.0, physics: int = 1, solver_iters: int = 0):
        self.lib = load_library(lib_path)
        self.tables = tables
        c = tables.cfg
        st, self._keep = fill_scene_tables(tables, test_mode=test_mode, self_collision=self_collision, erp=erp, solver_iters=solver_iters)
        tmin, tmax = float(st.time_lim_min), float(st.time_lim_max)
        self.has_perturbs = bool(st.enable_rand_perturbs)
        if c.timer_type not in ("uniform", "exp"):
            raise ValueError("unsupported timer type %r (util/Timer.cpp:27-45: uniform | exp)" % c.timer_type)
        self._timer = (c.timer_type, tmin, tmax, float(c.time_lim_exp)); self._timer_pinned = tmin == tmax
        self._seed, self._env_off = int(seed) & (2 ** 64 - 1), int(env_id_offset)
        info = _CreateInfo(int(num_envs), int(device_id), int(seed) & (2 ** 64 - 1), int(precision), int(max_contacts), int(env_id_offset), int(wave_packing), int(physics))
        self.h = C.c_void_p()
        self._chk(self.lib.dm_create(C.byref(info), C.byref(st), C.byref(self.h)))
        dims = np.zeros(8, dtype=np.int32)
        self._chk(self.lib.dm_dims(self.h, _ip(dims)))
        self.S, self.G, self.A, self.P, self.J, self.D, self.F, self.N = [int(x) for x in dims]
        self.duration = float(self.lib.dm_motion_duration(self.h))
        self.precision = precision
        pinfo = np.zeros(2, dtype=np.int32)
        self._chk(self.lib.dm_physics_info(self.h, _ip(pinfo)))
        self.physics, self.max_contacts = int(pinfo[0]), int(pinfo[1])      # DM-physics version; effective contact cap per character
        self.amp_size = int(self.lib.dm_amp_obs_size(self.h))      # GetAMPObsSize; 0 unless `--scene imitate_amp`
        self.num_clips = int(tables.num_clips); self.has_obj = tables.goal_kind == 5
        self._has_goal_row = bool(tables.goal_kind != 0 or tables.num_clips > 1 or c.enable_rand_rot_reset)      # dm_host.cpp: st.goal is allocated under the same condition
        if self._timer[0] == "exp":
            # `--timer_type exp` (util/Timer.cpp:27-45, 64-67): the device draws min(min + Exp(time_lim_exp), max) at every reset, in-kernel auto-resets
            # included (round 4).  dm_create's own first reset drew the uniform timer: reset once more under the exponential one.
            self._chk(self.lib.dm_set_timer_exp(self.h, C.c_double(self._timer[3])))
            if not self._timer_pinned:
                self.reset()

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libdm_hip: %s" % self.lib.dm_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.dm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- cDeepMimicCore-style operations on all envs
    def reset(self, env_ids: Optional[Sequence[int]] = None, kin_times=None, max_times=None):
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        n = self.N if ids is None else ids.size
        kt = None if kin_times is None else np.ascontiguousarray(np.broadcast_to(kin_times, (n,)), dtype=np.float64)
        mt = None if max_times is None else np.ascontiguousarray(np.broadcast_to(max_times, (n,)), dtype=np.float64)
        self._chk(self.lib.dm_reset(self.h, _ip(ids), n, _dp(kt), _dp(mt)))

    def _check_auto_reset(self, auto_reset):
        pass            # (until round 4 the in-kernel auto-reset refused `--timer_type exp`; the device draws it now)

    def set_time_limits(self, time_lim_min: float, time_lim_max: float, time_lim_exp: Optional[float] = None):
        self._chk(self.lib.dm_set_time_limits(self.h, C.c_double(time_lim_min), C.c_double(time_lim_max)))
        self._timer = (self._timer[0], float(time_lim_min), float(time_lim_max), self._timer[3] if time_lim_exp is None else float(time_lim_exp))
        self._timer_pinned = time_lim_min == time_lim_max          # test mode: cRLSceneSimChar::ResetTimers pins the limit, whatever the type
        if self._timer[0] == "exp":
            self._chk(self.lib.dm_set_timer_exp(self.h, C.c_double(self._timer[3])))

    def set_sample_count(self, sample_count: int, test_mode: bool = False):
        """cRLSceneSimChar::SetSampleCount / SetMode for the whole batch: anneal the episode-length limits (model.timer_limits)."""
        from . import model
        self.set_time_limits(*model.timer_limits(self.tables.cfg, test_mode, sample_count), model.timer_exp(self.tables.cfg, test_mode, sample_count))

    def set_action(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.N, self.A)
        self._chk(self.lib.dm_set_action(self.h, _fp(a), 0))

    def update(self, timestep: float, n_updates: int = 1):
        self._chk(self.lib.dm_update(self.h, C.c_double(timestep), int(n_updates)))

    def query(self):
        s = np.zeros((self.N, self.S), np.float32); r = np.zeros(self.N, np.float32)
        t = np.zeros(self.N, np.int32); v = np.zeros(self.N, np.int32); e = np.zeros(self.N, np.int32); nn = np.zeros(self.N, np.int32)
        self._chk(self.lib.dm_query(self.h, _fp(s), _fp(r), _ip(t), _ip(v), _ip(e), _ip(nn), 0))
        out = dict(state=s, reward=r, terminate=t, valid=v, episode_end=e, need_new_action=nn)
        if self.G:
            out["goal"] = self.last_goals()
        return out

    def step(self, actions=None, timestep: float = 1.0 / 600, n_updates: int = 20, auto_reset=False, open_loop=False, amp=False, end_early=None):
        """amp=True (imitate_amp scenes): also returns "amp_obs" = RecordAMPObsAgent at the end of the step (before any auto reset).
        end_early (default: follows auto_reset): an env whose episode is over after an update takes no further updates in this call,
        as the reference's driver ends an episode at the update where IsEpisodeEnd turns true (DeepMimic.py:62-80)."""
        end_early = auto_reset if end_early is None else end_early
        self._check_auto_reset(auto_reset)
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.float32).reshape(self.N, self.A)
        s = np.zeros((self.N, self.S), np.float32); r = np.zeros(self.N, np.float32)
        t = np.zeros(self.N, np.int32); v = np.zeros(self.N, np.int32); e = np.zeros(self.N, np.int32)
        flags = (DM_AUTO_RESET if auto_reset else 0) | (DM_OPEN_LOOP if open_loop else 0) | (DM_END_EPISODE_EARLY if end_early else 0)
        if amp:
            o = np.zeros((self.N, self.amp_size), np.float32)
            self._chk(self.lib.dm_step_batch_amp(self.h, _fp(a), C.c_double(timestep), int(n_updates), _fp(s), _fp(r), _ip(t), _ip(v), _ip(e), _fp(o), flags))
            out = dict(state=s, reward=r, terminate=t, valid=v, episode_end=e, amp_obs=o)
            if self.G:
                out["goal"] = self.last_goals()
            return out
        self._chk(self.lib.dm_step_batch(self.h, _fp(a), C.c_double(timestep), int(n_updates), _fp(s), _fp(r), _ip(t), _ip(v), _ip(e), flags))
        out = dict(state=s, reward=r, terminate=t, valid=v, episode_end=e)
        if self.G:
            out["goal"] = self.last_goals()
        return out

    def step_envs(self, env_ids, actions=None, timestep: float = 1.0 / 600, n_updates: int = 20, auto_reset=False, open_loop=False, end_early=None,
                  amp=False, want_clocks=True):
        """`step` for a subset of the envs (include/dm_hip.h dm_step_envs): compact arrays, row i <-> env_ids[i]; the other envs are not touched."""
        end_early = auto_reset if end_early is None else end_early
        ids = np.ascontiguousarray(env_ids, dtype=np.int32).ravel()
        n = ids.size
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.float32).reshape(n, self.A)
        s = np.zeros((n, self.S), np.float32); r = np.zeros(n, np.float32)
        t = np.zeros(n, np.int32); v = np.zeros(n, np.int32); e = np.zeros(n, np.int32)
        o = np.zeros((n, self.amp_size), np.float32) if (amp and self.amp_size) else None
        clk = np.zeros((n, 5)) if want_clocks else None
        flags = (DM_AUTO_RESET if auto_reset else 0) | (DM_OPEN_LOOP if open_loop else 0) | (DM_END_EPISODE_EARLY if end_early else 0)
        self._chk(self.lib.dm_step_envs(self.h, _ip(ids), int(n), _fp(a), C.c_double(timestep), int(n_updates), _fp(s), _fp(r), _ip(t), _ip(v), _ip(e),
                                        _fp(o), _dp(clk), flags))
        out = dict(state=s, reward=r, terminate=t, valid=v, episode_end=e)
        if o is not None:
            out["amp_obs"] = o
        if clk is not None:
            out["clocks"] = clk
        return out

    def set_env_keys(self, env_ids, seeds):
        """include/dm_hip.h dm_set_env_keys: the listed envs draw like env 0 of one-env contexts created with `seeds` (then reset them)"""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32).ravel(); sd = np.ascontiguousarray(seeds, dtype=np.uint64).ravel()
        assert ids.size == sd.size
        self._chk(self.lib.dm_set_env_keys(self.h, _ip(ids), int(ids.size), sd.ctypes.data_as(C.POINTER(C.c_uint64))))

    def set_draw_tape(self, tape):
        """include/dm_hip.h dm_set_draw_tape: bind (N x TAPE_STRIDE doubles) or, with None, unbind the reference's generators as position-indexed tables"""
        if tape is None:
            self._chk(self.lib.dm_set_draw_tape(self.h, None)); return
        t = np.ascontiguousarray(tape, dtype=np.float64).reshape(self.N, TAPE_STRIDE)
        self._chk(self.lib.dm_set_draw_tape(self.h, _dp(t)))

    def set_draw_tape_envs(self, env_ids, rows):
        """include/dm_hip.h dm_set_draw_tape_envs: the tape rows of the listed envs (n x TAPE_STRIDE)"""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32).ravel()
        if ids.size == 0:                   # bind the rows the device already holds
            self._chk(self.lib.dm_set_draw_tape_envs(self.h, None, 0, None)); return
        r = np.ascontiguousarray(rows, dtype=np.float64).reshape(ids.size, TAPE_STRIDE)
        self._chk(self.lib.dm_set_draw_tape_envs(self.h, _ip(ids), int(ids.size), _dp(r)))

    def draw_tape_state_envs(self, env_ids):
        ids = np.ascontiguousarray(env_ids, dtype=np.int32).ravel()
        out = np.zeros((ids.size, TAPE_HDR))
        self._chk(self.lib.dm_get_draw_tape_state_envs(self.h, _ip(ids), int(ids.size), _dp(out)))
        return out

    def draw_tape_state(self):
        """include/dm_hip.h dm_get_draw_tape_state: the tape headers after a launch (N x TAPE_HDR: positions consumed, saved normal deviate, error flag, ...)"""
        out = np.zeros((self.N, TAPE_HDR))
        self._chk(self.lib.dm_get_draw_tape_state(self.h, _dp(out)))
        return out

    def clip_table(self):
        """(durations, cdf) of the dataset's clips (include/dm_hip.h dm_clip_table)"""
        nc = max(1, int(self.num_clips))
        d, c = np.zeros(nc), np.zeros(nc)
        self._chk(self.lib.dm_clip_table(self.h, _dp(d), _dp(c)))
        return d, c

    def set_mode(self, test_mode: bool):
        """cRLScene::SetMode for the goal scenes' device-side logic (the episode-timer limits are set_time_limits' business)"""
        self._chk(self.lib.dm_set_mode(self.h, int(bool(test_mode))))

    def get_goal_aux(self):
        """N x 8: [0:2] heading_amp_getup {get-up timer, -} / strike_amp {target hit, hit time}; [2:7] dribble_amp {ball position at the
        last action, target-object timer time / limit}"""
        out = np.zeros((self.N, 8))
        self._chk(self.lib.dm_get_goal_aux(self.h, _dp(out)))
        return out

    def set_goal_aux(self, aux):
        aux = np.ascontiguousarray(aux, dtype=np.float64).reshape(self.N, 8)
        self._chk(self.lib.dm_set_goal_aux(self.h, _dp(aux)))

    def get_perturb_state(self):
        """enable_rand_perturbs: N x 16 = time since the last perturbation, time of the next, draw counter, two slots of {part + 1 (0 = free),
        force xyz, duration, elapsed} (include/dm_hip.h)"""
        out = np.zeros((self.N, 16))
        self._chk(self.lib.dm_get_perturb_state(self.h, _dp(out)))
        return out

    def set_perturb_state(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(self.N, 16)
        self._chk(self.lib.dm_set_perturb_state(self.h, _dp(p)))

    def get_obj_state(self):
        """dribble_amp: the ball of every env, N x 13 = pos(3), rot wxyz(4), vel(3), ang vel(3)"""
        out = np.zeros((self.N, 13))
        self._chk(self.lib.dm_get_obj_state(self.h, _dp(out)))
        return out

    def set_obj_state(self, obj):
        obj = np.ascontiguousarray(obj, dtype=np.float64).reshape(self.N, 13)
        self._chk(self.lib.dm_set_obj_state(self.h, _dp(obj)))

    def last_goals(self):
        g = np.zeros((self.N, max(self.G, 1)), np.float32)
        self._chk(self.lib.dm_last_goals(self.h, _fp(g)))
        return g

    def last_goals_device(self, dst_ptr: int):
        """RecordGoal of the last step into a device buffer (N x G float32), on the ctx stream, no host sync"""
        self._chk(self.lib.dm_last_goals_device(self.h, C.c_void_p(int(dst_ptr))))

    def query_goal(self):
        """RecordGoal for every env (goal scenes): N x G"""
        g = np.zeros((self.N, max(self.G, 1)), np.float32)
        self._chk(self.lib.dm_query_goal(self.h, _fp(g), 0))
        return g[:, :self.G]

    def get_goal_state(self):
        out = np.zeros((self.N, 12))
        self._chk(self.lib.dm_get_goal_state(self.h, _dp(out)))
        return out

    def set_goal_state(self, gs):
        gs = np.ascontiguousarray(gs, dtype=np.float64).reshape(self.N, 12)
        self._chk(self.lib.dm_set_goal_state(self.h, _dp(gs)))

    def get_clips(self):
        out = np.zeros(self.N, np.int32)
        self._chk(self.lib.dm_get_clips(self.h, _ip(out)))
        return out

    def set_clips(self, clips):
        """include/dm_hip.h dm_set_clips: which clip counts as active before the next reset (the draw tape draws the reset time over ITS duration)"""
        cl = np.ascontiguousarray(np.broadcast_to(clips, (self.N,)), dtype=np.int32)
        self._chk(self.lib.dm_set_clips(self.h, _ip(cl)))

    def amp_expert_clips(self, n: int, clips=None, times=None, ground_h=None):
        o = np.zeros((int(n), self.amp_size), np.float32)
        cc = None if clips is None else np.ascontiguousarray(np.broadcast_to(clips, (n,)), dtype=np.int32)
        tt = None if times is None else np.ascontiguousarray(np.broadcast_to(times, (n,)), dtype=np.float64)
        gh = None if ground_h is None else np.ascontiguousarray(np.broadcast_to(ground_h, (n,)), dtype=np.float64)
        self._chk(self.lib.dm_amp_expert_clips(self.h, int(n), _ip(cc), _dp(tt), _dp(gh), _fp(o)))
        return o

    def query_amp(self):
        """RecordAMPObsAgent for every env (scenes/SceneImitateAMP.cpp:101-113)."""
        o = np.zeros((self.N, self.amp_size), np.float32)
        self._chk(self.lib.dm_query_amp(self.h, _fp(o), 0))
        return o

    def amp_expert(self, n: int, times=None, ground_h=None):
        """n expert observations (RecordAMPObsExpert, :115-138); times None -> ~U[0, duration) from the ctx generator."""
        o = np.zeros((int(n), self.amp_size), np.float32)
        tt = None if times is None else np.ascontiguousarray(np.broadcast_to(times, (n,)), dtype=np.float64)
        gh = None if ground_h is None else np.ascontiguousarray(np.broadcast_to(ground_h, (n,)), dtype=np.float64)
        self._chk(self.lib.dm_amp_expert(self.h, int(n), _dp(tt), _dp(gh), _fp(o), 0))
        return o

    def step_device(self, actions_ptr, states_ptr, rewards_ptr, term_ptr, valid_ptr, end_ptr,
                    timestep: float = 1.0 / 600, n_updates: int = 20, auto_reset=False, open_loop=False, amp_ptr=0, end_early=None):
        """Same as step() on raw device pointers (ints), asynchronous on the ctx stream."""
        end_early = auto_reset if end_early is None else end_early
        self._check_auto_reset(auto_reset)
        flags = DM_DEVICE_PTRS | (DM_AUTO_RESET if auto_reset else 0) | (DM_OPEN_LOOP if open_loop else 0) | (DM_END_EPISODE_EARLY if end_early else 0)
        vp = lambda p: C.c_void_p(p) if p else None
        if amp_ptr:
            self._chk(self.lib.dm_step_batch_amp(self.h, vp(actions_ptr), C.c_double(timestep), int(n_updates), vp(states_ptr),
                                                 vp(rewards_ptr), vp(term_ptr), vp(valid_ptr), vp(end_ptr), vp(amp_ptr), flags))
            return
        self._chk(self.lib.dm_step_batch(self.h, vp(actions_ptr), C.c_double(timestep), int(n_updates), vp(states_ptr),
                                         vp(rewards_ptr), vp(term_ptr), vp(valid_ptr), vp(end_ptr), flags))

    def set_stream(self, stream_handle: int):
        # torch's default stream has the null handle, which dm_set_stream reads as "back to the ctx's own stream": select the legacy
        # default stream explicitly, so that `env.set_stream(torch.cuda.current_stream().cuda_stream)` orders the launches against
        # the caller's work whatever stream is current
        if not stream_handle:
            self._chk(self.lib.dm_set_stream_default(self.h))
        else:
            self._chk(self.lib.dm_set_stream(self.h, C.c_void_p(stream_handle)))

    def own_stream(self) -> int:
        """handle of the ctx's own non-blocking HIP stream (wrap it with torch.cuda.ExternalStream to order torch work against it)"""
        own = C.c_void_p(0)
        self._chk(self.lib.dm_get_stream(self.h, C.byref(own), None))
        return int(own.value or 0)

    def use_own_stream(self):
        """back to the ctx's own non-blocking stream"""
        self._chk(self.lib.dm_set_stream(self.h, None))

    def synchronize(self):
        self._chk(self.lib.dm_synchronize(self.h))

    def bench_rollout(self, warmup: int, steps: int, timestep=1.0 / 600, n_updates=20, auto_reset=True, open_loop=True):
        ms = C.c_double(0)
        self._check_auto_reset(auto_reset)
        flags = (DM_AUTO_RESET if auto_reset else 0) | (DM_OPEN_LOOP if open_loop else 0) | (DM_END_EPISODE_EARLY if auto_reset else 0)
        self._chk(self.lib.dm_bench_rollout(self.h, int(warmup), int(steps), C.c_double(timestep), int(n_updates), flags, None, None, C.byref(ms)))
        return ms.value

    def offsets_scales(self):
        so, ss = np.zeros(self.S), np.zeros(self.S); ao, as_, amin, amax = np.zeros(self.A), np.zeros(self.A), np.zeros(self.A), np.zeros(self.A)
        g = np.zeros(self.S, np.int32)
        self._chk(self.lib.dm_build_offsets_scales(self.h, _dp(so), _dp(ss), _dp(ao), _dp(as_), _dp(amin), _dp(amax), _ip(g)))
        return dict(state_offset=so, state_scale=ss, action_offset=ao, action_scale=as_, action_min=amin, action_max=amax, state_norm_groups=g)

    # ---- snapshots / taps
    def get_state(self):
        pose = np.zeros((self.N, self.P)); vel = np.zeros((self.N, self.P)); tar = np.zeros((self.N, self.P))
        kin = np.zeros((self.N, 7)); clk = np.zeros((self.N, 5)); flg = np.zeros((self.N, 4), np.int32)
        self._chk(self.lib.dm_get_state(self.h, _dp(pose), _dp(vel), _dp(tar), _dp(kin), _dp(clk), _ip(flg)))
        return dict(pose=pose, vel=vel, tar=tar, kin=kin, clocks=clk, flags=flg)

    def snapshot(self):
        """Everything the next control step depends on, as host arrays: character state and clocks, and for the task scenes the goal row
        (target, timers, draw counter), its scene-specific block and the free body.  Taken at an action boundary (after a step / reset /
        query) it is a checkpoint: `restore` + the same actions reproduce the rollout bit for bit (the AMP pose history is re-latched by
        the first update after a boundary, so it is not part of it)."""
        snap = self.get_state()
        if self._has_goal_row:        # decided up front, as the host allocates the row: a failed device copy must raise, not yield a rollback without the goal
            snap["goal"] = self.get_goal_state(); snap["aux"] = self.get_goal_aux()
        if self.has_obj:
            snap["obj"] = self.get_obj_state()
        if self.has_perturbs:
            snap["pert"] = self.get_perturb_state()
        if self.physics == 2:
            snap["manif"] = self.get_manifolds()        # the ground manifolds are contact HISTORY: without them a restored v2 env rebuilds one point per substep
        return snap

    def restore(self, snap):
        self.set_state(pose=snap["pose"], vel=snap["vel"], tar=snap["tar"], kin=snap["kin"], clocks=snap["clocks"], flags=snap["flags"])
        if "goal" in snap:
            self.set_goal_state(snap["goal"]); self.set_goal_aux(snap["aux"])
        if "obj" in snap:
            self.set_obj_state(snap["obj"])
        if "pert" in snap:
            self.set_perturb_state(snap["pert"])
        if "manif" in snap:
            self.set_manifolds(snap["manif"])           # after set_state, which empties them

    def get_manifolds(self):
        """DM-physics v2: N x J x 25 (count, 4 x (body-frame point, plane point x z, distance)) -- include/dm_hip.h dm_get_manifolds"""
        m = np.zeros((self.N, self.J, 25))
        self._chk(self.lib.dm_get_manifolds(self.h, _dp(m)))
        return m

    def set_manifolds(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64).reshape(self.N, self.J, 25)
        self._chk(self.lib.dm_set_manifolds(self.h, _dp(m)))

    def set_state(self, pose=None, vel=None, tar=None, kin=None, clocks=None, flags=None):
        f = lambda a, sh: None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(sh)
        pose, vel, tar = f(pose, (self.N, self.P)), f(vel, (self.N, self.P)), f(tar, (self.N, self.P))
        kin, clocks = f(kin, (self.N, 7)), f(clocks, (self.N, 5))
        flags = None if flags is None else np.ascontiguousarray(flags, dtype=np.int32).reshape(self.N, 4)
        self._chk(self.lib.dm_set_state(self.h, _dp(pose), _dp(vel), _dp(tar), _dp(kin), _dp(clocks), _ip(flags)))

    def set_tau(self, tau):
        tau = np.ascontiguousarray(tau, dtype=np.float64).reshape(self.N, self.D)
        self._chk(self.lib.dm_set_tau(self.h, _dp(tau)))

    def probe(self, what: int, dt: float):
        self._chk(self.lib.dm_probe(self.h, int(what), C.c_double(dt)))

    def debug(self, name: str):
        shapes = {"H": (self.N, self.D, self.D), "C": (self.N, self.D), "vstar": (self.N, self.D), "lambda": (self.N, 64),
                  "rows": (self.N, 2), "tau": (self.N, self.D), "kin_pose": (self.N, self.P), "kin_vel": (self.N, self.P),
                  "reward_terms": (self.N, 5), "links": (self.N, self.J, 21), "prof": (self.N, 16), "fallback": (self.N,), "borrowed": (self.N,)}
        out = np.zeros(shapes[name])
        self._chk(self.lib.dm_get_debug(self.h, name.encode(), _dp(out)))
        return out


class RefRand:
    """cRand of the reference (util/Rand.cpp) as libdm_hip.so implements it with the same <random> types (include/dm_hip.h: dm_refrand_*):
    a host that replays the reference's call order draws the reference's numbers.  Used by the cDeepMimicCore facade (DM_RNG=reference)."""

    def __init__(self, seed: int = 0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        for f in ("dm_refrand_double", "dm_refrand_exp", "dm_refrand_norm"):
            getattr(self.lib, f).restype = C.c_double
        self.h = C.c_void_p()
        if self.lib.dm_refrand_create(C.c_ulong(int(seed) & (2 ** 64 - 1)), C.byref(self.h)) != 0:
            raise RuntimeError("dm_refrand_create failed")

    def seed(self, seed: int):
        self.lib.dm_refrand_seed(self.h, C.c_ulong(int(seed) & (2 ** 64 - 1)))

    def rand_double(self, lo: float = 0.0, hi: float = 1.0) -> float:
        return float(self.lib.dm_refrand_double(self.h, C.c_double(lo), C.c_double(hi)))

    def rand_exp(self, lam: float) -> float:
        return float(self.lib.dm_refrand_exp(self.h, C.c_double(lam)))

    def rand_norm(self, mean: float = 0.0, stdev: float = 1.0) -> float:
        return float(self.lib.dm_refrand_norm(self.h, C.c_double(mean), C.c_double(stdev)))

    def rand_int(self) -> int:
        return int(self.lib.dm_refrand_int(self.h))

    def rand_int_range(self, lo: int, hi: int) -> int:
        return int(self.lib.dm_refrand_int_range(self.h, int(lo), int(hi)))

    def rand_uint(self) -> int:
        return int(self.lib.dm_refrand_uint(self.h))

    def discard(self, n: int):
        if self.lib.dm_refrand_discard(self.h, C.c_long(int(n))) != 0:
            raise RuntimeError("dm_refrand_discard failed")

    def norm_state(self):
        """(available, value) of the second deviate the normal distribution keeps for its next call"""
        a, v = C.c_int(0), C.c_double(0.0)
        self.lib.dm_refrand_norm_state(self.h, 0, C.byref(a), C.byref(v))
        return int(a.value), float(v.value)

    def set_norm_state(self, avail: int, value: float):
        a, v = C.c_int(int(avail)), C.c_double(float(value))
        self.lib.dm_refrand_norm_state(self.h, 1, C.byref(a), C.byref(v))

    def state(self):
        """(engine state, normal available, normal value): everything a snapshot needs"""
        s = C.c_ulong(0)
        self.lib.dm_refrand_engine_state(self.h, 0, C.byref(s))
        return (int(s.value),) + self.norm_state()

    def set_state(self, st):
        s = C.c_ulong(int(st[0]))
        if self.lib.dm_refrand_engine_state(self.h, 1, C.byref(s)) != 0:
            raise RuntimeError("dm_refrand_engine_state failed")
        self.set_norm_state(st[1], st[2])

    def tape(self, normal: bool = False, integer: bool = False):
        """include/dm_hip.h dm_refrand_tape: the position-indexed tables of this generator (u [K], e [K], n3 [3 K] or None, i2 [2 K] or None); the generator
        is not advanced"""
        u, e = np.zeros(TAPE_K), np.zeros(TAPE_K)
        n3 = np.zeros(3 * TAPE_K) if normal else None
        i2 = np.zeros(2 * TAPE_K) if integer else None
        if self.lib.dm_refrand_tape(self.h, TAPE_K, _dp(u), _dp(e), _dp(n3), _dp(i2)) != 0:
            raise RuntimeError("dm_refrand_tape failed")
        return u, e, n3, i2

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dm_refrand_destroy(self.h); self.h = None
        except Exception:
            pass


class Comm:
    """dm_comm of the C-ABI (include/dm_hip.h): the RCCL communicator behind dm_gather_records, for hosts that shard without
    torch.  `unique_id` = the 128 bytes rank 0 got from Comm.unique_id(), shipped to every rank by the caller."""

    def __init__(self, world: int, rank: int, device_id: int = 0, unique_id: Optional[bytes] = None, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        self.h = C.c_void_p()
        uid = None if unique_id is None else C.create_string_buffer(bytes(unique_id), 128)
        if self.lib.dm_comm_create(uid, int(world), int(rank), int(device_id), C.byref(self.h)) != 0:
            raise RuntimeError("libdm_hip: %s" % self.lib.dm_last_error().decode())
        self.world, self.rank = world, rank

    @staticmethod
    def unique_id(lib_path: Optional[str] = None) -> bytes:
        lib = load_library(lib_path)
        buf = C.create_string_buffer(128)
        if lib.dm_comm_unique_id(buf) != 0:
            raise RuntimeError("libdm_hip: %s" % lib.dm_last_error().decode())
        return buf.raw

    def gather(self, env: "BatchEnv", slot: int, send_ptr: int, recv_ptr: int, count: int):
        """async all-gather of `count` floats per rank, ordered after the work already enqueued on env's stream"""
        env._chk(self.lib.dm_gather_records(env.h, self.h, int(slot), C.c_void_p(send_ptr), C.c_void_p(recv_ptr), C.c_size_t(count)))

    def wait(self, env: "BatchEnv", slot: int):
        env._chk(self.lib.dm_gather_wait(env.h, self.h, int(slot)))

    def close(self):
        if self.h:
            self.lib.dm_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
