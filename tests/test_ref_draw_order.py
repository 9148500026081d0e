"""The reference's draw ORDER on the device: with `DM_RNG=reference` the one-env cDeepMimicCore facade serves every draw of a scene --
the ones the kernels make too: clip choice, reset clip time, random yaw, goal re-sampling, target timers, perturbations, recovery coin -- from the
reference's two generators (cMathUtil::gRand and the scene's cScene::mRand) in the reference's call order, through the draw tape (include/dm_hip.h
DM_TAPE_*).  Checked against the reference's OWN compiled scene classes (oracle/_ref: scenes/Scene{,Imitate,ImitateAMP,TargetAMP,HeadingAMP,
HeadingAMPGetup,StrikeAMP,DribbleAMP}.cpp, anim/ClipsController.cpp, util/{Rand,Timer,MathUtil}.cpp built unmodified; oracle/ref_standins.cpp
ref3_* issue the drawing calls of Init / Reset / Update in the order the reference's Init / ResetScene / Update make them) seeded with the same seed:
every clip id, clip time, episode limit, yaw, target heading / speed / timer and perturbation must be EQUAL; quantities that add a character position
(target positions) are equal to the accuracy of that position (the two sides sample the clip with their own, 1e-12-equal, kinematics)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import ref_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "deepmimic_amd", "compat")
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "DeepMimicCore")), reason="needs the reference checkout (arg files, motion data) next to oracle/_ref")

KINDS = {"imitate_amp": 0, "target_amp": 1, "heading_amp": 2, "heading_amp_getup": 3, "strike_amp": 4, "dribble_amp": 5, "imitate": 6}
# keys whose parsing names a builder that lives in a Bullet translation unit (the stand-ins replace what those builders would build)
SKIP_KEYS = ("char_types", "char_ctrls", "kin_ctrl", "terrain_file", "character_files", "char_ctrl_files", "motion_file", "agent_files", "scene", "arg_file")


# sessions logged into tests/golden/ref_draws.npz for the GPU box (tests/golden/make_ref_draw_golden.py): asset of deepmimic_amd/assets, the reference arguments the
# asset was compiled from (tools/compile_assets.py), seed, resets, control steps per episode, annealing
GOLDEN_SESSIONS = {
    "heading4": ("amp_heading_clips4", lambda: ["--arg_file", "args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt", "--scene", "heading_amp"], 20231, 12, 8, {4: 32000000}),
    "strike": ("amp_strike_punch", lambda: ["--arg_file", "args/train_amp_strike_humanoid3d_walk_punch_args.txt", "--motion_file",
                                            os.path.join(ROOT, "tools", "datasets", "humanoid3d_clips_walk_punch_local.txt")], 77, 12, 8, None),
}
GOLDEN_SESSIONS.update({
    "dribble": ("amp_dribble_zombie", lambda: ["--arg_file", "args/train_amp_dribble_humanoid3d_zombie_args.txt"], 12345, 8, 8, None),
    "getup": ("amp_heading_getup", lambda: ["--arg_file", "args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt"], 99, 10, 8, {0: 64000000}),
})
PERTURB_ARGS = ["--enable_rand_perturbs", "true", "--perturb_time_min", "0.05", "--perturb_time_max", "0.2", "--min_pertrub_duration", "0.02", "--max_perturb_duration", "0.08",
                "--timer_type", "exp", "--time_lim_min", "0.3", "--time_lim_max", "2.0", "--time_lim_exp", "0.5", "--time_end_lim_min", "0.3", "--time_end_lim_max", "2.0",
                "--time_end_lim_exp", "0.5"]


def _perturb_tables():
    """humanoid3d_walk under --scene imitate with PERTURB_ARGS (the packaged asset carries the arg file's values; the GPU box has no arg files)"""
    from deepmimic_amd import model
    t = model.load_asset("humanoid3d_walk")
    c = t.cfg
    c.enable_rand_perturbs = True; c.perturb_time_min = 0.05; c.perturb_time_max = 0.2; c.min_pertrub_duration = 0.02; c.max_perturb_duration = 0.08
    c.timer_type = "exp"; c.time_lim_min = c.time_end_lim_min = 0.3; c.time_lim_max = c.time_end_lim_max = 2.0; c.time_lim_exp = c.time_end_lim_exp = 0.5
    return t


GOLDEN_SESSIONS.update({
    "target": ("amp_target_zombie", lambda: ["--arg_file", "args/train_amp_target_humanoid3d_zombie_args.txt"], 5, 8, 8, None),
    "perturb": (_perturb_tables, lambda: ["--arg_file", "args/run_humanoid3d_walk_args.txt"] + PERTURB_ARGS, 4242, 8, 10, None),
})
GOLDEN_KINDS = {"heading4": 2, "strike": 4, "dribble": 5, "getup": 3, "target": 1, "perturb": 6}


def golden_tables(asset):
    from deepmimic_amd import model
    return asset() if callable(asset) else model.load_asset(asset)
GOLDEN_POLICY_SCALE = {"getup": 1.0}          # (random actions: the character falls, recovery episodes happen)


def _core_module():
    if COMPAT not in sys.path:
        sys.path.insert(0, COMPAT)
    from DeepMimicCore import DeepMimicCore
    return DeepMimicCore


class RefSession:
    """the compiled reference scene, driven through ref3_* (oracle/ref_standins.cpp)"""
    full = True            # answers reward / goal / state / AMP observation at action boundaries
    per_update = True      # ... and torque, kinematic pose, flags, action latch after every update (live only)

    def __init__(self, ref, args, seed, test_mode=False):
        from deepmimic_amd import model
        p = model.ArgParser(args)
        af = p.str("arg_file", "")
        if af:
            p.load_file(os.path.join(REF, af))
        self.kind = KINDS[p.str("scene", "imitate")]
        toks = []
        for k, v in p.table.items():
            if k not in SKIP_KEYS:
                toks += ["--" + k] + list(v)
        mf = p.str("motion_file", "")
        path = lambda f: f if os.path.isabs(f) else os.path.join(REF, f)
        arr = (C.c_char_p * len(toks))(*[t.encode() for t in toks])
        gids = [int(x) for x in (p.get("getup_motion_ids") or [])]
        garr = (C.c_int * max(1, len(gids)))(*gids) if gids else (C.c_int * 1)(0)
        ref.ref3_open.restype = C.c_void_p
        self.ref = ref
        self.h = C.c_void_p(ref.ref3_open(self.kind, C.c_long(seed), arr, len(toks), path(p.str("character_files", "")).encode(), path(p.str("char_ctrl_files", "")).encode(),
                                         path(mf).encode(), int(p.str("kin_ctrl", "motion") == "clips"), REF.encode(), int(test_mode), garr, len(gids)))
        assert self.h.value, "ref3_open failed"

    def get(self):
        out = np.zeros(48)
        self.ref.ref3_get(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def reset(self):
        return int(self.ref.ref3_reset(self.h))

    def set_char(self, pose, vel, fallen=False):
        a, b = np.ascontiguousarray(pose, dtype=np.float64), np.ascontiguousarray(vel, dtype=np.float64)
        self.ref.ref3_set_char(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), int(fallen))

    def set_ball(self, pos):
        a = np.ascontiguousarray(pos, dtype=np.float64)
        self.ref.ref3_set_ball(self.h, a.ctypes.data_as(C.POINTER(C.c_double)))

    def update(self, dt):
        self.ref.ref3_update(self.h, C.c_double(dt))

    def expert(self, n):
        out = np.zeros(n)
        m = self.ref.ref3_expert(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        assert m == n
        return out

    def set_sample_count(self, n):
        self.ref.ref3_set_sample_count(self.h, int(n))

    def set_mode(self, test):
        self.ref.ref3_set_mode(self.h, int(test))

    def update_kin(self, dt):
        self.ref.ref3_update_kin(self.h, C.c_double(dt))

    def reward_goal(self, ctrl_time, prev_time, prev_com, prev_ball, gdim):
        out = np.zeros(1 + max(gdim, 8)); com = np.ascontiguousarray(prev_com, dtype=np.float64)
        pb = None if prev_ball is None else np.ascontiguousarray(prev_ball, dtype=np.float64)
        n = self.ref.ref3_reward_goal(self.h, C.c_double(ctrl_time), C.c_double(prev_time), com.ctypes.data_as(C.POINTER(C.c_double)),
                                      None if pb is None else pb.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert n == gdim, (n, gdim)
        return float(out[0]), out[1:1 + n].copy()

    def record_state(self, ctrl_time, n):
        out = np.zeros(n)
        assert self.ref.ref3_record_state(self.h, C.c_double(ctrl_time), out.ctypes.data_as(C.POINTER(C.c_double))) == n
        return out

    def amp_agent(self, n):
        out = np.zeros(n)
        assert self.ref.ref3_amp_agent(self.h, out.ctypes.data_as(C.POINTER(C.c_double))) == n
        return out

    def new_action(self):
        self.ref.ref3_new_action(self.h)

    def tables(self, which, nmax=512):
        out = np.zeros(nmax)
        n = self.ref.ref3_tables(self.h, int(which), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out[:n].copy()

    def time_warper(self, build):
        self.ref.ref3_time_warper(self.h, int(build))

    def apply_action(self, action, P):
        a = np.ascontiguousarray(action, dtype=np.float64); tar = np.zeros(P)
        self.ref.ref3_apply_action(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), tar.ctypes.data_as(C.POINTER(C.c_double)))
        return tar

    def spd_tau(self, dt, P):
        out = np.zeros(P)
        self.ref.ref3_spd_tau(self.h, C.c_double(dt), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def set_ball_full(self, s13):
        a = np.ascontiguousarray(s13, dtype=np.float64)
        self.ref.ref3_set_ball_full(self.h, a.ctypes.data_as(C.POINTER(C.c_double)))

    def kin_pose(self, n):
        p, v = np.zeros(n), np.zeros(n)
        assert self.ref.ref3_kin_pose(self.h, p.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double))) == n
        return p

    def flags(self, contact_mask):
        """(CheckTerminate(0), IsEpisodeEnd()) of the compiled scene for the stand-in character's state and these contact flags"""
        self.ref.ref3_set_contacts(self.h, int(contact_mask))
        out = (C.c_int * 2)()
        self.ref.ref3_flags(self.h, out)
        return int(out[0]), bool(out[1])

    def close(self):
        self.ref.ref3_close(self.h)


class Recorder:
    """a RefSession that logs what it answered, in call order (tests/golden/make_ref_draw_golden.py writes the log to tests/golden/ref_draws.npz)"""

    full, per_update = True, False

    def __init__(self, rs):
        self.rs, self.kind, self.h = rs, rs.kind, rs.h
        self.tag, self.rows, self.experts, self.vals = [], [], [], []

    def reward_goal(self, *a):
        r, g = self.rs.reward_goal(*a); self.vals.append(np.concatenate([[r], g])); return r, g

    def record_state(self, *a):
        v = self.rs.record_state(*a); self.vals.append(v); return v

    def amp_agent(self, *a):
        v = self.rs.amp_agent(*a); self.vals.append(v); return v

    def reset(self):
        rec = self.rs.reset(); self.tag.append(10 + rec); self.rows.append(self.rs.get()); return rec

    def update(self, dt):
        self.rs.update(dt); self.tag.append(1); self.rows.append(self.rs.get())

    def get(self):
        return self.rows[-1]

    def flags(self, contact_mask):
        return None                     # (not part of the committed log: the GPU box checks the draws)

    def expert(self, n):
        v = self.rs.expert(n); self.tag.append(2); self.rows.append(np.zeros(48)); self.experts.append(v); return v

    def __getattr__(self, name):
        return getattr(self.rs, name)

    def save(self, path, key, store):
        store[key + "_tag"] = np.array(self.tag, dtype=np.int32); store[key + "_rows"] = np.array(self.rows)
        store[key + "_experts"] = np.array(self.experts) if self.experts else np.zeros((0, 1))
        store[key + "_vals"] = np.concatenate(self.vals) if self.vals else np.zeros(0)          # reward | goal, state, AMP observation of every action boundary, in call order
        store[key + "_val_ends"] = np.cumsum([v.size for v in self.vals]).astype(np.int64) if self.vals else np.zeros(0, np.int64)


class Replay:
    """the log of a Recorder standing in for the compiled reference where the reference checkout does not exist (the GPU box)"""

    per_update = False

    def __init__(self, store, key, kind, values=True):
        self.kind, self.h = kind, None
        self.tag, self.rows, self.experts = store[key + "_tag"], store[key + "_rows"], store[key + "_experts"]
        self.i = -1; self.ie = 0; self.iv = 0
        self.full = bool(values and (key + "_vals") in store and store[key + "_vals"].size)
        if self.full:
            self.vals, self.val_ends = store[key + "_vals"], store[key + "_val_ends"]

    def _val(self, n):
        a = 0 if self.iv == 0 else int(self.val_ends[self.iv - 1]); b = int(self.val_ends[self.iv]); self.iv += 1
        assert b - a == n, "the run left the recorded call sequence (value of %d entries, %d logged)" % (n, b - a)
        return self.vals[a:b]

    def reward_goal(self, ctrl_time, prev_time, prev_com, prev_ball, gdim):
        v = self._val(1 + gdim); return float(v[0]), v[1:]

    def record_state(self, ctrl_time, n):
        return self._val(n)

    def amp_agent(self, n):
        return self._val(n)

    def set_ball_full(self, *a): pass
    def new_action(self): pass
    def time_warper(self, build): pass
    def apply_action(self, *a): return None

    def _next(self, want):
        self.i += 1
        assert self.i < len(self.tag) and (self.tag[self.i] == want or (want == 10 and self.tag[self.i] in (10, 11))), "the run left the recorded call sequence"

    def reset(self):
        self._next(10); return int(self.tag[self.i] - 10)

    def update(self, dt):
        self._next(1)

    def get(self):
        return self.rows[self.i]

    def expert(self, n):
        self._next(2); v = self.experts[self.ie]; self.ie += 1; return v

    def set_char(self, *a, **k): pass
    def update_kin(self, dt): pass
    def flags(self, contact_mask): return None
    def set_ball(self, *a): pass
    def set_sample_count(self, n): pass
    def set_mode(self, test): pass
    def close(self): pass


def _facade(mod, lib, args, seed, monkeypatch, test_mode=False, precision="64", tables=None):
    from deepmimic_amd import model
    monkeypatch.setenv("DM_HIP_LIB", lib); monkeypatch.setenv("DM_PRECISION", precision); monkeypatch.delenv("DM_RNG", raising=False)
    monkeypatch.setenv("DM_FACADE_BATCH", "0")           # one launch per Update: the test looks at the scene after every update
    monkeypatch.setenv("DM_DATA_ROOT", REF)
    core = mod.cDeepMimicCore(False)
    core.SeedRand(seed)
    t = tables if tables is not None else model.load_scene_from_args(list(args), data_root=REF)
    core.LoadTables(t, 10)
    core.Init()
    assert core._tape or not (t.goal_kind or t.num_clips > 1 or t.cfg.enable_rand_rot_reset or t.cfg.enable_rand_perturbs), "the facade did not choose the draw tape for this scene"
    return core, t


def _dev(core):
    """what the device holds after the last launch, in the layout of ref3_get"""
    env = core._env
    st = env.get_state()
    out = {"limit": float(st["clocks"][0][4]), "kin_time": float(st["clocks"][0][0]), "clip": int(env.get_clips()[0]), "kin_rot": np.array(st["kin"][0][3:7], dtype=np.float64), "kin_pos": np.array(st["kin"][0][0:3], dtype=np.float64),
           "pose": np.array(st["pose"][0], dtype=np.float64), "vel": np.array(st["vel"][0], dtype=np.float64), "contacts": int(st["flags"][0][1]), "ctrl_time": float(st["clocks"][0][1])}
    if env._has_goal_row and core._tables.goal_kind:
        g = env.get_goal_state()[0]; aux = env.get_goal_aux()[0]
        out.update(target=g[0:3].copy(), heading=float(g[3]), speed=float(g[4]), ttimer=float(g[5]), ttimer_max=float(g[6]), aux=aux.copy())
    if env.has_perturbs:
        out["pert"] = env.get_perturb_state()[0].copy()
    if env.has_obj:
        out["ball"] = env.get_obj_state()[0].copy()
    return out


def _check(kind, d, r, where, pos_tol=1e-9, after_reset=False, exact=True, live_kin=False):
    if not exact:
        return _check_close(kind, d, r, where, after_reset)
    assert d["limit"] == r[0], (where, "episode limit", d["limit"], r[0])
    if after_reset:
        assert d["clip"] == int(r[2]), (where, "clip", d["clip"], r[2])
        assert d["kin_time"] == r[1], (where, "clip time", d["kin_time"], r[1])
        # the yaw: kin origin rotation about +y (cKinCharacter::RotateOrigin) -- equal as rotations (the device stores its own quaternion of the same angle)
        assert abs(abs(float(np.dot(d["kin_rot"], r[3:7]))) - 1.0) < 1e-12, (where, "yaw", d["kin_rot"], r[3:7])
    if not after_reset and len(r) > 42 and live_kin:
        # cSceneImitate::UpdateKinChar as compiled, on the device's character: the kinematic clock and -- through SyncKinCharNewCycle at every cycle boundary -- its origin
        assert abs(d["kin_time"] - r[1]) < 1e-9, (where, "kin time", d["kin_time"], r[1])
        assert abs(abs(float(np.dot(d["kin_rot"], r[3:7]))) - 1.0) < 1e-9, (where, "kin origin rotation", d["kin_rot"], r[3:7])
    if 1 <= kind <= 5:
        assert d["ttimer_max"] == r[12], (where, "target timer limit", d["ttimer_max"], r[12])
        assert d["speed"] == r[11], (where, "target speed", d["speed"], r[11])
        if kind in (2, 3):
            assert d["heading"] == r[10], (where, "target heading", d["heading"], r[10])
        assert np.abs(d["target"] - r[7:10]).max() < pos_tol, (where, "target position", d["target"], r[7:10])
    if kind == 4:
        assert bool(d["aux"][0]) == bool(r[21]), (where, "target hit", d["aux"][0], r[21])
        if r[21]:
            assert d["aux"][1] == r[22], (where, "hit time", d["aux"][1], r[22])
    if kind == 5:
        assert d["aux"][6] == r[30], (where, "object timer limit", d["aux"][6], r[30])
        assert np.abs(d["ball"][0:3] - r[23:26]).max() < pos_tol, (where, "ball position", d["ball"][0:3], r[23:26])
        assert abs(abs(float(np.dot(d["ball"][3:7], r[26:30]))) - 1.0) < 1e-12, (where, "ball rotation", d["ball"][3:7], r[26:30])
    if "pert" in d:
        assert d["pert"][1] == r[14], (where, "next perturbation time", d["pert"][1], r[14])


def _check_close(kind, d, r, where, after_reset):
    """fp32 kernels: the same draws through float parameters and a float character state"""
    near = lambda a, b, tol=2e-6: abs(a - b) <= tol * max(1.0, abs(b))
    assert near(d["limit"], r[0]), (where, "episode limit", d["limit"], r[0])
    if after_reset:
        assert d["clip"] == int(r[2]) and near(d["kin_time"], r[1]), (where, "clip / clip time", d["clip"], d["kin_time"], r[1:3])
        assert abs(abs(float(np.dot(d["kin_rot"], r[3:7]))) - 1.0) < 1e-6, (where, "yaw")
    if 1 <= kind <= 5:
        assert near(d["ttimer_max"], r[12]) and near(d["speed"], r[11]), (where, "target timer / speed", d["ttimer_max"], d["speed"], r[11:13])
        if kind in (2, 3):
            assert near(d["heading"], r[10]), (where, "target heading", d["heading"], r[10])
    if kind == 4:
        assert bool(d["aux"][0]) == bool(r[21]), (where, "target hit")
    if "pert" in d:
        assert near(d["pert"][1], r[14]), (where, "next perturbation time", d["pert"][1], r[14])


def _run(mod, lib, args, seed, monkeypatch, n_resets, steps, precision="64", pos_tol=1e-9, anneal_at=None, policy_scale=0.0, tables=None, provider=None, exact=True, test_mode=False, val_tol=1.0):
    """provider: None = the compiled reference, live; a Recorder (logs it) or a Replay (a committed log).  exact = False (fp32 kernels): the scene parameters
    are floats there, values that are a parameter times a draw agree to float accuracy"""
    core, t = _facade(mod, lib, args, seed, monkeypatch, precision=precision, tables=tables)
    rs = provider if provider is not None else RefSession(ref_lib.load("ref"), args, seed)
    kind = rs.kind
    env = core._env
    if test_mode:                      # the learner switches modes after Init (learning/rl_world.py): the scene's Init drew in train mode
        core.SetMode(core.eModeTest); rs.set_mode(1)
    if rs.per_update:
        # the learner-facing tables of the scene class (cRLScene's virtuals as each scene answers them: offsets, scales, normalisation groups, action bounds, reward range)
        dev_tables = [core.BuildStateOffset(0), core.BuildStateScale(0), core.BuildStateNormGroups(0), core.BuildGoalOffset(0), core.BuildGoalScale(0), core.BuildGoalNormGroups(0),
                      core.BuildActionOffset(0), core.BuildActionScale(0), core.BuildActionBoundMin(0), core.BuildActionBoundMax(0)]
        if kind <= 5:
            dev_tables += [core.GetAMPObsOffset(), core.GetAMPObsScale(), core.GetAMPObsNormGroup()]
        else:
            dev_tables += [[], [], []]
        dev_tables.append([core.GetRewardMin(0), core.GetRewardMax(0), core.GetRewardFail(0), core.GetRewardSucc(0)])
        for which, dv in enumerate(dev_tables):
            rv = rs.tables(which)
            dv = np.asarray(dv, dtype=np.float64)
            assert dv.shape == rv.shape and (dv.size == 0 or np.abs(dv - rv).max() <= 1e-12 * max(1.0, np.abs(rv).max())), ("scene table %d" % which, dv[:8], rv[:8])
    warp = kind == 0 and test_mode and rs.h is not None          # imitate_amp's test-mode return: cSceneImitateAMP::CalcRewardTimeWarp
    if warp:
        rs.time_warper(1)
    dt = 1.0 / 600
    rng = np.random.RandomState(seed & 0xffff)
    n_pert = 0; n_rec = 0; n_rew = 0; n_amp = 0; n_tau = 0; n_warp = 0; samplers = {}
    live = rs.per_update
    import parity_common as pc
    dof_idx = pc.dof_index(t)
    from deepmimic_amd import model
    fall_bits = int(sum(1 << j for j, f in enumerate(t.fall_mask()) if f)) if t.cfg.enable_char_contact_fall else 0      # (cSimCharacter::BuildBodyLinks: no part gets the flag without enable_char_contact_fall)
    try:
        for ep in range(n_resets):
            if anneal_at and ep in anneal_at:
                core.SetSampleCount(anneal_at[ep]); rs.set_sample_count(anneal_at[ep])
            # the stand-in character of the reference side holds the device's state of the moment (the recovery decision and the ball's reset read it)
            d = _dev(core)
            q = core._query()
            rs.set_char(d["pose"], d["vel"], fallen=bool(q["terminate"][0] == 1))
            if kind == 5:
                rs.set_ball(d["ball"][0:3])
            core.Reset()
            rec = rs.reset(); n_rec += rec
            pose_before = d["pose"]
            d = _dev(core)
            assert bool(rec) == bool(np.array_equal(pose_before, d["pose"])), ("reset %d" % ep, "recovery episode (the character stays where it fell)", rec)
            _check(kind, d, rs.get(), "reset %d" % ep, pos_tol, after_reset=not rec, exact=exact)
            if kind == 0 and rs.h is not None:
                # imitate_amp re-initialises the pose history at Reset from the kinematic character one control period back: its origin height carries the reset's
                # ground-intersection lift, which is Bullet-side here -- taken from the device, then cSceneImitateAMP::InitHist runs again
                rs.ref.ref3_set_kin_origin_pos(rs.h, np.ascontiguousarray(d["kin_pos"]).ctypes.data_as(C.POINTER(C.c_double)))
                rs.ref.ref3_init_hist(rs.h)
                if warp:                          # cSceneImitateAMP::Reset -> ResetTimeWarper: first samples of both characters, where the device's are
                    rs.set_char(d["pose"], d["vel"]); rs.time_warper(0)
            for k in range(steps * 20):
                if core.NeedNewAction(0):
                    s_dev = np.array(core.RecordState(0)); g_dev = np.array(core.RecordGoal(0)); r_dev = core.CalcReward(0)
                    if rs.full and kind >= 1:
                        # the scene's own CalcReward / RecordGoal on the device's character, with the session's target / heading / speed / hit state
                        d0 = _dev(core); gs = env.get_goal_state()[0] if env._has_goal_row else np.zeros(12)
                        rs.set_char(d0["pose"], d0["vel"], fallen=bool(d0["contacts"] & fall_bits))
                        if kind == 5:
                            rs.set_ball_full(d0["ball"][:13])
                        r_ref, g_ref = rs.reward_goal(d0["ctrl_time"], float(gs[10]), gs[7:10], d0["aux"][2:5] if kind == 5 else None, g_dev.size)
                        assert (g_dev.size == 0 or np.abs(g_dev - g_ref).max() < 2e-6 * val_tol) and abs(r_dev - r_ref) < 2e-6 * val_tol, ("episode %d update %d" % (ep, k), "goal / reward", g_dev, g_ref, r_dev, r_ref)
                        # cCtController::RecordState of the reference's own controller on the device's character (phase from the controller clock, ground height, link frames)
                        s_ref = rs.record_state(d0["ctrl_time"], s_dev.size)
                        assert np.abs(s_dev - s_ref).max() < 2e-6 * val_tol * max(1.0, np.abs(s_ref).max()), ("episode %d update %d" % (ep, k), "state", int(np.argmax(np.abs(s_dev - s_ref))), np.abs(s_dev - s_ref).max())
                        n_rew += 1
                    if rs.full and kind <= 5:
                        # RecordAMPObsAgent: the pose latched at the previous action boundary (the scene's own NewActionUpdate) and the pose now
                        if kind == 0:
                            d0 = _dev(core); rs.set_char(d0["pose"], d0["vel"])
                        if k > 0 or kind == 0:      # (the task scenes do not re-initialise the history at Reset: the first pair of an episode is stale in the reference)
                            a_dev = np.array(core.RecordAMPObsAgent(0)); a_ref = rs.amp_agent(a_dev.size)
                            assert np.abs(a_dev - a_ref).max() < 5e-6 * val_tol * max(1.0, np.abs(a_ref).max()), ("episode %d update %d" % (ep, k), "AMP observation", int(np.argmax(np.abs(a_dev - a_ref))), np.abs(a_dev - a_ref).max())
                            n_amp += 1
                        rs.new_action()
                    act = (policy_scale * rng.randn(env.A)).astype(np.float32)
                    core.SetAction(0, act)
                    if live:                      # cCtPDController::ApplyAction of the reference's controller: the PD targets of this control step
                        tar_ref = rs.apply_action(act, env.P)
                if live:                          # the torque the reference's controller computes for the state this update starts from (the stand-in holds it)
                    tau_ref = pc.clamp_tau(t, rs.spd_tau(dt, env.P))[dof_idx]
                core.Update(dt)
                d = _dev(core)
                if live and hasattr(env, "debug"):          # (the debug taps are a private context's; a slot of the shared owner has none)
                    tau_dev = env.debug("tau")[0]
                    assert np.abs(tau_dev - tau_ref).max() < 1e-7 * max(1.0, np.abs(tau_ref).max()), ("episode %d update %d" % (ep, k), "stable-PD torque", int(np.argmax(np.abs(tau_dev - tau_ref))), np.abs(tau_dev - tau_ref).max())
                    n_tau += 1
                rs.update_kin(dt)                 # (cSceneImitate::UpdateCharacters: before the world steps)
                cmask = d["contacts"]
                rs.set_char(d["pose"], d["vel"], fallen=bool(cmask & fall_bits))      # cSimCharacter::HasFallen: a fall-contact body touches something
                if kind == 5:                     # position, rotation and velocities: the ball is a rigid body of the device's simulation, the scene only ever re-places it
                    rs.set_ball_full(d["ball"][:13])
                rs.update(dt)
                r = rs.get()
                if rs.per_update:                 # the kinematic character itself: the device's (clip, clip time, origin) through the host sampler vs cKinCharacter::GetPose as compiled
                    clip = d["clip"]
                    if clip not in samplers:
                        samplers[clip] = model.KinSampler(t, clip)
                    kp = samplers[clip].pose(d["kin_time"], d["kin_pos"], d["kin_rot"])
                    rp = rs.kin_pose(kp.size)
                    err = np.abs(kp - rp); err[1] = 0.0      # (root height: the reset's ground-intersection lift is Bullet-side; see draw_reset_scene)
                    for o in samplers[clip].quat_offs:       # q and -q are one rotation
                        if np.dot(kp[o:o + 4], rp[o:o + 4]) < 0:
                            err[o:o + 4] = np.abs(kp[o:o + 4] + rp[o:o + 4])
                    assert err.max() < 1e-7, ("episode %d update %d" % (ep, k), "kin pose", int(np.argmax(err)), err.max(), kp[:7], rp[:7])
                if live:                          # the 30 Hz action latch: cCtController::CheckNeedNewAction at the controller clock this update ended on
                    assert bool(rs.ref.ref3_need_new_action(rs.h, C.c_double(d["ctrl_time"]), C.c_double(dt))) == bool(core.NeedNewAction(0)), ("episode %d update %d" % (ep, k), "NeedNewAction")
                fl = rs.flags(cmask)
                if fl is not None:                # the scene's own CheckTerminate / IsEpisodeEnd on that state: the task scenes' success / failure rules, the clocks
                    assert (core.CheckTerminate(0), core.IsEpisodeEnd()) == fl, ("episode %d update %d" % (ep, k), "terminate / episode end", core.CheckTerminate(0), core.IsEpisodeEnd(), fl)
                _check(kind, d, r, "episode %d update %d" % (ep, k), pos_tol, exact=exact, live_kin=rs.per_update)
                if "pert" in d and int(r[31]) > n_pert:          # a perturbation fell due in this update: part, force, duration
                    n_pert = int(r[31])
                    slots = d["pert"][3:15].reshape(2, 6)
                    hit = [s for s in slots if int(s[0]) - 1 == int(r[16]) and s[4] == r[20]]
                    assert hit and np.abs(hit[0][1:4] - r[17:20]).max() < 1e-9 * max(1.0, np.abs(r[17:20]).max()), ("perturbation", slots, r[16:21])
                if core.IsEpisodeEnd() or not core.CheckValidEpisode():
                    if warp and core.IsEpisodeEnd():      # the evaluation return of the episode: alignment cost of the two sample series + the steps it fell short
                        w_dev = core.CalcReward(0); w_ref = rs.reward_goal(d["ctrl_time"], 0.0, np.zeros(3), None, 0)[0]
                        assert abs(w_dev - w_ref) < 1e-6 * max(1.0, abs(w_ref)), ("episode %d" % ep, "time-warp return", w_dev, w_ref)
                        n_warp += 1
                    break
            if core._is_amp():                                    # RecordAMPObsExpert: clip (gRand) and clip time (mRand) between episodes
                if rs.h is not None:
                    ko = np.ascontiguousarray(env.get_state()["kin"][0][0:3], dtype=np.float64)      # (ground height of the sample = the kin origin's)
                    rs.ref.ref3_set_kin_origin_pos(rs.h, ko.ctypes.data_as(C.POINTER(C.c_double)))
                a = np.array(core.RecordAMPObsExpert(0)); b = rs.expert(a.size)
                assert np.abs(a - b).max() < (1e-4 if exact else 2e-3), ("expert sample", np.abs(a - b).max())
    finally:
        rs.close()
        kind_of_env = type(env).__name__
        if hasattr(env, "close") and kind_of_env == "SharedEnv":      # leave the owner's slot now (the owner goes when its last worker has left)
            env.close()
    return {"perturbations": n_pert, "recoveries": n_rec, "rewards": n_rew, "amp_obs": n_amp, "torques": n_tau, "time_warp": n_warp, "env": kind_of_env}


@pytest.mark.parametrize("name", ["heading4", "dribble", "perturb", "getup", "strike", "target"])
def test_live_session_behind_the_shared_owner(emu_lib, monkeypatch, name):
    """the drop-in behind `DM_FACADE_SHARED=1` (deepmimic_amd/broker.py: the worker is a slot of the owner process's context, its draw tape travels with its
    requests) against the compiled scenes directly -- not only through "shared = private" (tests/test_broker.py) and "private = reference" (the tests below)"""
    import glob
    import time
    asset, args, seed, n_resets, steps, anneal = GOLDEN_SESSIONS[name]
    shm = "dm_live_%s_%d" % (name, os.getpid())
    monkeypatch.setenv("DM_FACADE_SHARED", "1"); monkeypatch.setenv("DM_FACADE_SHM", shm); monkeypatch.setenv("DM_FACADE_SHARED_MAX", "4")
    try:
        out = _run(_core_module(), emu_lib, args(), seed, monkeypatch, n_resets=4, steps=steps, pos_tol=1e-6 if name == "dribble" else 1e-9, anneal_at=anneal,
                   policy_scale=GOLDEN_POLICY_SCALE.get(name, 0.0))
        assert out["env"] == "SharedEnv" and out["rewards"] >= 20, out
        assert name != "perturb" or out["perturbations"] >= 4, out
    finally:
        t_end = time.monotonic() + 45
        while os.path.exists("/dev/shm/" + shm) and time.monotonic() < t_end:
            time.sleep(0.2)
        gone = not os.path.exists("/dev/shm/" + shm)
        for f in glob.glob("/dev/shm/%s.*" % shm):
            os.unlink(f)
    assert gone, "the owner process did not leave"


@pytest.mark.parametrize("name", ["heading4", "perturb", "dribble"])
def test_live_session_on_the_v2_kernels(emu_lib, monkeypatch, name):
    """`DM_PHYSICS=2`: the draw tape and the DeepMimic-side arithmetic in the V2 instantiations of the kernels (another kernel family each; the ball under v2 is family 23)"""
    asset, args, seed, n_resets, steps, anneal = GOLDEN_SESSIONS[name]
    monkeypatch.setenv("DM_PHYSICS", "2")
    out = _run(_core_module(), emu_lib, args(), seed, monkeypatch, n_resets=4, steps=steps, anneal_at=anneal, pos_tol=1e-6 if name == "dribble" else 1e-9)
    assert out["rewards"] >= 20 and out["torques"] >= 400, out


def test_heading_amp_four_clips(emu_lib, monkeypatch):
    """heading_amp over a 4-clip dataset, 20 resets x 10 control steps: clip by weight, clip time over the PREVIOUS clip's duration, yaw, target timer,
    sharp / Gaussian heading steps, speed changes"""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt", "--scene", "heading_amp"]      # (deepmimic_amd/assets/amp_heading_clips4)
    _run(mod, emu_lib, args, 20231, monkeypatch, n_resets=20, steps=10, anneal_at={5: 32000000, 12: 64000000})


def test_strike_amp(emu_lib, monkeypatch):
    """strike_amp (two-clip dataset): far / near coin on the scene generator, the three target coordinates and the initial-hit coin on cMathUtil's, the hit
    time back on the scene's"""
    mod = _core_module()
    ds = os.path.join(ROOT, "tools", "datasets", "humanoid3d_clips_walk_punch_local.txt")      # (the shipped dataset names clips that are not in the repository)
    args = ["--arg_file", "args/train_amp_strike_humanoid3d_walk_punch_args.txt", "--motion_file", ds, "--init_hit_prob", "0.3"]
    out = _run(mod, emu_lib, args, 77, monkeypatch, n_resets=14, steps=10)
    assert out["rewards"] >= 70 and out["amp_obs"] >= 55


def test_target_amp(emu_lib, monkeypatch):
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_target_humanoid3d_zombie_args.txt", "--rand_target_time_min", "0.05", "--rand_target_time_max", "0.3"]
    assert _run(mod, emu_lib, args, 5, monkeypatch, n_resets=5, steps=12)["rewards"] >= 30


def test_dribble_amp(emu_lib, monkeypatch):
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_dribble_humanoid3d_zombie_args.txt", "--rand_target_time_min", "0.05", "--rand_target_time_max", "0.3",
            "--rand_tar_obj_time_min", "0.1", "--rand_tar_obj_time_max", "0.4"]
    assert _run(mod, emu_lib, args, 12345, monkeypatch, n_resets=5, steps=12, pos_tol=1e-6)["rewards"] >= 30


def test_imitate_amp_perturbations_exp_timer(emu_lib, monkeypatch):
    """imitate_amp with random perturbations (part, direction, magnitude, duration, next time: all on the scene generator), an exponential episode timer with a
    real range (cMathUtil's generator, through -log(1 - u)) and a random yaw"""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_humanoid3d_run_args.txt", "--enable_rand_perturbs", "true", "--perturb_time_min", "0.05", "--perturb_time_max", "0.2",
            "--min_pertrub_duration", "0.02", "--max_perturb_duration", "0.08", "--timer_type", "exp", "--time_lim_min", "0.3", "--time_lim_max", "2.0", "--time_lim_exp", "0.5",
            "--time_end_lim_min", "0.3", "--time_end_lim_max", "2.0", "--time_end_lim_exp", "0.5", "--enable_rand_rot_reset", "true"]
    out = _run(mod, emu_lib, args, 4242, monkeypatch, n_resets=6, steps=12)
    n = out["perturbations"]
    assert out["amp_obs"] >= 30          # (imitate_amp: the first pair of an episode included -- cSceneImitateAMP::Reset re-initialises the history)
    assert n >= 6, "no perturbation fell due: the test would not see their draws"


def test_heading_amp_getup_recovery_episodes(emu_lib, monkeypatch):
    """heading_amp_getup in train mode: an episode that ended in a fall continues as a recovery episode when the scene generator's coin says so
    (cSceneHeadingAMPGetup::ActivateRecoveryEpisode) -- one timer reset instead of the scene reset's four, no character reset"""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt", "--time_lim_min", "0.2", "--time_lim_max", "3.0", "--time_end_lim_min", "0.2",
            "--time_end_lim_max", "3.0", "--recover_episode_prob", "0.5"]
    n = _run(mod, emu_lib, args, 99, monkeypatch, n_resets=9, steps=30, policy_scale=1.0)["recoveries"]
    assert n >= 2, "no recovery episode happened: the test would not see its draws"


def test_batched_control_steps_draw_the_same(emu_lib, monkeypatch):
    """The default route consumes a control step in one launch (and rolls back when the caller looks inside it); the generators travel with the snapshot, so the
    draws are the ones of the update-by-update route."""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt", "--scene", "heading_amp", "--enable_rand_perturbs", "true",
            "--perturb_time_min", "0.05", "--perturb_time_max", "0.2", "--min_pertrub_duration", "0.02", "--max_perturb_duration", "0.08"]
    logs = []
    for batch in ("0", "1"):
        core, t = _facade(mod, emu_lib, args, 31337, monkeypatch)
        monkeypatch.setenv("DM_FACADE_BATCH", batch)
        core = mod.cDeepMimicCore(False); core.SeedRand(31337); core.LoadTables(t, 10); core.Init()
        core.SetSampleCount(64000000)
        log = []
        for ep in range(4):
            core.Reset()
            for k in range(12 * 20):
                if core.NeedNewAction(0):
                    log.append((core.RecordState(0), core.RecordGoal(0), core._env.get_goal_state()[0].copy(), core._env.get_perturb_state()[0].copy()))
                    core.SetAction(0, np.zeros(core._env.A))
                core.Update(1.0 / 600)
                if ep == 1 and k % 20 == 7:
                    log.append(core.RecordGoal(0))        # a look inside a control step: rollback + replay on the batched route
                if core.IsEpisodeEnd():
                    break
        logs.append(log)
        if batch == "1":
            assert core.stats["rollbacks"] > 0 and core.stats["launches"] < core.stats["updates"]
    assert len(logs[0]) == len(logs[1])
    for a, b in zip(logs[0], logs[1]):
        if isinstance(a, tuple):
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x), np.asarray(y))
        else:
            assert np.array_equal(np.asarray(a), np.asarray(b))


def test_strike_amp_test_mode(emu_lib, monkeypatch):
    """test mode: the episode limit is drawn with the timer's parameters and then pinned (cRLSceneSimChar::ResetTimers), no initial-hit coin"""
    mod = _core_module()
    ds = os.path.join(ROOT, "tools", "datasets", "humanoid3d_clips_walk_punch_local.txt")
    args = ["--arg_file", "args/train_amp_strike_humanoid3d_walk_punch_args.txt", "--motion_file", ds, "--init_hit_prob", "0.3", "--time_lim_min", "0.2", "--time_lim_max", "0.6",
            "--time_end_lim_min", "0.3", "--time_end_lim_max", "0.7"]
    _run(mod, emu_lib, args, 4711, monkeypatch, n_resets=5, steps=14, test_mode=True)


def test_perturbations_only(emu_lib, monkeypatch):
    """a scene whose only device-side draws are the perturbations (no goal row on the device): clip time and episode limit are drawn on the host in the
    reference's order, the perturbation clock and forces come off the tape"""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_humanoid3d_run_args.txt", "--enable_rand_perturbs", "true", "--perturb_time_min", "0.05", "--perturb_time_max", "0.2",
            "--min_pertrub_duration", "0.02", "--max_perturb_duration", "0.08", "--enable_rand_rot_reset", "false", "--time_lim_min", "0.3", "--time_lim_max", "0.9",
            "--time_end_lim_min", "0.3", "--time_end_lim_max", "0.9", "--perturb_part_ids", "1", "2", "6", "9"]
    core, _ = _facade(mod, emu_lib, args, 808, monkeypatch)
    assert not core._env._has_goal_row
    n = _run(mod, emu_lib, args, 808, monkeypatch, n_resets=5, steps=12)["perturbations"]
    assert n >= 5


def test_more_draws_in_a_control_step_than_a_tape_holds(emu_lib, monkeypatch):
    """ball and target re-sampled at every update: 14 raw values of the scene generator per update, 280 per control step -- a batched control step runs past the
    tape (96 per launch); the facade then takes that control step update by update: same draws as the update-by-update route (which the compiled reference
    confirms, test_dribble_amp)"""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_dribble_humanoid3d_zombie_args.txt", "--rand_target_time_min", "0.001", "--rand_target_time_max", "0.0015",
            "--rand_tar_obj_time_min", "0.001", "--rand_tar_obj_time_max", "0.0015"]
    logs = []
    for batch in ("0", "1"):
        core, t = _facade(mod, emu_lib, args, 17, monkeypatch)
        monkeypatch.setenv("DM_FACADE_BATCH", batch)
        core = mod.cDeepMimicCore(False); core.SeedRand(17); core.LoadTables(t, 10); core.Init()
        log = []
        for ep in range(2):
            core.Reset()
            for k in range(4 * 20):
                if core.NeedNewAction(0):
                    log.append((core.RecordGoal(0), core._env.get_goal_state()[0].copy(), core._env.get_obj_state()[0].copy())); core.SetAction(0, np.zeros(core._env.A))
                core.Update(1.0 / 600)
        logs.append(log)
        assert (core.stats.get("tape_fallbacks", 0) > 0) == (batch == "1")
    assert len(logs[0]) == len(logs[1]) >= 8
    for a, b in zip(*logs):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("arg_file,steps,resets", [("args/run_humanoid3d_walk_args.txt", 30, 2), ("args/train_humanoid3d_spinkick_args.txt", 30, 3), ("args/train_dog3d_pace_args.txt", 8, 2),
                                                   ("args/run_dog3d_spin_args.txt", 24, 1),                        # sync_char_root_rot: the kinematic origin turns with the simulated heading
                                                   ("args/train_humanoid3d_roll_args.txt", 20, 2),                 # enable_char_contact_fall false: rolling on the ground is no fall
                                                   ("args/train_humanoid3d_getup_facedown_args.txt", 20, 2),       # a non-looping clip: the end of the motion fails the episode
                                                   ("args/train_humanoid3d_backflip_args.txt", 20, 2)])
def test_imitate_scenes_live(emu_lib, monkeypatch, arg_file, steps, resets):
    """`--scene imitate` (the headline scene): the compiled cSceneImitate / cKinCharacter / cCtPDController on the device's character after every update -- kinematic
    pose through the cycle boundaries (SyncKinCharNewCycle), the stable-PD torque of every update (cCtPDController::ApplyAction -> cImpPDController on cRBDModel, clamped
    per joint), CalcRewardImitate and RecordState at every action boundary, CheckTerminate / IsEpisodeEnd (fall,
    end of a non-looping motion, episode timer), reset clip times and limits from the reference's generator"""
    mod = _core_module()
    out = _run(mod, emu_lib, ["--arg_file", arg_file], 31, monkeypatch, n_resets=resets, steps=steps)
    assert out["rewards"] >= 5 * resets and out["torques"] >= 80 * resets


def _shipped_arg_files():
    import glob
    files = sorted(os.path.relpath(f, REF) for f in glob.glob(os.path.join(REF, "args", "*.txt")))
    # no agent (the viewer's kin_char scenes) / motion files that are not in the reference's repository
    files = [f for f in files if "play_motion" not in f and "_locomotion_args" not in f and "walk_punch" not in f]
    return files[::9] if os.environ.get("DM_LIVE_SWEEP") == "sample" else files


@pytest.mark.parametrize("arg_file", _shipped_arg_files())
def test_every_shipped_arg_file_live(emu_lib, monkeypatch, arg_file):
    """a short live session (2 resets x 3 control steps, every update checked) straight from the reference's own arg file: its parameters reach the compiled scene
    through the scene's own ParseArgs and the device through this repo's loader.  All 88 files that have their data (45 s of CPU; every ninth with
    DM_LIVE_SWEEP=sample; profiles/r05_live_sessions_every_arg_file.txt)"""
    ev = lambda k, d: int(os.environ.get(k, d))          # a longer sweep off-line: DM_LIVE_SEED / DM_LIVE_RESETS / DM_LIVE_STEPS / DM_LIVE_POLICY (action noise) / DM_LIVE_TEST (test mode)
    out = _run(_core_module(), emu_lib, ["--arg_file", arg_file], ev("DM_LIVE_SEED", 7), monkeypatch, n_resets=ev("DM_LIVE_RESETS", 2), steps=ev("DM_LIVE_STEPS", 3),
               policy_scale=float(os.environ.get("DM_LIVE_POLICY", 0.0)), test_mode=bool(ev("DM_LIVE_TEST", 0)))
    assert out["torques"] >= 60


def test_imitate_amp_time_warp_return_live(emu_lib, monkeypatch):
    """imitate_amp in test mode: the return of an episode is the dynamic-time-warping cost between the simulated and the kinematic character's joint positions sampled
    at every action boundary (cSceneImitateAMP::CalcRewardTimeWarp on cDynamicTimeWarper) -- computed on the host by the drop-in, by the compiled scene here"""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_humanoid3d_run_args.txt", "--time_lim_min", "0.4", "--time_lim_max", "0.8", "--time_end_lim_min", "0.4", "--time_end_lim_max", "0.8"]
    out = _run(mod, emu_lib, args, 99, monkeypatch, n_resets=4, steps=30, test_mode=True)
    assert out["time_warp"] >= 3


def test_heading_amp_getup_test_mode(emu_lib, monkeypatch):
    """heading_amp_getup in test mode: a fall does not end the episode, it starts a get-up (cSceneHeadingAMPGetup::UpdateTestGetup); the get-up phase rides in the goal,
    the get-up reward replaces the task reward while it lasts"""
    mod = _core_module()
    args = ["--arg_file", "args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt", "--time_lim_min", "0.5", "--time_lim_max", "1.5", "--time_end_lim_min", "0.5",
            "--time_end_lim_max", "1.5"]
    out = _run(mod, emu_lib, args, 123, monkeypatch, n_resets=2, steps=40, policy_scale=1.0, test_mode=True)
    assert out["rewards"] >= 35


@pytest.mark.parametrize("arg_file", ["args/train_amp_target_humanoid3d_zombie_args.txt", "args/train_amp_dribble_humanoid3d_zombie_args.txt", "args/train_amp_heading_humanoid3d_zombie_args.txt"])
def test_task_scenes_test_mode(emu_lib, monkeypatch, arg_file):
    """the task scenes in test mode (pinned episode limit; the rewards' test-mode branches)"""
    out = _run(_core_module(), emu_lib, ["--arg_file", arg_file, "--time_end_lim_min", "0.6", "--time_end_lim_max", "0.6"], 77, monkeypatch, n_resets=3, steps=20, test_mode=True,
               pos_tol=1e-6)
    assert out["rewards"] >= 30
