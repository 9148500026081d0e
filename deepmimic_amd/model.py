"""Scene description loader: DeepMimic JSON / arg files -> raw numeric tables.

This module only *parses*.  It turns the on-disk formats the hot path consumes
(character skeleton + body defs, PD-controller file, motion clip, arg file)
into plain numpy tables laid out exactly like the reference's in-memory
matrices, so that the native host library (csrc/dm_host.cpp) and the test
oracle can each derive everything else on their own:

* ``joint_mat``  [J x 19]  columns = cKinTree::eJointDesc
  (reference: DeepMimicCore/anim/KinTree.h:24-47, parse KinTree.cpp:433-481,959-987)
* ``body_defs``  [J x 17]  columns = cKinTree::eBodyParam
  (KinTree.h:49-70, parse KinTree.cpp:125-197)
* ``pd_params``  [J x 2]   (Kp, Kd) per joint (sim/PDController.cpp:50-93)
* ``frames``     [F x (1+P)] raw motion frames, column 0 = frame duration
  (anim/Motion.cpp:344-378)

Arg-file semantics follow util/ArgParser.cpp:31-120 (``--key v1 v2``, ``#``
comments, first occurrence of a key wins, ``--arg_file`` chaining as in
DeepMimicCore.cpp:25-44).
"""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

# --- cKinTree enums (anim/KinTree.h:13-70) ---------------------------------
JOINT_TYPES = ["revolute", "planar", "prismatic", "fixed", "spherical", "none"]
JT_REVOLUTE, JT_PLANAR, JT_PRISMATIC, JT_FIXED, JT_SPHERICAL, JT_NONE = range(6)

JOINT_DESC_KEYS = [
    "Type", "Parent", "AttachX", "AttachY", "AttachZ",
    "AttachThetaX", "AttachThetaY", "AttachThetaZ",
    "LimLow0", "LimLow1", "LimLow2", "LimHigh0", "LimHigh1", "LimHigh2",
    "TorqueLim", "ForceLim", "IsEndEffector", "DiffWeight", "Offset",
]
(JD_TYPE, JD_PARENT, JD_AX, JD_AY, JD_AZ, JD_ATX, JD_ATY, JD_ATZ,
 JD_LL0, JD_LL1, JD_LL2, JD_LH0, JD_LH1, JD_LH2,
 JD_TORQUE_LIM, JD_FORCE_LIM, JD_IS_EE, JD_DIFF_W, JD_PARAM_OFFSET) = range(19)

BODY_KEYS = [
    "Shape", "Mass", "ColGroup", "EnableFallContact",
    "AttachX", "AttachY", "AttachZ", "AttachThetaX", "AttachThetaY", "AttachThetaZ",
    "Param0", "Param1", "Param2", "ColorR", "ColorG", "ColorB", "ColorA",
]
(BD_SHAPE, BD_MASS, BD_COLGROUP, BD_FALL, BD_AX, BD_AY, BD_AZ, BD_ATX, BD_ATY, BD_ATZ,
 BD_P0, BD_P1, BD_P2, BD_CR, BD_CG, BD_CB, BD_CA) = range(17)

SHAPES = ["null", "box", "capsule", "sphere", "cylinder", "plane"]  # anim/Shape.h:8-17
SH_NULL, SH_BOX, SH_CAPSULE, SH_SPHERE, SH_CYLINDER, SH_PLANE = range(6)


def _joint_desc_default() -> np.ndarray:
    """cKinTree::BuildJointDesc() (anim/KinTree.cpp:1132-1156)."""
    d = np.zeros(19)
    d[JD_TYPE] = JT_REVOLUTE
    d[JD_PARENT] = -1
    d[JD_LL0:JD_LL2 + 1] = 1
    d[JD_LH0:JD_LH2 + 1] = 0
    d[JD_TORQUE_LIM] = np.inf
    d[JD_FORCE_LIM] = np.inf
    d[JD_DIFF_W] = 1
    return d


def _body_def_default() -> np.ndarray:
    """cKinTree::BuildBodyDef() (anim/KinTree.cpp:1158-1179)."""
    d = np.zeros(17)
    d[BD_SHAPE] = SH_NULL
    d[BD_COLGROUP] = -1
    d[BD_CA] = 1
    return d


def joint_param_size(jtype: int, is_root: bool) -> int:
    """cKinTree::GetParamSize (anim/KinTree.cpp:768-802)."""
    if is_root:
        return 7
    return {JT_REVOLUTE: 1, JT_PRISMATIC: 1, JT_PLANAR: 3, JT_FIXED: 0, JT_SPHERICAL: 4}[jtype]


def parse_skeleton(char_json: dict) -> np.ndarray:
    joints = char_json["Skeleton"]["Joints"]
    J = len(joints)
    jm = np.zeros((J, 19))
    for j, jj in enumerate(joints):
        d = _joint_desc_default()
        d[JD_TYPE] = JOINT_TYPES.index(jj["Type"])
        for i, key in enumerate(JOINT_DESC_KEYS):
            if i != JD_TYPE and key in jj and jj[key] is not None:
                d[i] = float(jj[key])
        jm[j] = d
    for j in range(J):
        if int(jm[j, JD_PARENT]) >= j:
            raise ValueError("parent id must be < child id (joint %d)" % j)
    # PostProcessJointMat (anim/KinTree.cpp:1005-1020)
    off = 0
    for j in range(J):
        jm[j, JD_PARAM_OFFSET] = off
        off += joint_param_size(int(jm[j, JD_TYPE]), j == 0)
    jm[0, JD_AX:JD_AZ + 1] = 0
    return jm


def parse_body_defs(char_json: dict) -> np.ndarray:
    defs = char_json["BodyDefs"]
    bd = np.zeros((len(defs), 17))
    for b, bj in enumerate(defs):
        d = _body_def_default()
        d[BD_SHAPE] = SHAPES.index(bj.get("Shape", "null"))
        for i, key in enumerate(BODY_KEYS):
            v = bj.get(key)
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                d[i] = float(v)
        bd[b] = d
    return bd


@dataclass
class SceneConfig:
    """Values of the arg-file keys the hot path consumes."""
    scene: str = "imitate"
    num_update_substeps: int = 1        # DeepMimicCore.cpp:43
    num_sim_substeps: int = 1           # scenes/SceneSimChar.cpp:60
    world_scale: float = 1.0            # scenes/SceneSimChar.cpp:61
    gravity: Sequence[float] = (0.0, -9.8, 0.0)   # util/MathUtil.h:25
    fall_contact_bodies: Optional[List[int]] = None
    sync_char_root_pos: bool = True     # scenes/SceneImitate.cpp:132
    sync_char_root_rot: bool = False
    enable_rand_rot_reset: bool = False
    enable_root_rot_fail: bool = False
    enable_fall_end: bool = True        # scenes/RLSceneSimChar.cpp:5
    enable_char_contact_fall: bool = True
    enable_rand_char_placement: bool = True
    enable_amp_obs_local_root: bool = False   # scenes/SceneImitateAMP.cpp:30,42 (`--scene imitate_amp` only)
    enable_test_time_warp: bool = True        # :31,43; the test-mode time-warp score is not on the accelerated path
    time_lim_min: float = np.inf        # util/Timer.cpp:7-8
    time_lim_max: float = np.inf
    time_lim_exp: float = 1.0
    time_end_lim_min: Optional[float] = None
    time_end_lim_max: Optional[float] = None
    time_end_lim_exp: Optional[float] = None
    timer_type: str = "uniform"
    anneal_samples: int = -1
    character_file: str = ""
    char_ctrl_file: str = ""
    motion_file: str = ""
    terrain_file: str = ""
    kin_ctrl: str = "motion"            # "clips": multi-clip dataset (anim/ClipsController.cpp), goal scenes
    # ---- goal-conditioned AMP task scenes (scenes/SceneTargetAMP.cpp:83-120, SceneHeadingAMP.cpp:45-93): constructor defaults
    rand_target_time_min: float = 1.0
    rand_target_time_max: float = 5.0
    max_target_dist: float = 3.0
    target_succ_dist: float = 0.5
    tar_fail_dist: float = np.inf
    tar_speed: float = 1.0
    enable_min_tar_vel: bool = False
    pos_reward_scale: float = 1.0
    max_heading_turn_rate: float = 0.15
    sharp_turn_prob: float = 0.025
    speed_change_prob: float = 0.1
    tar_speed_min: Optional[float] = None
    tar_speed_max: Optional[float] = None
    vel_reward_scale: float = 1.0
    # ---- heading_amp_getup (scenes/SceneHeadingAMPGetup.cpp:58-83)
    getup_motion_ids: Optional[List[int]] = None
    getup_height_root: float = 0.5
    getup_height_head: float = 0.5
    head_id: int = 0
    recover_episode_prob: float = 0.0
    # ---- strike_amp (scenes/SceneStrikeAMP.cpp:189-231)
    target_min: Sequence[float] = (-0.5, 1.2, 0.6)
    target_max: Sequence[float] = (0.5, 1.4, 1.1)
    target_radius: float = 0.2
    target_hit_reset_time: float = 2.0
    tar_reward_scale: float = 2.0
    hit_tar_speed: float = 1.5
    init_hit_prob: float = 0.0
    tar_far_prob: float = 0.4
    tar_near_dist: float = 1.4
    strike_bodies: Optional[List[int]] = None
    fail_tar_contact_bodies: Optional[List[int]] = None
    # ---- dribble_amp (scenes/SceneDribbleAMP.cpp:124-149)
    rand_tar_obj_time_min: float = 100.0
    rand_tar_obj_time_max: float = 200.0
    min_tar_obj_dist: float = 0.5
    max_tar_obj_dist: float = 10.0
    ball_radius: float = 0.2
    # ---- random perturbations (cSceneSimChar::tPerturbParams, scenes/SceneSimChar.cpp:41-51; keys :92-99, the duration key's typo is the reference's)
    enable_rand_perturbs: bool = False
    perturb_time_min: float = np.inf
    perturb_time_max: float = np.inf
    min_perturb: float = 50.0
    max_perturb: float = 100.0
    min_pertrub_duration: float = 0.1
    max_perturb_duration: float = 0.5
    perturb_part_ids: Optional[List[int]] = None


# cSceneDribbleAMP::BuildTarObjs (SceneDribbleAMP.cpp:398-420): the ball's constants are literals there, not arg-file keys
BALL_MASS, BALL_FRICTION, BALL_LIN_DAMPING, BALL_ANG_DAMPING = 0.43, 0.4, 0.4, 0.4
def amp_local_root(cfg) -> bool:
    """The `--enable_amp_obs_local_root` the scene actually runs with.  Only cSceneImitateAMP::ParseArgs reads the key (scenes/SceneImitateAMP.cpp:38-44), and the
    task scenes skip it: cSceneTargetAMP::ParseArgs calls cSceneImitate::ParseArgs directly (scenes/SceneTargetAMP.cpp:103-105; heading / strike / dribble / get-up
    chain to it), so for them the flag keeps the constructor's `false` whatever the arg file says -- every shipped task arg file says `true`.  Found by running the
    compiled scene classes (tests/test_ref_draw_order.py: cSceneImitateAMP::RecordAMPObsExpert of a heading_amp scene)."""
    return bool(getattr(cfg, "enable_amp_obs_local_root", False)) and cfg.scene not in GOAL_SCENES


GOAL_SCENES = {"target_amp": 1, "heading_amp": 2, "heading_amp_getup": 3, "strike_amp": 4, "dribble_amp": 5}
AMP_SCENES = ("imitate_amp",) + tuple(GOAL_SCENES)


@dataclass
class SceneTables:
    """Raw tables for one imitate scene, reference memory layout."""
    joint_mat: np.ndarray
    body_defs: np.ndarray
    pd_params: np.ndarray            # [J x 2] Kp, Kd
    frames: np.ndarray               # [F x (1+P)] duration + pose
    loop: bool
    enable_phase_input: bool = False
    record_world_root_pos: bool = False
    record_world_root_rot: bool = False
    query_rate: float = 30.0         # key "QueryRate" (sim/CtController.cpp:165)
    cfg: SceneConfig = field(default_factory=SceneConfig)
    joint_names: List[str] = field(default_factory=list)
    # multi-clip datasets (`--kin_ctrl clips`): `frames` is the concatenation of every clip's frames; clip c owns rows
    # clip_starts[c] .. clip_starts[c+1]-1.  None = one clip (the whole of `frames`, loop mode `loop`).
    clip_starts: Optional[np.ndarray] = None
    clip_weights: Optional[np.ndarray] = None      # cClipsController::tMotionEntry::mWeight
    clip_loops: Optional[np.ndarray] = None

    @property
    def num_clips(self) -> int:
        return 1 if self.clip_starts is None else len(self.clip_starts) - 1

    @property
    def goal_kind(self) -> int:
        """0: no goal (imitate / imitate_amp), 1: target_amp, 2: heading_amp, 3: heading_amp_getup, 4: strike_amp"""
        return GOAL_SCENES.get(self.cfg.scene, 0)

    @property
    def goal_dim(self) -> int:
        # cSceneTargetAMP / cSceneHeadingAMP::GetGoalSize: 3; cSceneHeadingAMPGetup (+ get-up phase) and cSceneStrikeAMP (pos + hit phase): 4
        return (0, 3, 3, 4, 4, 3)[self.goal_kind]           # cSceneDribbleAMP::RecordGoal: direction (2) + distance from the ball to the target

    def clip_duration(self, c: int) -> float:
        """cMotion::GetDuration of clip c: the frame durations but the last one's (anim/Motion.cpp PostProcessFrames)"""
        col = self.frames[:-1, 0] if self.clip_starts is None else self.frames[int(self.clip_starts[c]):int(self.clip_starts[c + 1]) - 1, 0]
        d = 0.0
        for x in col.tolist():        # in frame order, like the reference's loop (numpy's sum is pairwise: other last bits)
            d += x
        return d

    @property
    def getup_time(self) -> float:
        """cSceneHeadingAMPGetup::CalcGetupTime (:266-291): the longest get-up clip"""
        ids = self.cfg.getup_motion_ids or []
        return max([self.clip_duration(i) for i in ids], default=0.0)

    @property
    def getup_clip_mask(self) -> int:
        m = 0
        for i in (self.cfg.getup_motion_ids or []):
            if not 0 <= int(i) < min(self.num_clips, 31):
                raise ValueError("getup_motion_ids: %r is not a clip of the dataset (%d clips)" % (i, self.num_clips))
            m |= 1 << int(i)
        return m

    @property
    def num_joints(self) -> int:
        return self.joint_mat.shape[0]

    @property
    def pose_dim(self) -> int:
        return int(self.joint_mat[-1, JD_PARAM_OFFSET]) + joint_param_size(
            int(self.joint_mat[-1, JD_TYPE]), self.num_joints == 1)

    @property
    def action_dim(self) -> int:
        a = 0
        for j in range(1, self.num_joints):
            t = int(self.joint_mat[j, JD_TYPE])
            a += 3 if t == JT_SPHERICAL else joint_param_size(t, False)
        return a

    @property
    def state_dim(self) -> int:
        J = self.num_joints
        s = (1 if self.enable_phase_input else 0) + (J * 9 + 1) + J * 6
        if self.cfg.scene == "dribble_amp":
            s += 15                                # cSceneDribbleAMP::GetTaskStateSize (SceneDribbleAMP.cpp:541-545): the ball in the origin frame
        return s

    def fall_mask(self) -> np.ndarray:
        """Per-link fall-contact flags: args override JSON (scenes/SceneSimChar.cpp:460-476)."""
        J = self.num_joints
        m = (self.body_defs[:, BD_FALL] != 0).astype(np.int32)
        if self.cfg.fall_contact_bodies:
            m[:] = 0
            for b in self.cfg.fall_contact_bodies:
                m[b] = 1
        return m

    # ---- (de)serialisation to the compact asset format used in-tree ----
    def to_json(self) -> dict:
        c = self.cfg
        return {
            "format": "deepmimic_amd.scene_tables.v1",
            "joint_names": self.joint_names,
            "joint_mat": _mat_to_list(self.joint_mat),
            "body_defs": _mat_to_list(self.body_defs),
            "pd_params": _mat_to_list(self.pd_params),
            "frames": _mat_to_list(self.frames),
            "loop": bool(self.loop),
            "enable_phase_input": bool(self.enable_phase_input),
            "record_world_root_pos": bool(self.record_world_root_pos),
            "record_world_root_rot": bool(self.record_world_root_rot),
            "query_rate": float(self.query_rate),
            "cfg": {k: _json_val(v) for k, v in c.__dict__.items()},
            "clip_starts": None if self.clip_starts is None else [int(x) for x in self.clip_starts],
            "clip_weights": None if self.clip_weights is None else [float(x) for x in self.clip_weights],
            "clip_loops": None if self.clip_loops is None else [int(x) for x in self.clip_loops],
        }

    @staticmethod
    def from_json(d: dict) -> "SceneTables":
        cfg = SceneConfig()
        for k, v in d.get("cfg", {}).items():
            if v == "inf":
                v = np.inf
            setattr(cfg, k, v)
        return SceneTables(
            joint_mat=_list_to_mat(d["joint_mat"]),
            body_defs=_list_to_mat(d["body_defs"]),
            pd_params=_list_to_mat(d["pd_params"]),
            frames=_list_to_mat(d["frames"]),
            loop=bool(d["loop"]),
            enable_phase_input=bool(d["enable_phase_input"]),
            record_world_root_pos=bool(d["record_world_root_pos"]),
            record_world_root_rot=bool(d["record_world_root_rot"]),
            query_rate=float(d.get("query_rate", 30.0)),
            cfg=cfg,
            joint_names=list(d.get("joint_names", [])),
            clip_starts=None if d.get("clip_starts") is None else np.array(d["clip_starts"], dtype=np.int32),
            clip_weights=None if d.get("clip_weights") is None else np.array(d["clip_weights"], dtype=np.float64),
            clip_loops=None if d.get("clip_loops") is None else np.array(d["clip_loops"], dtype=np.int32),
        )


def _json_val(v):
    if isinstance(v, float) and np.isinf(v):
        return "inf"
    if isinstance(v, tuple):
        return list(v)
    return v


def _mat_to_list(m: np.ndarray):
    return [[("inf" if np.isposinf(x) else "-inf" if np.isneginf(x) else float(x)) for x in row]
            for row in np.asarray(m, dtype=np.float64)]


def _list_to_mat(rows) -> np.ndarray:
    return np.array([[np.inf if x == "inf" else -np.inf if x == "-inf" else x for x in r]
                     for r in rows], dtype=np.float64)


# --- arg files ---------------------------------------------------------------
class ArgParser:
    """util/ArgParser.cpp:31-120.  First occurrence of a key wins."""

    def __init__(self, args: Sequence[str] = ()):
        self.table: Dict[str, List[str]] = {}
        self.load_args(args)

    def load_args(self, arg_strs: Sequence[str]) -> None:
        key, vals = "", []
        for s in arg_strs:
            if s.startswith("#"):
                continue
            if len(s) >= 3 and s[0] == "-" and s[1] == "-":   # cArgParser::IsKey
                if key and key not in self.table:
                    self.table[key] = vals
                key, vals = s[2:], []
            else:
                vals.append(s)
        if key and key not in self.table:
            self.table[key] = vals

    def load_file(self, path: str) -> bool:
        if not os.path.isfile(path):
            return False
        toks: List[str] = []
        with open(path) as f:
            for line in f:
                if not line or line.startswith("#"):
                    continue
                toks.extend(t for t in re.split(r"[ \t\n\r,]+", line) if t)
        self.load_args(toks)
        return True

    def get(self, key, default=None):
        return self.table.get(key, default)

    def str(self, key, default=""):
        v = self.table.get(key)
        return v[0] if v else default

    def float(self, key, default):
        v = self.table.get(key)
        return float(v[0]) if v else default

    def int(self, key, default):
        v = self.table.get(key)
        return int(v[0]) if v else default

    def bool(self, key, default):
        v = self.table.get(key)
        if not v:
            return default
        return v[0] in ("true", "1", "True", "T", "t")   # cArgParser::ParseBool(str)

    def ints(self, key):
        v = self.table.get(key)
        return [int(x) for x in v] if v else None

    def floats(self, key):
        v = self.table.get(key)
        return [float(x) for x in v] if v else None


def parse_scene_config(parser: ArgParser) -> SceneConfig:
    c = SceneConfig()
    c.scene = parser.str("scene", c.scene)
    c.num_update_substeps = parser.int("num_update_substeps", c.num_update_substeps)
    c.num_sim_substeps = parser.int("num_sim_substeps", c.num_sim_substeps)
    c.world_scale = parser.float("world_scale", c.world_scale)
    g = parser.floats("gravity")
    if g:
        c.gravity = tuple(g[:3])
    c.fall_contact_bodies = parser.ints("fall_contact_bodies")
    for k in ("sync_char_root_pos", "sync_char_root_rot", "enable_rand_rot_reset",
              "enable_root_rot_fail", "enable_fall_end", "enable_char_contact_fall",
              "enable_rand_char_placement", "enable_amp_obs_local_root", "enable_test_time_warp"):
        setattr(c, k, parser.bool(k, getattr(c, k)))
    c.time_lim_min = parser.float("time_lim_min", c.time_lim_min)
    c.time_lim_max = parser.float("time_lim_max", c.time_lim_max)
    c.time_lim_exp = parser.float("time_lim_exp", c.time_lim_exp)
    # mTimerParamsEnd starts as a copy of mTimerParams (scenes/RLSceneSimChar.cpp:19-22)
    c.time_end_lim_min = parser.float("time_end_lim_min", c.time_lim_min)
    c.time_end_lim_max = parser.float("time_end_lim_max", c.time_lim_max)
    c.time_end_lim_exp = parser.float("time_end_lim_exp", c.time_lim_exp)
    c.timer_type = parser.str("timer_type", "uniform") or "uniform"
    c.anneal_samples = parser.int("anneal_samples", c.anneal_samples)
    c.character_file = parser.str("character_files", "")
    c.char_ctrl_file = parser.str("char_ctrl_files", "")
    c.motion_file = parser.str("motion_file", "")
    c.terrain_file = parser.str("terrain_file", "")
    c.kin_ctrl = parser.str("kin_ctrl", "motion") or "motion"
    for k in ("rand_target_time_min", "rand_target_time_max", "max_target_dist", "target_succ_dist", "tar_fail_dist", "tar_speed",
              "pos_reward_scale", "max_heading_turn_rate", "sharp_turn_prob", "speed_change_prob", "vel_reward_scale"):
        setattr(c, k, parser.float(k, getattr(c, k)))
    c.enable_min_tar_vel = parser.bool("enable_min_tar_vel", c.enable_min_tar_vel)
    if c.scene in ("heading_amp", "heading_amp_getup"):
        # cSceneHeadingAMP(): target timer 0.2 .. 0.5 s unless the args say otherwise (SceneHeadingAMP.cpp:45-48)
        c.rand_target_time_min = parser.float("rand_target_time_min", 0.2)
        c.rand_target_time_max = parser.float("rand_target_time_max", 0.5)
    # cSceneHeadingAMP::ParseArgs (:66-84): the speed range defaults to [tar_speed, tar_speed]; tar_speed is clamped into it
    c.tar_speed_min = parser.float("tar_speed_min", c.tar_speed)
    c.tar_speed_max = parser.float("tar_speed_max", c.tar_speed)
    if c.scene in ("heading_amp", "heading_amp_getup"):
        c.tar_speed = min(max(c.tar_speed, c.tar_speed_min), c.tar_speed_max)
    c.getup_motion_ids = parser.ints("getup_motion_ids")
    for k in ("getup_height_root", "getup_height_head", "recover_episode_prob", "target_radius", "target_hit_reset_time", "tar_reward_scale",
              "hit_tar_speed", "init_hit_prob", "tar_far_prob", "tar_near_dist"):
        setattr(c, k, parser.float(k, getattr(c, k)))
    c.head_id = parser.int("head_id", c.head_id)
    v = parser.floats("target_min")
    if v:
        c.target_min = tuple(v[:3])
    v = parser.floats("target_max")
    if v:
        c.target_max = tuple(v[:3])
    if c.scene == "dribble_amp":
        # cSceneDribbleAMP(): target timer 50 .. 100 s unless the args say otherwise (SceneDribbleAMP.cpp:124-133)
        c.rand_target_time_min = parser.float("rand_target_time_min", 50.0)
        c.rand_target_time_max = parser.float("rand_target_time_max", 100.0)
    for k in ("rand_tar_obj_time_min", "rand_tar_obj_time_max", "min_tar_obj_dist", "max_tar_obj_dist", "ball_radius"):
        setattr(c, k, parser.float(k, getattr(c, k)))
    c.strike_bodies = parser.ints("strike_bodies")
    c.fail_tar_contact_bodies = parser.ints("fail_tar_contact_bodies")
    c.enable_rand_perturbs = parser.bool("enable_rand_perturbs", c.enable_rand_perturbs)
    for k in ("perturb_time_min", "perturb_time_max", "min_perturb", "max_perturb", "min_pertrub_duration", "max_perturb_duration"):
        setattr(c, k, parser.float(k, getattr(c, k)))
    c.perturb_part_ids = parser.ints("perturb_part_ids")
    return c


def timer_limits(cfg: SceneConfig, test_mode: bool = False, sample_count: int = 0):
    """(min, max) of the uniform episode-length draw at the next Reset.

    Train mode: `cRLSceneSimChar::UpdateTimerParams` (scenes/RLSceneSimChar.cpp:338-347) blends `time_lim_*` towards
    `time_end_lim_*` by lerp = clamp(sample_count / anneal_samples, 0, 1)^4 (`SetupTimerAnnealer`, :330-336;
    `cTimer::tParams::Blend`, util/Timer.cpp:12-20; `cAnnealer::Eval`, util/Annealer.cpp) when `anneal_samples` > 0.
    Test mode: `ResetTimers` (:277-284) pins the limit to `time_end_lim_max`.  An unset limit is +inf (util/Timer.cpp:7-8);
    the NaN the reference's `0 * inf` would produce never satisfies `mTime >= mMaxTime`, i.e. behaves as +inf."""
    if cfg.timer_type not in ("uniform", "exp"):
        raise ValueError("unsupported timer type %r (util/Timer.cpp:27-45: uniform | exp)" % cfg.timer_type)
    emin = cfg.time_lim_min if cfg.time_end_lim_min is None else cfg.time_end_lim_min
    emax = cfg.time_lim_max if cfg.time_end_lim_max is None else cfg.time_end_lim_max
    if test_mode:
        return float(emax), float(emax)
    lerp = 0.0
    if cfg.anneal_samples > 0:
        lerp = min(max(float(sample_count) / cfg.anneal_samples, 0.0), 1.0) ** 4.0

    def blend(a, b):
        with np.errstate(invalid="ignore"):
            v = np.float64(1.0 - lerp) * np.float64(a) + np.float64(lerp) * np.float64(b)
        return float("inf") if np.isnan(v) else float(v)

    return blend(cfg.time_lim_min, emin), blend(cfg.time_lim_max, emax)


def timer_exp(cfg: SceneConfig, test_mode: bool = False, sample_count: int = 0) -> float:
    """mTimeExp of the blended timer parameters (`cTimer::tParams::Blend`, util/Timer.cpp:12-20), for `--timer_type exp`."""
    eexp = cfg.time_lim_exp if cfg.time_end_lim_exp is None else cfg.time_end_lim_exp
    lerp = 0.0
    if cfg.anneal_samples > 0 and not test_mode:
        lerp = min(max(float(sample_count) / cfg.anneal_samples, 0.0), 1.0) ** 4.0
    return float((1.0 - lerp) * cfg.time_lim_exp + lerp * eexp)


def draw_time_limit(timer_type: str, lo: float, hi: float, exp: float, u: float) -> float:
    """`cTimer::Reset` (util/Timer.cpp:55-73) from one uniform draw u in [0, 1): uniform -> U[lo, hi]; exp -> min(lo + Exp(rate 1 / exp), hi)
    (`cMathUtil::RandDoubleExp` = std::exponential_distribution: -ln(1 - u) / rate)."""
    if timer_type == "exp":
        return float(min(lo - exp * np.log1p(-u), hi))
    return float(lo + (hi - lo) * u) if hi > lo else float(hi)


# --- scene loading -------------------------------------------------------------
def load_scene(char_file: str, ctrl_file: str, motion_file: str,
               cfg: Optional[SceneConfig] = None, data_root: Optional[str] = None) -> SceneTables:
    with open(char_file) as f:
        cj = json.load(f)
    char_json = cj
    with open(ctrl_file) as f:
        kj = json.load(f)
    with open(motion_file) as f:
        mj = json.load(f)
    jm = parse_skeleton(cj)
    bd = parse_body_defs(cj)
    J = jm.shape[0]
    if bd.shape[0] != J:
        raise ValueError("joint / body-def count mismatch")
    pds = kj["PDControllers"]
    if len(pds) != J:
        raise ValueError("PD controller count mismatch")
    pd = np.array([[float(p.get("Kp", 0)), float(p.get("Kd", 0))] for p in pds])
    clip_starts = clip_weights = clip_loops = None
    if "Frames" not in mj:
        if "Motions" not in mj:
            raise ValueError("%s has neither \"Frames\" nor \"Motions\"" % motion_file)
        # multi-clip dataset (cClipsController::LoadMotions, anim/ClipsController.cpp:150-188): files relative to the data root
        root = data_root if data_root is not None else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(motion_file))))
        all_frames, starts, weights, loops = [], [0], [], []
        for ent in mj["Motions"]:
            path = ent.get("File", "")
            path = path if os.path.isabs(path) else os.path.join(root, path)
            with open(path) as f:
                cj = json.load(f)
            fr = np.array(cj["Frames"], dtype=np.float64)
            lp = cj.get("Loop", "none")
            if lp not in ("none", "wrap"):
                raise ValueError("unsupported loop mode %r in %s" % (lp, path))
            all_frames.append(fr); starts.append(starts[-1] + fr.shape[0])
            weights.append(float(ent.get("Weight", 1.0))); loops.append(1 if lp == "wrap" else 0)
        frames = np.concatenate(all_frames, 0)
        loop_str = "wrap" if loops[0] else "none"
        clip_starts, clip_weights, clip_loops = np.array(starts, np.int32), np.array(weights), np.array(loops, np.int32)
    else:
        frames = np.array(mj["Frames"], dtype=np.float64)
        loop_str = mj.get("Loop", "none")
        if loop_str not in ("none", "wrap"):
            raise ValueError("unsupported loop mode %r" % loop_str)
    t = SceneTables(
        joint_mat=jm, body_defs=bd, pd_params=pd, frames=frames, loop=(loop_str == "wrap"),
        enable_phase_input=bool(kj.get("EnablePhaseInput", False)),
        record_world_root_pos=bool(kj.get("RecordWorldRootPos", False)),
        record_world_root_rot=bool(kj.get("RecordWorldRootRot", False)),
        query_rate=float(kj.get("QueryRate", 30.0)),
        cfg=cfg or SceneConfig(),
        joint_names=[j.get("Name", "") for j in char_json["Skeleton"]["Joints"]],
        clip_starts=clip_starts, clip_weights=clip_weights, clip_loops=clip_loops,
    )
    if frames.shape[1] - 1 != t.pose_dim:
        raise ValueError("DOF mismatch, char dof %d, motion dof %d" % (t.pose_dim, frames.shape[1] - 1))
    return t


def load_kin_scene(char_file: str, motion_file: str) -> SceneTables:
    """`--scene kin_char` (scenes/SceneKinChar.cpp): skeleton + one clip, no bodies, no controller (the viewer's motion playback)."""
    with open(char_file) as f:
        cj = json.load(f)
    with open(motion_file) as f:
        mj = json.load(f)
    jm = parse_skeleton(cj)
    frames = np.array(mj["Frames"], dtype=np.float64)
    J = jm.shape[0]
    cfg = SceneConfig(); cfg.scene = "kin_char"; cfg.character_file = char_file; cfg.motion_file = motion_file
    t = SceneTables(joint_mat=jm, body_defs=np.zeros((J, 17)), pd_params=np.zeros((J, 2)), frames=frames,
                    loop=(mj.get("Loop", "none") == "wrap"), cfg=cfg)
    if t.pose_dim != frames.shape[1] - 1:
        raise ValueError("DOF mismatch, char dof %d, motion dof %d" % (t.pose_dim, frames.shape[1] - 1))
    return t


def load_scene_from_args(args: Sequence[str], data_root: str = ".") -> SceneTables:
    """Mirror of cDeepMimicCore::ParseArgs + cSceneImitate::ParseArgs for `--scene imitate`."""
    p = ArgParser(args)
    arg_file = p.str("arg_file", "")
    if arg_file:
        path = arg_file if os.path.isabs(arg_file) else os.path.join(data_root, arg_file)
        if not p.load_file(path):
            raise FileNotFoundError("Failed to load args from: %s" % arg_file)
    cfg = parse_scene_config(p)
    if cfg.scene == "kin_char":             # motion playback without a simulated character: host-side only (the facade serves it)
        cf, mf = p.str("character_file", ""), p.str("motion_file", "")
        rs = lambda pth: pth if os.path.isabs(pth) else os.path.join(data_root, pth)
        return load_kin_scene(rs(cf), rs(mf))
    if cfg.scene != "imitate" and cfg.scene not in AMP_SCENES:
        raise ValueError("only `--scene imitate`, `imitate_amp`, `heading_amp`, `heading_amp_getup`, `target_amp`, `strike_amp` and `dribble_amp` "
                         "are on the accelerated path (got %r)" % cfg.scene)

    def res(pth):
        return pth if os.path.isabs(pth) else os.path.join(data_root, pth)

    return load_scene(res(cfg.character_file), res(cfg.char_ctrl_file), res(cfg.motion_file), cfg, data_root=data_root)


ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


# the reference arg files whose compiled copies ship in-tree (tools/compile_assets.py made assets/<name>.json from exactly these): where the reference's data
# files are absent (the GPU box), `ParseArgs(["--arg_file", <key>])` of the drop-in loads the compiled copy
ARG_FILE_ASSETS = {
    "args/run_humanoid3d_walk_args.txt": "humanoid3d_walk", "args/train_humanoid3d_spinkick_args.txt": "humanoid3d_spinkick",
    "args/train_dog3d_pace_args.txt": "dog3d_pace", "args/run_humanoid3d_run_args.txt": "humanoid3d_run",
    "args/run_humanoid3d_backflip_args.txt": "humanoid3d_backflip", "args/run_dog3d_spin_args.txt": "dog3d_spin",
    "args/train_amp_heading_humanoid3d_zombie_args.txt": "amp_heading_zombie", "args/train_amp_target_humanoid3d_zombie_args.txt": "amp_target_zombie",
    "args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt": "amp_heading_getup", "args/train_amp_dribble_humanoid3d_zombie_args.txt": "amp_dribble_zombie",
}


def load_asset(name: str) -> SceneTables:
    """Load an in-tree compiled scene (``assets/<name>.json``), e.g. ``humanoid3d_walk``."""
    with open(os.path.join(ASSET_DIR, name + ".json")) as f:
        return SceneTables.from_json(json.load(f))


# --- host-side kinematics (numpy): used by the facade's test-mode time-warp score, never by the stepping path --------------
def _qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3], a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def _qrot(q, v):
    u = q[1:4]
    uv = 2.0 * np.cross(u, v)
    return v + q[0] * uv + np.cross(u, uv)


def _slerp(a, b, t):
    d = float(np.dot(a, b))
    ad = abs(d)
    if ad >= 1.0 - np.finfo(np.float64).eps:
        s0, s1 = 1.0 - t, t
    else:
        th = np.arccos(ad); st = np.sin(th)
        s0, s1 = np.sin((1.0 - t) * th) / st, np.sin(t * th) / st
    if d < 0:
        s1 = -s1
    return s0 * a + s1 * b


def _rot_euler(e):
    """cMathUtil::RotateMat(euler) = Rz Ry Rx (util/MathUtil.cpp:159-186)"""
    xs, xc, ys, yc, zs, zc = np.sin(e[0]), np.cos(e[0]), np.sin(e[1]), np.cos(e[1]), np.sin(e[2]), np.cos(e[2])
    return np.array([[yc * zc, xs * ys * zc - xc * zs, xc * ys * zc + xs * zs],
                     [yc * zs, xs * ys * zs + xc * zc, xc * ys * zs - xs * zc],
                     [-ys, xs * yc, xc * yc]])


def _quat_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def joint_world_positions(tables: SceneTables, pose) -> np.ndarray:
    """cKinTree::CalcJointWorldPos for every joint (anim/KinTree.cpp:538-553, 1034-1091): J x 3"""
    jm = tables.joint_mat
    J = jm.shape[0]
    R = [None] * J
    p = np.zeros((J, 3))
    for j in range(J):
        par = int(jm[j, JD_PARENT]); ty = int(jm[j, JD_TYPE]); off = int(jm[j, JD_PARAM_OFFSET])
        if j == 0:
            R[j] = _quat_mat(np.asarray(pose[3:7], dtype=np.float64)); p[j] = pose[0:3]
            continue
        A = _rot_euler(jm[j, JD_ATX:JD_ATZ + 1])
        if ty == JT_SPHERICAL:
            Rj = _quat_mat(np.asarray(pose[off:off + 4], dtype=np.float64))
        elif ty == JT_REVOLUTE:
            c, s = np.cos(pose[off]), np.sin(pose[off])
            Rj = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        else:
            Rj = np.eye(3)
        R[j] = R[par] @ A @ Rj
        p[j] = p[par] + R[par] @ jm[j, JD_AX:JD_AZ + 1]
    return p


class KinSampler:
    """cKinCharacter::CalcPose on the host (anim/KinCharacter.cpp:363-386, MotionController.cpp:25-41, Motion.cpp:249-293): the
    pose of the kinematic character at a clip time for a given origin.  `clip`: which clip of a multi-clip dataset (cClipsController's active motion)."""

    def __init__(self, tables: SceneTables, clip: int = 0):
        self.t = tables
        if tables.clip_starts is None:
            fr = np.array(tables.frames, dtype=np.float64)
        else:
            fr = np.array(tables.frames[int(tables.clip_starts[clip]):int(tables.clip_starts[clip + 1])], dtype=np.float64)
        self.loop = bool(tables.loop if tables.clip_starts is None else tables.clip_loops[clip])
        self.times = np.concatenate([[0.0], np.cumsum(fr[:-1, 0])])
        self.frames = fr[:, 1:].copy()
        self.frames[:, 0] -= self.frames[0, 0]; self.frames[:, 2] -= self.frames[0, 2]       # PostProcessMotion
        self.quat_offs = [3] + [int(tables.joint_mat[j, JD_PARAM_OFFSET]) for j in range(1, tables.num_joints)
                                if int(tables.joint_mat[j, JD_TYPE]) == JT_SPHERICAL]
        for o in self.quat_offs:
            self.frames[:, o:o + 4] /= np.linalg.norm(self.frames[:, o:o + 4], axis=1, keepdims=True)
        self.duration = float(self.times[-1])
        self.cycle_delta = self.frames[-1, 0:3] - self.frames[0, 0:3]; self.cycle_delta[1] = 0

    def pose(self, time, origin_pos, origin_rot):
        dur = self.duration
        cc = int(np.floor(time / dur))
        cycle = cc if self.loop else min(max(cc, 0), 1)
        if not self.loop and time <= 0:
            idx, blend = 0, 0.0
        elif not self.loop and time >= dur:
            idx, blend = len(self.times) - 2, 1.0
        else:
            tt = time - cycle * dur
            idx = int(np.searchsorted(self.times, tt, side="right")) - 1
            idx = min(max(idx, 0), len(self.times) - 2)
            blend = (tt - self.times[idx]) / (self.times[idx + 1] - self.times[idx])
        f0, f1 = self.frames[idx], self.frames[idx + 1]
        out = (1.0 - blend) * f0 + blend * f1
        for o in self.quat_offs:
            q = _slerp(f0[o:o + 4], f1[o:o + 4], blend); out[o:o + 4] = q / np.linalg.norm(q)
        if self.loop:
            out[0:3] += cycle * self.cycle_delta
        orot = np.asarray(origin_rot, dtype=np.float64)
        out[0:3] = _qrot(orot, out[0:3]) + np.asarray(origin_pos, dtype=np.float64)
        q = _qmul(orot, out[3:7]); out[3:7] = q if q[0] >= 0 else -q
        return out


def time_warp_cost(data0: np.ndarray, data1: np.ndarray) -> float:
    """cDynamicTimeWarper::CalcAlignment (util/DynamicTimeWarper.cpp:117-140) with cSceneImitateAMP::TimeWarpCost
    (scenes/SceneImitateAMP.cpp:5-25: mean point distance): data0 [n x 3J], data1 [m x 3J]."""
    n, m = data0.shape[0], data1.shape[0]
    d = np.linalg.norm(data0.reshape(n, 1, -1, 3) - data1.reshape(1, m, -1, 3), axis=3).mean(axis=2)
    cost = np.full((n, m), np.inf)
    cost[0, 0] = 0.0
    for i in range(1, n):
        for j in range(1, m):
            cost[i, j] = d[i, j] + min(cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1])
    return float(cost[n - 1, m - 1])
