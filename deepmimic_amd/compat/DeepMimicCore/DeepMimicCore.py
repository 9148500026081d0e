"""Drop-in replacement for the reference's SWIG module ``DeepMimicCore.DeepMimicCore`` on the ``--scene imitate`` path.

The reference builds ``DeepMimicCore/DeepMimicCore.py`` + ``_DeepMimicCore.so`` with SWIG from
``DeepMimicCore/DeepMimicCore.i:1-34`` and ``env/deepmimic_env.py:3`` imports it as
``from DeepMimicCore import DeepMimicCore``.  Put ``<repo>/deepmimic_amd/compat`` (this directory's parent) and
``<repo>`` on ``PYTHONPATH`` ahead of the reference's own ``DeepMimicCore/`` directory and the unmodified
``env/deepmimic_env.py`` / ``learning/`` agents drive the MI355X library instead of Bullet.

``cDeepMimicCore`` mirrors ``DeepMimicCore/DeepMimicCore.h:9-87`` method for method (same names, argument meaning and
return types; vectors are returned as Python lists of float / int like SWIG's ``std::vector`` wrappers).  One instance is
one env (``dm_ctx`` with ``num_envs = 1``) on the GPU given by ``DM_DEVICE`` (default 0).  The batched extension --
thousands of envs per GPU, one kernel launch per control step -- is ``deepmimic_amd.core.BatchEnv`` /
``include/dm_hip.h:dm_step_batch``.

Launch economy.  The reference's driver (DeepMimic.py:62-80, learning/rl_world.py) calls, per 1/600 s update:
``NeedNewAction`` -> ``Update`` -> ``CheckValidEpisode`` -> ``IsEpisodeEnd``, and ``RecordState / CalcReward / SetAction``
once per 1/30 s.  Served naively that is >= 2 kernel launches and 3 device round trips per update.  Here:
* ``NeedNewAction`` and ``GetTime`` are answered from a host mirror of the controller / timer clocks (same double arithmetic
  as cMathUtil::CheckNextInterval), no device traffic;
* ``Update`` is one fused launch (update + RecordState / CalcReward / flags), the following queries read its outputs;
* with ``DM_FACADE_BATCH`` (default on) the first ``Update`` after a ``SetAction`` runs the whole control step -- all updates up
  to the next action boundary, ending early at the update where the episode is over -- in ONE launch; the driver's following
  ``Update`` calls of the same timestep only advance a host counter and its per-update ``IsEpisodeEnd / CheckValidEpisode``
  questions are answered from that launch's outcome.  A call that needs the state of an intermediate update (``RecordState``
  mid-step, a different timestep, a ``SetAction`` mid-step) rolls the env back to the snapshot taken before the launch and
  replays update by update, so results never depend on the batching.  (The launch also stops at the update after which the
  episode is invalid -- a link velocity beyond 100, ``cSimCharacter::HasVelExploded`` --, so ``CheckValidEpisode`` turns false at
  the update the reference's driver would see it at.)

Errors: the reference ``assert(false)``s (DeepMimicCore.cpp:36-40); this module raises ``RuntimeError`` instead.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.abspath(os.path.join(_HERE, "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from deepmimic_amd import model as _model  # noqa: E402
from deepmimic_amd.core import BatchEnv as _BatchEnv  # noqa: E402
from deepmimic_amd.core import RefRand as _RefRand  # noqa: E402

_DT_EPS = 0.0


class _TapeExhausted(RuntimeError):
    pass


class cDeepMimicCore(object):
    # cRLScene::eMode (scenes/RLScene.h:11-16)
    eModeTrain, eModeTest = 0, 1

    def __init__(self, enable_draw):
        if enable_draw:
            # DeepMimicCore.cpp:9-18 sets up GL state; rendering is outside the accelerated path
            raise RuntimeError("cDeepMimicCore(enable_draw=True): the MI355X path is headless (use enable_draw=False)")
        self._seed = 0
        self._args = None
        self._tables = None
        self._env = None
        self._mode = self.eModeTrain
        self._sample_count = 0
        self._num_update_substeps = 1
        self._cache = None
        self._time = 0.0
        self._playback_speed = 1.0
        self._updates_per_sec = 0.0
        # DM_RNG=reference (default here; "counter" = the batched path's streams keyed by (seed, env id, episode)): the episode draws of this one-env
        # drop-in -- clip time at every reset, episode time limit, the scene generator's expert-sample times -- come from the reference's own
        # generator (util/Rand.cpp: std::default_random_engine + <random> distributions, include/dm_hip.h dm_refrand_*) consumed in the
        # reference's call order, so that SeedRand(s) reproduces the reference's reset times and limits bit for bit (tests/test_ref_rng.py,
        # against the compiled cRand / cTimer of oracle/_ref).  cMathUtil::gRand starts time-seeded (util/MathUtil.cpp:6, Rand.cpp:6-9).
        self._ref_rng = os.environ.get("DM_RNG", "reference") == "reference"
        self._grand = self._srand = None
        if self._ref_rng:
            import time
            self._grand = _RefRand(int(time.time()), lib_path=os.environ.get("DM_HIP_LIB"))

    # ---- construction (DeepMimicCore.cpp:20-54)
    def SeedRand(self, seed):
        self._seed = int(seed)
        if self._grand is not None:
            # cMathUtil::SeedRand (util/MathUtil.cpp:114-118): gRand.Seed(seed); srand(gRand.RandInt()) -- the int argument widens to unsigned long
            self._grand.seed(int(seed))
            self._grand.rand_int()

    def ParseArgs(self, args):
        args = [str(a) for a in args]
        data_root = os.environ.get("DM_DATA_ROOT", ".")
        p = _model.ArgParser(args)
        af = p.str("arg_file", "")
        if af and len(args) == 2 and not os.path.exists(af if os.path.isabs(af) else os.path.join(data_root, af)) and af in _model.ARG_FILE_ASSETS:
            # the reference's data files are not here (e.g. the GPU box): the in-tree compiled copy of exactly this arg file (tools/compile_assets.py)
            self._tables = _model.load_asset(_model.ARG_FILE_ASSETS[af])
            self._num_update_substeps = int(self._tables.cfg.num_update_substeps)
            return
        self._tables = _model.load_scene_from_args(args, data_root=data_root)
        if af:
            p.load_file(af if os.path.isabs(af) else os.path.join(data_root, af))
        self._num_update_substeps = p.int("num_update_substeps", 1)      # DeepMimicCore.cpp:43

    def LoadTables(self, tables, num_update_substeps=1):
        """Extension: hand over already-parsed scene tables (e.g. ``deepmimic_amd.model.load_asset``) instead of arg files."""
        self._tables = tables
        self._num_update_substeps = int(num_update_substeps)

    def _kin_only(self):
        """`--scene kin_char` (scenes/SceneKinChar.cpp): the viewer's motion playback -- a kinematic character on the host, no agent, no device"""
        return self._tables is not None and self._tables.cfg.scene == "kin_char"

    def GetKinPose(self):
        """Extension (kin_char): the pose the reference would draw at the current time (cKinCharacter::GetPose)"""
        return [float(x) for x in self._kin.pose(self._time, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0))]

    def Init(self):
        if self._tables is None:
            raise RuntimeError("cDeepMimicCore.Init(): ParseArgs() has not been called")
        if self._kin_only():
            self._kin = _model.KinSampler(self._tables); self._time = 0.0
            self.stats = {"launches": 0, "updates": 0, "rollbacks": 0}
            return
        self._env = None
        if os.environ.get("DM_FACADE_SHARED", "0") == "1":
            # W worker processes behind ONE context and one launch per control step (deepmimic_amd/broker.py): the reference's `mpiexec -n W` deployment
            from deepmimic_amd.broker import SharedEnv
            try:
                self._env = SharedEnv(self._tables, seed=self._seed, device_id=int(os.environ.get("DM_DEVICE", "0")),
                                      precision=int(os.environ.get("DM_PRECISION", "32")), lib_path=os.environ.get("DM_HIP_LIB"), physics=int(os.environ.get("DM_PHYSICS", "1")))
            except NotImplementedError as ex:
                import warnings
                warnings.warn("DM_FACADE_SHARED: %s; this worker uses a context of its own" % ex, RuntimeWarning, stacklevel=2)
        if self._env is None:
            self._env = _BatchEnv(self._tables, 1, device_id=int(os.environ.get("DM_DEVICE", "0")), seed=self._seed,
                                  precision=int(os.environ.get("DM_PRECISION", "32")), lib_path=os.environ.get("DM_HIP_LIB"),
                                  physics=int(os.environ.get("DM_PHYSICS", "1")))
        self._off = self._env.offsets_scales()
        self._apply_mode()
        self._batch = os.environ.get("DM_FACADE_BATCH", "1") != "0"      # (DM-physics v2 too since round 4: the snapshot carries the ground manifolds)
        self._period = 1.0 / float(self._tables.query_rate)
        self.stats = {"launches": 0, "updates": 0, "rollbacks": 0}     # kernel launches of the stepping path vs Update() calls
        self._build_time_warper()
        self._ref_active = False
        self._tape = False                 # the device-side draws come from the reference's generators too (the draw tape, include/dm_hip.h DM_TAPE_*)
        self._tape_key = None
        c = self._tables.cfg
        goal_row = bool(self._tables.goal_kind != 0 or self._tables.num_clips > 1 or c.enable_rand_rot_reset)      # (core.BatchEnv._has_goal_row)
        if self._ref_rng:
            # the draws the device makes come off the draw tape of this worker's generators, on a context of its own and on a slot of the shared owner alike
            self._tape = goal_row or bool(c.enable_rand_perturbs)
            self._ref_init_draws()
        else:
            self._after_reset()

    # ---- the reference's draw order on cMathUtil::gRand (DM_RNG=reference) ------------------------------------------------------------
    def _ref_draw_timer(self, sample_count):
        """one cTimer::Reset (util/Timer.cpp:55-73) with the annealed parameters cTimer holds whatever the mode (RLSceneSimChar.cpp:338-347)"""
        c = self._tables.cfg
        lo, hi = _model.timer_limits(c, False, sample_count)
        if c.timer_type == "exp":
            return min(lo + self._grand.rand_exp(1.0 / _model.timer_exp(c, False, sample_count)), hi)
        return self._grand.rand_double(lo, hi)                     # (no draw when min == max: every shipped imitate arg file)

    def _ref_init_draws(self):
        """cDeepMimicCore::Init -> SetupScene (DeepMimicCore.cpp:635-660).  On gRand unless noted, in order: cScene::cScene seeds the scene's own
        generator mRand with one RandUint (scenes/Scene.cpp:5); cSceneImitate::Init builds the kinematic controller first (SceneImitate.cpp:153-161) -- a
        cClipsController selects its first clip (anim/ClipsController.cpp:24-34); cRLSceneSimChar::Init runs cScene::Init twice (through cRLScene::Init and
        cSceneSimChar::Init, RLSceneSimChar.cpp:27-31), each = InitTimers (one cTimer::Reset, Scene.cpp:124-127) + ResetParams, which for this class is two
        ResetTimers (RLSceneSimChar.cpp:234-238) -- six timer draws with the un-annealed parameters; cSceneSimChar::Init resets the perturbation clock
        (mRand, SceneSimChar.cpp:110-113); cGround::cGround takes one RandUint (sim/Ground.cpp:68); then the task scene's own Init (_ref_init_task_draws).
        The env is left at clip time 0 with the last limit drawn; the driver resets before it steps (learning/rl_world.py)."""
        self._ref_active = True
        self._srand = _RefRand(self._grand.rand_uint(), lib_path=os.environ.get("DM_HIP_LIB"))     # cScene::mRand (RandUint returns int: a negative value widens to unsigned long like in the reference)
        c = self._tables.cfg
        clip0 = 0
        if self._tape and c.kin_ctrl == "clips":
            clip0 = self._ref_select_clip()
        mt = np.inf
        for _ in range(6):
            mt = self._ref_draw_timer(0)
        if self._tape and c.enable_rand_perturbs:
            self._srand.rand_double(c.perturb_time_min, c.perturb_time_max)      # ResetRandPertrub; between the two cScene::Init's timer triples in call order, but on the other generator
        self._grand.rand_uint()
        if self._tape:
            self._ref_init_task_draws()
        self._after_reset(kin_time=0.0, max_time=mt, init=True)
        if self._tape and self._env._has_goal_row:
            self._env.set_clips([clip0])

    def _ref_select_clip(self):
        """cClipsController::SelectNewMotion (anim/ClipsController.cpp:226-243): upper_bound of one gRand draw in the weight CDF"""
        _, cdf = self._env.clip_table()
        return int(np.searchsorted(np.asarray(cdf), self._grand.rand_double(0.0, 1.0), side="right"))

    def _ref_init_task_draws(self):
        """What the Init of the task scenes draws after cSceneImitate::Init, for the positions of the two generators (every value is overwritten by the Reset
        the driver issues before it steps).  cSceneTargetAMP::Init (SceneTargetAMP.cpp:118-123): InitTarget = mTargetTimer.Init -> one cTimer::Reset (gRand),
        ResetTarget (virtual) -- target_amp: SampleRandTargetPos, two mRand draws (:274-285); heading_amp[_getup]: that + the target speed (mRand,
        SceneHeadingAMP.cpp:230-239; the get-up timers have min == max: no draw); strike_amp: far/near coin (mRand), three cMathUtil draws (SceneStrikeAMP.cpp:
        318-374), ResetTargetHit's coin (gRand, train mode and init_hit_prob > 0, :376-383) and the hit time when hit (mRand, :301-316; cScene::GetTime() is 0);
        dribble_amp (SceneDribbleAMP.cpp:151-160, 389-394): cSceneTargetAMP::Init as above with its own SampleRandTargetPos (:506-520, two mRand draws), then
        InitTarObjs = object timer Init (gRand) + ResetTarObjs (mRand r, theta; SetTarObjPos: three mRand axis draws, one gRand angle, :422-504), then
        mTargetTimer.Reset (gRand) and ResetTarget again.  RandDouble draws nothing when min == max (util/Rand.cpp:30-41)."""
        c, g, m = self._tables.cfg, self._grand, self._srand
        scene = c.scene
        if scene not in _model.GOAL_SCENES:
            return
        pi = float(np.pi)
        smin = c.tar_speed if c.tar_speed_min is None else c.tar_speed_min
        smax = c.tar_speed if c.tar_speed_max is None else c.tar_speed_max

        def reset_target():
            if scene == "strike_amp":
                far = m.rand_double(0.0, 1.0) < c.tar_far_prob
                g.rand_double(-pi, pi) if far else g.rand_double(c.target_min[0], c.target_max[0])
                g.rand_double(c.target_min[1], c.target_max[1])
                g.rand_double(c.target_min[2], c.max_target_dist) if far else g.rand_double(c.target_min[2], c.target_max[2])
                hit = False
                if self._mode == self.eModeTrain and c.init_hit_prob > 0.0:
                    hit = g.rand_double(0.0, 1.0) < c.init_hit_prob
                if hit:
                    m.rand_double(0.0 - c.target_hit_reset_time, 0.0)
            elif scene == "dribble_amp":
                m.rand_double(c.ball_radius, c.max_target_dist); m.rand_double(-pi, pi)
            else:
                m.rand_double(0.0, c.max_target_dist); m.rand_double(0.0, 2 * pi)
                if scene in ("heading_amp", "heading_amp_getup"):
                    m.rand_double(smin, smax)

        g.rand_double(c.rand_target_time_min, c.rand_target_time_max)          # InitTarget
        reset_target()
        if scene == "dribble_amp":
            g.rand_double(c.rand_tar_obj_time_min, c.rand_tar_obj_time_max)    # mTarObjTimer.Init
            m.rand_double(c.min_tar_obj_dist, c.max_tar_obj_dist); m.rand_double(-pi, pi)
            for _ in range(3):
                m.rand_double(-1.0, 1.0)
            g.rand_double(-pi, pi)
            g.rand_double(c.rand_target_time_min, c.rand_target_time_max)      # mTargetTimer.Reset
            reset_target()

    # ---- the draw tape: the device-side draws of a launch in the reference's order (include/dm_hip.h DM_TAPE_*)
    def _tape_fill(self):
        if not self._tape:
            return
        from deepmimic_amd.core import TAPE_K, TAPE_HDR, TAPE_STRIDE
        c = self._tables.cfg
        lo, hi = _model.timer_limits(c, False, self._sample_count)
        hdr = (1.0 if c.kin_ctrl == "clips" else 0.0, float(lo), float(hi), float(_model.timer_exp(c, False, self._sample_count)) if c.timer_type == "exp" else 0.0,
               float(_model.timer_limits(c, True, self._sample_count)[1]) if self._mode == self.eModeTest else -1.0)
        key = (self._grand.state(), self._srand.state(), hdr)
        if key == self._tape_key:
            return                            # nothing was drawn since the tape on the device was written
        T = np.zeros(TAPE_STRIDE)
        T[2], T[3] = key[1][1], key[1][2]
        T[5:10] = hdr
        ug, eg, _, _ = self._grand.tape()
        um, _, n3, i2 = self._srand.tape(normal=True, integer=True)
        K = TAPE_K
        T[TAPE_HDR:TAPE_HDR + K] = ug; T[TAPE_HDR + K:TAPE_HDR + 2 * K] = eg; T[TAPE_HDR + 2 * K:TAPE_HDR + 3 * K] = um
        T[TAPE_HDR + 3 * K:TAPE_HDR + 6 * K] = n3; T[TAPE_HDR + 6 * K:TAPE_HDR + 8 * K] = i2
        self._env.set_draw_tape(T)
        self._tape_key = key

    def _tape_commit(self):
        if not self._tape:
            return
        h = self._env.draw_tape_state()[0]
        if h[4] != 0.0:
            self._tape_key = None
            raise _TapeExhausted("cDeepMimicCore: a launch consumed more random draws than the draw tape holds (DM_TAPE_K raw values per generator)")
        if h[0] != 0.0 or h[1] != 0.0 or (int(h[2]), float(h[3])) != self._tape_key[1][1:]:
            self._grand.discard(int(h[0])); self._srand.discard(int(h[1])); self._srand.set_norm_state(int(h[2]), float(h[3]))
            self._tape_key = None

    def _ref_reset_draws(self):
        """cScene::Reset -> cRLSceneSimChar::ResetScene (RLSceneSimChar.cpp:240-244): cRLScene::ResetScene and cSceneSimChar::ResetScene each run
        ResetParams = two ResetTimers -> four cTimer::Reset draws, the last one stands (test mode pins it to time_end_lim_max afterwards,
        :277-284); then ResetCharacters -> cSceneImitate::ResetKinChar -> CalcRandKinResetTime = RandDouble(0, duration) (SceneImitate.cpp:320-343,
        494-500).  (Random perturbations and enable_rand_rot_reset draw from the SCENE generator in the reference; their device-side draws keep the
        counter-based streams.)"""
        mt = np.inf
        for _ in range(4):
            mt = self._ref_draw_timer(self._sample_count)
        if self._mode == self.eModeTest:
            mt = _model.timer_limits(self._tables.cfg, True, self._sample_count)[1]
        kt = self._grand.rand_double(0.0, float(self._env.duration))
        return kt, mt

    def _apply_mode(self):
        lo, hi = _model.timer_limits(self._tables.cfg, self._mode == self.eModeTest, self._sample_count)
        self._env.set_time_limits(lo, hi, _model.timer_exp(self._tables.cfg, self._mode == self.eModeTest, self._sample_count))   # (exp: --timer_type exp)
        if self._goal_size():
            self._env.set_mode(self._mode == self.eModeTest)       # get-up on a fall / recovery episodes / strike_amp's test reward

    def _need_env(self):
        if self._env is None:
            raise RuntimeError("cDeepMimicCore: Init() has not been called")
        return self._env

    # ---- test-mode time-warp score of `--scene imitate_amp` (cSceneImitateAMP::BuildTimeWarper / UpdateTimeWarper /
    # CalcRewardTimeWarp, scenes/SceneImitateAMP.cpp:173-205,417-480): an evaluation-only return, computed on the host from the
    # env state at every action boundary; the stepping kernels do not know about it
    def _build_time_warper(self):
        self._tw = None
        c = self._tables.cfg
        if c.scene != "imitate_amp" or not getattr(c, "enable_test_time_warp", True):
            return
        ends = [x for x in (c.time_lim_max, c.time_end_lim_max) if x is not None]
        max_time = max(ends) if ends else np.inf
        if not np.isfinite(max_time):
            return                                                   # BuildTimeWarper only builds it for a finite episode length
        # (multi-clip datasets since round 4: the kinematic side is the clip the env was reset to, cClipsController's active motion)
        self._tw = {"size": int(np.ceil(self._tables.query_rate * max_time)) + 2, "sim": [], "kin": [], "samplers": {}}

    def _tw_sample(self, st=None):
        if self._tw is None or self._mode != self.eModeTest:
            return
        if len(self._tw["sim"]) >= self._tw["size"]:
            raise RuntimeError("Time warper buffer overflow, capacity: %d" % self._tw["size"])   # cDynamicTimeWarper::AddSample0
        st = self._env.get_state() if st is None else st
        kin = st["kin"][0]

        def data(pose):                                              # BuildTimeWarpData (:462-480)
            jp = _model.joint_world_positions(self._tables, pose)
            out = jp - jp[0]
            out[0] = (0.0, jp[0][1], 0.0)
            return out.reshape(-1)
        self._tw["sim"].append(data(st["pose"][0]))
        clip = int(self._env.get_clips()[0]) if self._tables.num_clips > 1 else 0
        if clip not in self._tw["samplers"]:
            self._tw["samplers"][clip] = _model.KinSampler(self._tables, clip)
        self._tw["kin"].append(data(self._tw["samplers"][clip].pose(float(st["clocks"][0][0]), kin[0:3], kin[3:7])))

    def _time_warp_reward(self):
        tw = self._tw
        if not tw["sim"]:
            return 0.0
        cost = _model.time_warp_cost(np.array(tw["sim"]), np.array(tw["kin"]))
        return cost + (tw["size"] - len(tw["sim"])) * 1.0           # term_step_cost = 1 per step the episode fell short

    # ---- stepping engine ------------------------------------------------------------------------------------------------
    def _after_reset(self, kin_time=None, max_time=None, init=False):
        tape_reset = kin_time is None and self._tape and self._env._has_goal_row
        if kin_time is None and getattr(self, "_ref_active", False) and not tape_reset:
            kin_time, max_time = self._ref_reset_draws()
        if tape_reset:
            self._tape_fill()                  # timers, perturbation clock, clip time, clip, yaw and goal state come off the tape inside the reset kernel
            self._env.reset()
            self._tape_commit()
        elif kin_time is None:
            self._env.reset()
        elif self._tape and not init:          # (Init leaves the env at clip time 0 with no tape bound yet: what the Init of the scene drew is accounted for on the host)
            self._tape_fill()                  # (perturbation clock)
            self._env.reset(kin_times=[float(kin_time)], max_times=[float(max_time)])
            self._tape_commit()
        else:
            self._env.reset(kin_times=[float(kin_time)], max_times=[float(max_time)])
        if self._tw is not None:
            self._tw["sim"], self._tw["kin"] = [], []
            self._tw_sample()                                        # ResetTimeWarper -> UpdateTimeWarper
        self._sync_clocks()
        self._need = True                  # cDeepMimicCharController::Reset leaves NeedNewAction() == true
        self._pending = None               # action handed over by SetAction, applied by the next launch
        self._spec = None                  # outcome of a whole-control-step launch being consumed update by update
        self._cache = None

    def _sync_clocks(self):
        c = self._env.get_state()["clocks"][0]
        self._clk = {"ctrl": float(c[1]), "off": float(c[2]), "timer": float(c[3])}

    def _advance_clocks(self, dt):
        """host mirror of one update: timers += dt, then cCtController::CheckNeedNewAction (CtController.cpp:221-227)"""
        self._clk["ctrl"] += dt; self._clk["timer"] += dt
        cur, pad = self._clk["ctrl"] + self._clk["off"], 0.001 * dt
        self._need = int(np.floor((cur + pad) / self._period)) != int(np.floor((cur + pad - dt) / self._period))

    def _updates_to_next_action(self, dt, cap=64):
        ctrl, off = self._clk["ctrl"], self._clk["off"]
        for k in range(1, cap + 1):
            ctrl += dt
            cur, pad = ctrl + off, 0.001 * dt
            if int(np.floor((cur + pad) / self._period)) != int(np.floor((cur + pad - dt) / self._period)):
                return k
        return cap

    def _launch(self, action, dt, n, end_early):
        self.stats["launches"] += 1
        self._tape_fill()
        out = self._env.step(action, dt, n, end_early=end_early)
        self._tape_commit()
        return out

    def _virtual(self):
        """inside a consumed-in-advance control step, before its last executed update"""
        return self._spec is not None and self._spec["v"] < self._spec["n_done"]

    def _materialize(self):
        """A caller needs the env AT an intermediate update of a batched control step: roll back to the snapshot taken before the
        launch and replay the updates one by one (exactly what the unbatched path would have run)."""
        sp = self._spec
        self._spec = None
        self.stats["rollbacks"] += 1
        snap = sp["snap"]
        self._env.restore(snap)              # character, clocks; goal scenes: target / timers / draw counter / get-up or hit state; the ball
        if sp.get("rng") is not None:        # the reference's generators as they were before the launch: the replay draws the same numbers again
            self._grand.set_state(sp["rng"][0]); self._srand.set_state(sp["rng"][1]); self._tape_key = None
        self._clk = dict(sp["clk0"])
        out = None
        for i in range(sp["v"]):
            out = self._launch(sp["action"] if i == 0 else None, sp["dt"], 1, False)
            self._advance_clocks(sp["dt"])
        self._cache = out

    def _query(self):
        if self._virtual():
            self._materialize()
        if self._cache is None:
            self._cache = self._need_env().query()
        return self._cache

    # ---- stepping (DeepMimicCore.cpp:56-65)
    def Update(self, timestep):
        if self._kin_only():
            self._time += float(timestep)      # cSceneKinChar::Update -> cKinCharacter::Update
            return
        env = self._need_env()
        dt = float(timestep)
        self.stats["updates"] += 1
        if not dt > 0:
            return                           # cImpPDController::UpdateControlForce ignores non-positive steps
        sp = self._spec
        if sp is not None:
            if sp["v"] < sp["n_done"] and dt == sp["dt"]:
                sp["v"] += 1                 # this update already ran on the device
                self._advance_clocks(dt)
                self._cache = sp["out"] if sp["v"] == sp["n_done"] else None
                return
            if sp["v"] < sp["n_done"]:
                self._materialize()          # a different timestep mid-step
            self._spec = None
        action, self._pending = self._pending, None
        snap = None
        if self._need and self._tw is not None and self._mode == self.eModeTest:
            snap = env.get_state()
            self._tw_sample(snap)                # cSceneImitateAMP::NewActionUpdate -> UpdateTimeWarper (:140-150)
        k = self._updates_to_next_action(dt) if (self._batch and action is not None) else 1
        if k > 1:
            snap = env.snapshot()
            clk0 = dict(self._clk)
            rng0 = (self._grand.state(), self._srand.state()) if self._tape else None
            try:
                out = self._launch(action, dt, k, True)
            except _TapeExhausted:
                # more draws inside one control step than a tape holds (very short target / perturbation timers): back to the state before the launch and
                # through this control step one update at a time -- a tape serves any single update
                env.restore(snap)
                self._grand.set_state(rng0[0]); self._srand.set_state(rng0[1]); self._tape_key = None
                self.stats["tape_fallbacks"] = self.stats.get("tape_fallbacks", 0) + 1
                self._cache = self._launch(action, dt, 1, False)
                self._advance_clocks(dt)
                return
            t1 = float(out["clocks"][0][3]) if "clocks" in out else float(env.get_state()["clocks"][0][3])      # (the shared route returns the clocks with the step: one round trip)
            n_done = min(k, int(round((t1 - clk0["timer"]) / dt)))
            if n_done <= 0:
                # the launch ended before its first update: the state was already invalid at entry (the caller kept updating after CheckValidEpisode
                # returned false).  The reference's Update has no such test -- it steps; so does a single update without the early end (ADVICE r3:
                # assuming one update had run desynchronised the host clock mirror from the device)
                self._cache = self._launch(None, dt, 1, False)
                self._advance_clocks(dt)
                return
            self._spec = {"snap": snap, "clk0": clk0, "rng": rng0, "action": action, "dt": dt, "k": k, "n_done": n_done, "v": 1, "out": out}
            self._advance_clocks(dt)
            self._cache = out if n_done == 1 else None
        else:
            self._cache = self._launch(action, dt, 1, False)
            self._advance_clocks(dt)

    def Reset(self):
        if self._kin_only():
            self._time = 0.0                   # cKinCharacter::Reset
            return
        self._need_env()
        self._spec = None
        self._after_reset()

    def GetTime(self):
        if self._kin_only():
            return float(self._time)           # cSceneKinChar::GetTime: the character's clock
        self._need_env()
        return float(self._clk["timer"])      # cScene::GetTime: the episode timer (host mirror, no device round trip)

    def GetName(self):
        # cSceneImitate::GetName (scenes/SceneImitate.cpp:207-210), cSceneImitateAMP::GetName (SceneImitateAMP.cpp:208-211)
        scene = self._tables.cfg.scene if self._tables is not None else "imitate"
        if scene == "kin_char":
            return "Kinematic Char"
        return {"imitate_amp": "Imitate AMP", "target_amp": "Target AMP", "heading_amp": "Heading AMP", "heading_amp_getup": "Heading AMP Getup",
                "strike_amp": "Strike AMP", "dribble_amp": "Dribble AMP"}.get(scene, "Imitate")

    def _is_amp(self):
        return self._tables is not None and self._tables.cfg.scene in _model.AMP_SCENES

    def _goal_size(self):
        return int(self._tables.goal_dim) if self._tables is not None else 0

    def EnableDraw(self):
        return False

    # ---- rendering / UI: no-ops on the headless path (DeepMimicCore.cpp:82-163)
    def Draw(self):
        pass

    def Keyboard(self, key, x, y):
        pass

    def MouseClick(self, button, state, x, y):
        pass

    def MouseMove(self, x, y):
        pass

    def Reshape(self, w, h):
        pass

    def Shutdown(self):
        if self._env is not None:
            self._env.close()
            self._env = None

    def IsDone(self):
        return False

    def GetDrawScene(self):
        return None

    def SetPlaybackSpeed(self, speed):
        self._playback_speed = float(speed)

    def SetUpdatesPerSec(self, updates_per_sec):
        self._updates_per_sec = float(updates_per_sec)

    def GetWinWidth(self):
        return 0

    def GetWinHeight(self):
        return 0

    def GetNumUpdateSubsteps(self):
        return int(self._num_update_substeps)

    # ---- RL interface (DeepMimicCore.cpp:165-480)
    def IsRLScene(self):
        return not self._kin_only()            # cDeepMimicCore::IsRLScene: mRLScene != nullptr (DeepMimicCore.cpp:165-168)

    def GetNumAgents(self):
        return 0 if self._kin_only() else 1

    def _chk_agent(self, agent_id):
        if self._kin_only():
            raise RuntimeError("`--scene kin_char` has no agents (cDeepMimicCore::IsRLScene() is false)")
        if int(agent_id) != 0:
            raise RuntimeError("agent id %r out of range (the imitate scene has one agent)" % (agent_id,))

    def NeedNewAction(self, agent_id):
        self._chk_agent(agent_id)
        self._need_env()
        return bool(self._need)

    def RecordState(self, agent_id):
        self._chk_agent(agent_id)
        return [float(x) for x in self._query()["state"][0]]

    def RecordGoal(self, agent_id):
        self._chk_agent(agent_id)
        if not self._goal_size():
            return []               # cRLSceneSimChar::RecordGoal: goal size 0 (scenes/RLSceneSimChar.cpp:64-68,88-91)
        # cSceneTargetAMP::RecordGoal (SceneTargetAMP.cpp:195-223) / cSceneHeadingAMP::RecordGoal (SceneHeadingAMP.cpp:150-166)
        return [float(x) for x in self._query()["goal"][0]]

    def SetAction(self, agent_id, action):
        self._chk_agent(agent_id)
        a = np.asarray(action, dtype=np.float32).reshape(1, -1)
        if a.shape[1] != self._env.A:
            raise RuntimeError("SetAction: expected %d values, got %d" % (self._env.A, a.shape[1]))
        if self._virtual():
            self._materialize()             # an action mid-step: leave the batched step at this update
        self._spec = None
        self._pending = a                   # applied by the launch of the next Update (cDeepMimicCharController::ApplyAction happens before it)

    def LogVal(self, agent_id, val):
        pass

    def GetActionSpace(self, agent_id):
        return 1                    # eActionSpaceContinuous (env/action_space.py)

    def GetStateSize(self, agent_id):
        return int(self._need_env().S)

    def GetGoalSize(self, agent_id):
        return self._goal_size()

    def GetActionSize(self, agent_id):
        return int(self._need_env().A)

    def GetNumActions(self, agent_id):
        return 0

    def BuildStateOffset(self, agent_id):
        return [float(x) for x in self._off["state_offset"]]

    def BuildStateScale(self, agent_id):
        return [float(x) for x in self._off["state_scale"]]

    def _is_scene(self, name):
        return self._tables is not None and self._tables.cfg.scene == name

    def BuildGoalOffset(self, agent_id):
        off = [0.0] * self._goal_size()         # cRLSceneSimChar::BuildGoalOffsetScale (scenes/RLSceneSimChar.cpp:111-116)
        if self._is_scene("heading_amp_getup"):
            off[3] = -0.5                       # cSceneHeadingAMPGetup::BuildGoalOffsetScale (SceneHeadingAMPGetup.cpp:134-142): the get-up phase
        return off

    def BuildGoalScale(self, agent_id):
        sc = [1.0] * self._goal_size()
        if self._is_scene("heading_amp_getup"):
            sc[3] = 2.0
        return sc

    def BuildActionOffset(self, agent_id):
        return [float(x) for x in self._off["action_offset"]]

    def BuildActionScale(self, agent_id):
        return [float(x) for x in self._off["action_scale"]]

    def BuildActionBoundMin(self, agent_id):
        return [float(x) for x in self._off["action_min"]]

    def BuildActionBoundMax(self, agent_id):
        return [float(x) for x in self._off["action_max"]]

    def BuildStateNormGroups(self, agent_id):
        return [int(x) for x in self._off["state_norm_groups"]]

    def BuildGoalNormGroups(self, agent_id):
        g = [0] * self._goal_size()             # cCharController::gNormGroupSingle (:136-140)
        if self._is_scene("heading_amp_getup"):
            g[3] = -1                           # gNormGroupNone (SceneHeadingAMPGetup.cpp:144-150)
        if self._is_scene("strike_amp"):
            g = [-1] * len(g)                   # cSceneStrikeAMP::BuildGoalNormGroups (SceneStrikeAMP.cpp:408-412)
        return g

    def CalcReward(self, agent_id):
        self._chk_agent(agent_id)
        if self._tw is not None and self._mode == self.eModeTest:
            # cSceneImitateAMP::CalcReward -> CalcRewardTimeWarp: 0 until the episode is over, then the alignment cost
            return self._time_warp_reward() if self.IsEpisodeEnd() else 0.0
        return float(self._query()["reward"][0])

    def GetRewardMin(self, agent_id):
        return 0.0                  # sim/DeepMimicCharController.cpp:200-208

    def GetRewardMax(self, agent_id):
        return 1.0

    def GetRewardFail(self, agent_id):
        return 0.0                  # scenes/RLScene.cpp:26-34

    def GetRewardSucc(self, agent_id):
        return 1.0

    def EnableAMPTaskReward(self):
        return self._goal_size() > 0            # cSceneTargetAMP::EnableAMPTaskReward (SceneTargetAMP.cpp:230-233)

    # ---- AMP observations (DeepMimicCore.h:75-81 -> scenes/SceneImitateAMP.cpp); empty for a plain imitate scene
    def GetAMPObsSize(self):
        return int(self._need_env().amp_size)

    def GetAMPObsOffset(self):
        return [0.0] * self.GetAMPObsSize()            # SceneImitateAMP.cpp:88-91

    def GetAMPObsScale(self):
        return [1.0] * self.GetAMPObsSize()            # :93-96

    def GetAMPObsNormGroup(self):
        return [0] * self.GetAMPObsSize()              # :98-101  cCharController::gNormGroupSingle = 0

    def RecordAMPObsAgent(self, agent_id):
        if not self._is_amp():
            return []
        self._chk_agent(agent_id)
        if self._virtual():
            self._materialize()
        return [float(x) for x in self._need_env().query_amp()[0]]

    def RecordAMPObsExpert(self, agent_id):
        """One expert sample at a random clip time (the reference draws it from the scene RNG; here from the ctx's
        counter-based generator), ground height = the kin character's origin height (SceneImitateAMP.cpp:115-138)."""
        if not self._is_amp():
            return []
        self._chk_agent(agent_id)
        env = self._need_env()
        if self._virtual():
            self._materialize()
        gh = float(env.get_state()["kin"][0][1])
        if self._tape and self._tables.cfg.kin_ctrl == "clips":      # SampleExpertMotion (:260-277): SampleMotionID draws the clip from gRand, then the time from mRand
            clip = self._ref_select_clip()
            dur, _ = env.clip_table()
            t = self._srand.rand_double(0.0, float(dur[clip]))
            if self._tables.num_clips > 1:
                return [float(x) for x in env.amp_expert_clips(1, [clip], [t], gh)[0]]
            return [float(x) for x in env.amp_expert(1, [t], gh)[0]]
        if self._tables.num_clips > 1:           # SampleExpertMotion with a cClipsController: clip by weight, time within that clip
            return [float(x) for x in env.amp_expert_clips(1, None, None, gh)[0]]
        if getattr(self, "_ref_active", False):  # mRand.RandDouble(0, motion_duration) on the scene's generator (SceneImitateAMP.cpp:119)
            return [float(x) for x in env.amp_expert(1, [self._srand.rand_double(0.0, float(env.duration))], gh)[0]]
        return [float(x) for x in env.amp_expert(1, None, gh)[0]]

    def IsEpisodeEnd(self):
        if self._kin_only():
            return False                    # not an RL scene (DeepMimicCore.cpp:467-475)
        if self._virtual():
            return False                    # the batched launch stops at the update where the episode is over: not yet
        return bool(self._query()["episode_end"][0])

    def CheckValidEpisode(self):
        if self._kin_only():
            return True
        if self._virtual():
            return True                     # (the batched launch stops before the update that would start from an invalid state: not yet)
        return bool(self._query()["valid"][0])

    def CheckTerminate(self, agent_id):
        self._chk_agent(agent_id)
        if self._virtual():
            return 0                        # eTerminateNull
        return int(self._query()["terminate"][0])

    def SetMode(self, mode):
        self._mode = int(mode)
        if self._env is not None:
            self._apply_mode()

    def SetSampleCount(self, count):
        """cRLSceneSimChar::SetSampleCount (scenes/RLSceneSimChar.cpp:223-227): anneals the episode-length limits."""
        self._sample_count = int(count)
        if self._env is not None:
            self._apply_mode()
