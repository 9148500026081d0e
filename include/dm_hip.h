/* dm_hip.h -- C-ABI of the MI355X-native DeepMimic imitate environment (libdm_hip.so).
 *
 * This is the drop-in boundary for the reference's SWIG class `cDeepMimicCore`
 * (/root/reference/DeepMimicCore/DeepMimicCore.h:9-119, exported by DeepMimicCore.i:1-34) restricted to
 * the `--scene imitate` hot path.  One `dm_ctx` owns N independent envs (N = 1 reproduces one
 * cDeepMimicCore instance) on one GPU and one HIP stream.  Plain pointers and sizes only; every function
 * returns 0 on success and a negative code on error, `dm_last_error()` gives the message.  The library
 * never aborts the caller (the reference asserts: DeepMimicCore.cpp:36-40, scenes/SceneBuilder.cpp:60-64).
 *
 * Array arguments are HOST pointers unless the name ends in `_dev` / the flag DM_DEVICE_PTRS is given.
 */
#ifndef DM_HIP_H
#define DM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Layout version of this header: bumped whenever dm_create_info / dm_scene_tables gain fields or an entry point changes meaning.  A host checks
 * `dm_abi_version() == DM_ABI_VERSION` (and a binding that mirrors the structs by hand, `dm_struct_sizes`) once after loading the library: the
 * tables are read as the CURRENT layout, a caller built against an older header would hand over a shorter struct. */
#define DM_ABI_VERSION 5

typedef struct dm_ctx dm_ctx;

typedef struct {
    int num_envs;            /* N independent characters */
    int device_id;           /* HIP device ordinal */
    uint64_t seed;           /* cDeepMimicCore::SeedRand (DeepMimicCore.cpp:20-23); keys the per-env reset RNG */
    int precision;           /* 32 (default, production) or 64 (algorithm-parity build of the same kernels) */
    int max_contacts;        /* manifold cap per character, <= 20 (0 -> 20) */
    int env_id_offset;       /* global id of env 0 (multi-GPU shards keep global RNG streams) */
    int wave_packing;        /* characters per wavefront of the step kernel: 0 = default (2 where possible; env DM_DUO=0 -> 1), 1, or
                                2 (biped class, even num_envs; same results up to fp rounding, dm_device_duo.h) */
    int physics;             /* rigid-body step: 0 / 1 = DM-physics v1 (default: analytic contact set rebuilt every substep, the nearer
                                row of a revolute limit), 2 = v2 (Bullet's manifold semantics as recalled, SURVEY App. C items 4, 7:
                                both rows of every revolute limit; one persistent manifold per link against the ground, refreshed and
                                given ONE new support point per substep, <= 4 points, breaking threshold 0.02 x angular-motion disc).
                                v2 has its own instantiations of the kernels (AMP code + manifolds), one or -- biped class, since round 4 -- two
                                characters per wavefront; not with the dribble ball; DESIGN.md 4.6 */
} dm_create_info;

/* Raw scene tables in the reference's in-memory layout (all host pointers, copied at create time). */
typedef struct {
    int num_joints;
    const double* joint_mat;     /* J x 19  cKinTree::tJointDesc rows   (anim/KinTree.h:24-47)  */
    const double* body_defs;     /* J x 17  cKinTree::tBodyDef rows     (anim/KinTree.h:49-70)  */
    const double* pd_params;     /* J x 2   Kp, Kd                      (sim/PDController.cpp:50-93) */
    int num_frames;
    const double* frames;        /* F x (1+P) duration + pose           (anim/Motion.cpp:344-378) */
    int loop;                    /* "Loop": "wrap"                      (anim/Motion.cpp:302-320) */
    const int32_t* fall_mask;    /* J  fall-contact links               (scenes/SceneSimChar.cpp:460-476) */
    /* arg-file keys */
    int num_sim_substeps;        /* --num_sim_substeps */
    double world_scale;          /* --world_scale (only scales Bullet's absolute tolerances) */
    double gravity[3];           /* --gravity, default (0,-9.8,0) */
    int sync_char_root_pos, sync_char_root_rot, enable_fall_end, enable_char_contact_fall;
    int enable_root_rot_fail, enable_rand_char_placement, enable_rand_rot_reset;
    double time_lim_min, time_lim_max;   /* episode timer, uniform (util/Timer.cpp:55-73); inf = no limit */
    /* controller file keys (sim/CtController.cpp:161-172) */
    int enable_phase_input, record_world_root_pos, record_world_root_rot;
    double query_rate;
    /* DM-physics v1 constants; 0 selects the default */
    double friction, erp; int solver_iters;
    int disable_self_collision;  /* 0 (default): links of the character collide with each other except parent-child pairs, as
                                    btMultiBody's default m_hasSelfCollision does (sim/SimCharacter.cpp:857,873,919) */
    /* `--scene imitate_amp` (scenes/SceneImitateAMP.cpp): CalcReward = 0, CheckTerminate = fall only, AMP observations on */
    int scene_amp;
    int enable_amp_obs_local_root;   /* --enable_amp_obs_local_root (SceneImitateAMP.cpp:30,42) */
    /* ---- goal-conditioned AMP task scenes, scene_amp must be set:
     * 1 = `--scene target_amp` (scenes/SceneTargetAMP.cpp), 2 = `--scene heading_amp` (scenes/SceneHeadingAMP.cpp).  RecordGoal has
     * size 3, CalcReward is the task reward, CheckTerminate adds the target-distance failure; keys of ParseArgs (:92-106 / :55-84) */
    int scene_goal;
    double rand_target_time_min, rand_target_time_max, max_target_dist, target_succ_dist, tar_fail_dist, tar_speed;
    int enable_min_tar_vel;
    double pos_reward_scale, max_heading_turn_rate, sharp_turn_prob, speed_change_prob, tar_speed_min, tar_speed_max, vel_reward_scale;
    /* ---- multi-clip dataset (`--kin_ctrl clips`, anim/ClipsController.cpp): `frames` is the concatenation of the clips, clip c owns
     * rows clip_starts[c] .. clip_starts[c+1]-1; a reset draws the clip by weight (SelectNewMotion, :226-243).  0 = single clip. */
    int num_clips;
    const int32_t* clip_starts;   /* num_clips + 1 */
    const double* clip_weights;   /* num_clips */
    const int32_t* clip_loops;    /* num_clips */
    /* ---- scene_goal 3 = `--scene heading_amp_getup` (scenes/SceneHeadingAMPGetup.cpp): heading_amp + a get-up timer -- RecordGoal has
     * size 4 (+ get-up phase, :123-130), the get-up reward while it runs (:4-38), no contact fall while it runs (:255-264), in test mode
     * a fall starts it (:244-253), in train mode a failed episode continues as a recovery episode with probability
     * recover_episode_prob (:109-121, 40-56).  getup_time = duration of the longest get-up clip (CalcGetupTime :266-291);
     * getup_clip_mask: bit c set = clip c is a get-up motion (--getup_motion_ids).
     * scene_goal 4 = `--scene strike_amp` (scenes/SceneStrikeAMP.cpp): target_amp with a target point in the air, RecordGoal size 4
     * (target in the origin frame + hit phase, :414-434), near / far / hit reward (:23-187), success and target-contact termination
     * (:485-541); strike_mask / fail_tar_mask: bit j = link j (--strike_bodies / --fail_tar_contact_bodies). */
    int mode_test;                /* 1 = cRLScene::eModeTest at creation; dm_set_mode changes it */
    double getup_time, getup_height_root, getup_height_head, recover_episode_prob;
    int head_id, getup_clip_mask;
    double tar_near_dist, tar_far_prob, target_radius, target_hit_reset_time, init_hit_prob, hit_tar_speed, tar_reward_scale;
    double target_min[3], target_max[3];
    int strike_mask, fail_tar_mask;
    /* ---- scene_goal 5 = `--scene dribble_amp` (scenes/SceneDribbleAMP.cpp): target_amp with a ball -- a free rigid sphere in the world
     * (BuildTarObjs :398-420: radius, mass 0.43, friction 0.4, linear / angular damping 0.4) the character has to move to the target.
     * RecordState gains 15 entries (the ball in the origin frame, :554-590), RecordGoal is the direction / distance from the ball to the
     * target (:276-302), the reward :21-106, success / distance termination :343-379, 454-475.  One character per wavefront.
     * ball_friction is the combined coefficient of a ball contact (ball 0.4 x link / ground 0.9). */
    double rand_tar_obj_time_min, rand_tar_obj_time_max, min_tar_obj_dist, max_tar_obj_dist;
    double ball_radius, ball_mass, ball_friction, ball_lin_damping, ball_ang_damping;
    /* ---- random perturbations (`--enable_rand_perturbs`, scenes/SceneSimChar.cpp:41-51, 92-99, 205-256, 618-626, 952-956; sim/Perturb.cpp;
     * sim/PerturbManager.cpp; sim/World.cpp:93-96): every U[perturb_time_min, perturb_time_max] seconds a force of U[min_perturb,
     * max_perturb] N in a uniformly drawn direction acts at the centre of mass of a random body part for U[min, max duration] seconds
     * (`min_pertrub_duration` [sic] / `max_perturb_duration`).  perturb_part_mask: bit j = part j may be hit (--perturb_part_ids), 0 = any.
     * Cleared and re-armed by every scene reset.  Needs 2 * perturb_time_min >= max_perturb_duration (two forces at once at most). */
    int enable_rand_perturbs;
    double perturb_time_min, perturb_time_max, min_perturb, max_perturb, min_perturb_duration, max_perturb_duration;
    int perturb_part_mask;
} dm_scene_tables;

/* DM_END_EPISODE_EARLY: an env whose episode is over after update u of the call (fall contact, clip end, episode timer) takes no
 * further updates in this call -- the reference's driver checks IsEpisodeEnd after EVERY Update and ends the episode there
 * (DeepMimic.py:62-80), so the terminal reward / state are those of that moment, not of the next action boundary. */
enum { DM_DEVICE_PTRS = 1, DM_AUTO_RESET = 2, DM_OPEN_LOOP = 4, DM_NO_EMIT = 8, DM_END_EPISODE_EARLY = 16 };

const char* dm_last_error(void);
/* 0 for libdm_hip.so; 1 for the CPU fiber-emulator build of the same sources (tests/emu, test infrastructure) */
int dm_is_emulator(void);
/* DM_ABI_VERSION the library was built with; out[0] = sizeof(dm_create_info), out[1] = sizeof(dm_scene_tables) as the library sees them */
int dm_abi_version(void);
int dm_struct_sizes(int32_t* out);

/* cDeepMimicCore ctor + ParseArgs + Init  (DeepMimicCore.cpp:9-54) */
int dm_create(const dm_create_info* info, const dm_scene_tables* tables, dm_ctx** out);
int dm_destroy(dm_ctx* ctx);
/* GetStateSize / GetGoalSize / GetActionSize (+ pose, links, dofs, frames of clip 0): out[8] = S,G,A,P,J,D,F,N */
int dm_dims(const dm_ctx* ctx, int32_t* out);
/* out[0] = physics version the ctx runs (1 or 2), out[1] = effective contact cap per character (physics 2: min(max_contacts, (64 - 2 x limited joints) / 3)) */
int dm_physics_info(const dm_ctx* ctx, int32_t* out);
double dm_motion_duration(const dm_ctx* ctx);
/* Run the kernels of this ctx on an external HIP stream (e.g. torch's current stream); NULL restores the own (non-blocking) stream.
 * The legacy default stream has the NULL handle too: dm_set_stream_default selects IT (torch's default stream when no torch.cuda.Stream is
 * current), so that work enqueued by the caller on the default stream and this ctx's launches are ordered against each other. */
int dm_set_stream(dm_ctx* ctx, void* hip_stream);
int dm_set_stream_default(dm_ctx* ctx);
/* The HIP stream handles of the ctx: *out_own = the non-blocking stream dm_create made for it, *out_current = the stream its kernels are
 * launched on right now (own stream, an external one, or NULL = the legacy default stream).  Either pointer may be NULL.  For callers that
 * order other work against the ctx with events (torch.cuda.ExternalStream(own), hipStreamWaitEvent): env groups on their own streams whose
 * records feed one collective (bench.py, deepmimic_amd/groups.py). */
int dm_get_stream(const dm_ctx* ctx, void** out_own, void** out_current);
int dm_synchronize(dm_ctx* ctx);

/* cDeepMimicCore::SetMode (DeepMimicCore.cpp:472-479 -> cRLSceneSimChar::SetMode / ResetTimers, scenes/RLSceneSimChar.cpp:
 * 270-290): the episode-timer range drawn at the next resets; train = (time_lim_min, time_lim_max), test = time_end_lim_*. */
int dm_set_time_limits(dm_ctx* ctx, double time_lim_min, double time_lim_max);
/* `--timer_type exp --time_lim_exp E` (scenes/Scene.cpp:19-23, util/Timer.cpp:27-45, 64-67): every reset -- explicit or inside a DM_AUTO_RESET
 * launch -- then draws min(time_lim_min + Exp(mean E), time_lim_max) instead of U[time_lim_min, time_lim_max].  0 (the default) = uniform timer.
 * Annealing blends E like the limits (cTimer::tParams::Blend): call again with the blended value. */
int dm_set_timer_exp(dm_ctx* ctx, double time_lim_exp);

/* cDeepMimicCore::Reset (DeepMimicCore.cpp:61-65).  env_ids NULL -> all envs.  kin_times / max_times NULL ->
 * per-env counter-based RNG: kin time ~ U[0,duration) (scenes/SceneImitate.cpp:494-500), timer ~ U[min,max]. */
int dm_reset(dm_ctx* ctx, const int32_t* env_ids, int n, const double* kin_times, const double* max_times);
/* cDeepMimicCore::SetAction for every env (DeepMimicCore.cpp:221-230): actions N x A float32 */
int dm_set_action(dm_ctx* ctx, const float* actions, int flags);
/* cDeepMimicCore::Update(timestep) n_updates times for every env (DeepMimicCore.cpp:56-59) */
int dm_update(dm_ctx* ctx, double timestep, int n_updates);
/* RecordState / CalcReward / CheckTerminate / CheckValidEpisode / IsEpisodeEnd / NeedNewAction for every env
 * (DeepMimicCore.cpp:191-204,450-458,...); any output may be NULL */
int dm_query(dm_ctx* ctx, float* states, float* rewards, int32_t* terminate, int32_t* valid, int32_t* episode_end,
             int32_t* need_new_action, int flags);
/* Batched control step = [SetAction] + n_updates x Update + query (+ auto reset), one kernel launch.
 * actions may be NULL (keep the latched targets).  With DM_AUTO_RESET, envs whose episode ended are reset after
 * their terminal reward/flags are written and `states` holds the first observation of the new episode. */
int dm_step_batch(dm_ctx* ctx, const float* actions, double timestep, int n_updates, float* states, float* rewards,
                  int32_t* terminate, int32_t* valid, int32_t* episode_end, int flags);

/* dm_step_batch for a SUBSET of the ctx's envs: env_ids[n] distinct ids; actions (n x A, NULL = keep the latched targets), states (n x S), rewards,
 * flags, amp_obs (n x dm_amp_obs_size(), imitate_amp scenes) and clocks (n x 5 doubles: kin_time, ctrl_time, init_time_offset, timer_time, timer_max
 * after the call) are COMPACT host arrays, row i <-> env_ids[i]; any output may be NULL.  The other envs are not touched.  One character per
 * wavefront (the kernel of an odd-sized / one-env ctx: an env stepped here follows the trajectory it would follow alone in a ctx of its own).  This is
 * what a process that owns the GPU for several one-env callers runs when some of them have asked for their next control step and others have not
 * (deepmimic_amd/broker.py: W cDeepMimicCore workers -- the reference's `mpiexec -n W`, mpi_run.py:16-24 -- behind ONE launch per control step). */
int dm_step_envs(dm_ctx* ctx, const int32_t* env_ids, int n, const float* actions, double timestep, int n_updates, float* states, float* rewards,
                 int32_t* terminate, int32_t* valid, int32_t* episode_end, float* amp_obs, double* clocks, int flags);

/* ---- `--scene imitate_amp` only (dm_scene_tables.scene_amp): adversarial-motion-prior observations
 * GetAMPObsSize (scenes/SceneImitateAMP.cpp:76-86): 2 x (pose features + velocity features); 0 for a plain imitate scene */
int dm_amp_obs_size(const dm_ctx* ctx);
/* RecordAMPObsAgent for every env (:101-113): [pose_t, pose_t-1, vel_t, vel_t-1] of the simulated character, t-1 = the
 * state at the last action latch (cSceneImitateAMP::UpdateHist, :166-171).  amp_obs N x dm_amp_obs_size() float32. */
int dm_query_amp(dm_ctx* ctx, float* amp_obs, int flags);
/* dm_step_batch that also writes RecordAMPObsAgent at the end of the control step (before any auto reset, i.e. the
 * end-of-path observation of a finished episode, learning/amp_agent.py:239-242). */
int dm_step_batch_amp(dm_ctx* ctx, const float* actions, double timestep, int n_updates, float* states, float* rewards,
                      int32_t* terminate, int32_t* valid, int32_t* episode_end, float* amp_obs, int flags);
/* RecordAMPObsExpert (:115-138), n samples: clip frames at times[i] and one control period earlier (raw cMotion::CalcFrame /
 * CalcFrameVel).  times NULL -> drawn ~ U[0, duration) from the ctx's counter-based generator (the reference draws from the
 * scene RNG); ground_h NULL -> 0 (the reference passes the kin character's origin height).  out n x dm_amp_obs_size(). */
int dm_amp_expert(dm_ctx* ctx, int n, const double* times, const double* ground_h, float* out, int flags);
/* Multi-clip variant (SampleExpertMotion, SceneImitateAMP.cpp:115-138 with a cClipsController): clips[i] = dataset clip of sample i;
 * clips NULL -> drawn by weight (and times ~ U[0, that clip's duration)) from the ctx generator.  Host pointers. */
int dm_amp_expert_clips(dm_ctx* ctx, int n, const int32_t* clips, const double* times, const double* ground_h, float* out);

/* ---- goal scenes (dm_scene_tables.scene_goal != 0)
 * RecordGoal for every env (SceneTargetAMP.cpp:195-223 / SceneHeadingAMP.cpp:150-166 / SceneHeadingAMPGetup.cpp:123-130 /
 * SceneStrikeAMP.cpp:414-434): goals N x dm_goal_size() float32 */
int dm_goal_size(const dm_ctx* ctx);          /* GetGoalSize: 0 (no goal scene), 3, or 4 (heading_amp_getup, strike_amp) */
int dm_query_goal(dm_ctx* ctx, float* goals, int flags);
/* the goals written by the most recent dm_step_batch / dm_query (no launch): what the agent reads next to `states` */
int dm_last_goals(dm_ctx* ctx, float* goals);
/* RecordGoal of the last emit into a DEVICE buffer (N x goal size floats), asynchronously on the ctx stream: no host round trip */
int dm_last_goals_device(dm_ctx* ctx, float* goals_dev);
/* Goal state snapshot (tests, checkpointing): N x 12 doubles = target pos(3), target heading, target speed, target timer time,
 * target timer max, COM at the last action(3), controller time of the last action, draws consumed so far.  NULL = leave as is. */
int dm_get_goal_state(dm_ctx* ctx, double* out);
int dm_set_goal_state(dm_ctx* ctx, const double* in);
/* the scene-specific part of the goal state, N x 8 doubles: [0..1] heading_amp_getup {get-up timer time, unused}; strike_amp {target hit
 * (0 / 1), scene time of the hit (-1 = none)}; [2..6] dribble_amp {ball position at the last action (3), target-object timer time, limit};
 * [7] reserved: reads 0, is ignored on write (the goal row's next slot is the env's own draw key, dm_set_env_keys, which a restored aux block never re-keys) */
int dm_get_goal_aux(dm_ctx* ctx, double* out);
int dm_set_goal_aux(dm_ctx* ctx, const double* in);
/* the random-perturbation state (enable_rand_perturbs), N x 16 doubles per env: [0] time since the last perturbation, [1] time of the next one
 * (cSceneSimChar::tPerturbParams::mTimer / mNextTime), [2] draw counter, then two force slots of 6: {body part + 1 (0 = free), force x, y, z,
 * duration, elapsed} (tPerturb, sim/Perturb.h), [15] unused.  For checkpointing and for tests that place a known force. */
int dm_get_perturb_state(dm_ctx* ctx, double* out);
int dm_set_perturb_state(dm_ctx* ctx, const double* in);
/* cRLScene::SetMode (DeepMimicCore.cpp SetMode -> scene): 0 train, 1 test.  Only the goal scenes read it on the device (get-up on a
 * fall instead of termination, recovery episodes, strike_amp's test reward); the episode-timer limits of the two modes are the
 * caller's business (dm_set_time_limits). */
int dm_set_mode(dm_ctx* ctx, int test_mode);
/* Give the listed envs a draw key of their own: env_ids[i] becomes what env 0 of a fresh ONE-env context created with seed seeds[i] (< 2^53) is before its first
 * reset -- every counter-based draw it makes from then on (clip, clip time, yaw, episode limit, goal re-sampling, perturbations) is keyed (seeds[i], env 0), its draw
 * and episode counters start at 0, its DM-physics v2 manifolds are empty -- so that W one-env callers can share one context and one launch per control step and
 * still see the trajectories of W private contexts (the shared-owner facade, deepmimic_amd/broker.py; the reference's deployment is one cDeepMimicCore per MPI
 * worker, mpi_run.py:16-24).  Reset the envs next (dm_reset), as dm_create does.  Synchronises the ctx stream. */
int dm_set_env_keys(dm_ctx* ctx, const int32_t* env_ids, int n, const uint64_t* seeds);
/* durations[k], cdf[k] of the num_clips clips of a `--kin_ctrl clips` dataset (cClipsController::mClipsCDF, anim/ClipsController.cpp; one entry -- the clip's
 * duration, 1 -- for a single-clip scene): what dm_amp_expert_clips draws its clips and clip times from */
int dm_clip_table(const dm_ctx* ctx, double* durations, double* cdf);
/* dribble_amp: the ball of every env, N x 13 doubles = position(3), rotation w x y z (4), linear velocity(3), angular velocity(3)
 * (cSimObj::GetPos / GetRotation / GetLinearVelocity / GetAngularVelocity of the target object) */
int dm_get_obj_state(dm_ctx* ctx, double* out);
int dm_set_obj_state(dm_ctx* ctx, const double* in);
/* DM-physics v2 (dm_create_info.physics = 2): the persistent link-vs-ground manifolds, N x J x 25 doubles per link = {point count (0..4),
 * 4 x (the point on the link in body coordinates (3), the point on the plane x, z, distance)} -- device state that dm_get_state / dm_set_state do
 * not carry (dm_set_state(pose) EMPTIES them, as a fresh pose has no contact history).  A checkpoint or a rollback under v2 is
 * dm_get_state + dm_get_manifolds / dm_set_state THEN dm_set_manifolds.  Error when the ctx runs v1. */
int dm_get_manifolds(dm_ctx* ctx, double* out);
int dm_set_manifolds(dm_ctx* ctx, const double* in);
/* clip each env's kinematic character was reset to (multi-clip datasets), N int32 */
int dm_get_clips(dm_ctx* ctx, int32_t* out);
/* Set that bookkeeping (N int32): the clip the kinematic controller counts as active -- the reference draws the clip time of a reset over the duration of the
 * clip that was active BEFORE the reset (scenes/SceneImitate.cpp:331-335, 494-500), and cClipsController::Init already selected one at construction
 * (anim/ClipsController.cpp:24-34).  The character's state is not touched; only a reset through the draw tape reads it. */
int dm_set_clips(dm_ctx* ctx, const int32_t* clips);

/* BuildStateOffset/Scale, BuildActionOffset/Scale/BoundMin/BoundMax, BuildStateNormGroups (DeepMimicCore.cpp:232-448) */
int dm_build_offsets_scales(const dm_ctx* ctx, double* s_off, double* s_scale, double* a_off, double* a_scale,
                            double* a_min, double* a_max, int32_t* s_norm_groups);

/* Env state snapshot (parity tests, checkpointing): pose/vel N x P, kin N x 7 (origin pos + rot), clocks N x 5
 * (kin_time, ctrl_time, init_time_offset, timer_time, timer_max), tar N x P.  Any pointer may be NULL. */
int dm_get_state(dm_ctx* ctx, double* pose, double* vel, double* tar, double* kin, double* clocks, int32_t* flags);
int dm_set_state(dm_ctx* ctx, const double* pose, const double* vel, const double* tar, const double* kin,
                 const double* clocks, const int32_t* flags);

/* Component taps used by the parity tests (no reference analogue): what = 0 SPD torque, 1 one rigid-body
 * substep of length `dt` with the latched torque, 2 SPD-model mass matrix / bias force.
 * After the call dm_get_debug copies the requested tap: name in {"H","C","vstar","lambda","rows","tau",
 * "kin_pose","kin_vel","reward_terms","links"}; out must hold the full N x ... array of doubles.
 * "tau" and "fallback" need no dm_probe.  "fallback" (N doubles): the number of rigid-body substeps the env has spent on
 * the 64-lane fallback of the two-characters-per-wavefront kernel (a pair of which one character had more than 32
 * constraint rows in that substep; both characters count it) since dm_create or the last dm_set_state -- a statistic
 * of the production kernels, kept in the pad word of the env's kin row; 0 on the one-character-per-wavefront kernels.
 * "borrowed" (N doubles): likewise the substeps in which ONE character of the pair had more than 32 rows, the two together
 * at most 64, and the pair stayed on the two-per-wavefront path with the heavy character's rows 32.. on lanes of its
 * partner's half (kept in the pad word of the env's clock row). */
int dm_probe(dm_ctx* ctx, int what, double dt);
int dm_set_tau(dm_ctx* ctx, const double* tau /* N x D, generalized-velocity layout */);
int dm_get_debug(dm_ctx* ctx, const char* name, double* out);

/* Fixed-action rollout timed with HIP events on the ctx stream: `steps` control steps of `n_updates` updates
 * after `warmup` untimed ones.  actions_dev NULL with DM_OPEN_LOOP tracks the clip.  Returns the elapsed GPU
 * milliseconds of the timed region in *elapsed_ms. */
int dm_bench_rollout(dm_ctx* ctx, int warmup, int steps, double timestep, int n_updates, int flags,
                     float* states_dev, float* rewards_dev, double* elapsed_ms);

/* ---- Multi-GPU: one process per GPU, env shards are independent; the only exchange is the all-gather of the
 * learner record {state[S], reward, terminate} once per control step, RCCL over xGMI.  For hosts that stay in C/C++ (the
 * reference's own launcher is `mpiexec -n W`, mpi_run.py:16-24); deepmimic_amd/dist.py is the torch.distributed mirror.
 * Bootstrap is the caller's: rank 0 calls dm_comm_unique_id, ships the 128 bytes to the other ranks by its own means (MPI_Bcast,
 * a file, a torch store) and every rank calls dm_comm_create (= ncclCommInitRank).  librccl is loaded lazily (dlopen) by these
 * calls only; world == 1 with a NULL unique id needs no RCCL at all (the gather is a device copy). */
typedef struct dm_comm dm_comm;
int dm_comm_unique_id(void* out_128_bytes);
int dm_comm_create(const void* unique_id_128_bytes, int world, int rank, int device_id, dm_comm** out);
int dm_comm_destroy(dm_comm* comm);
/* All-gather `count` floats per rank: recv_dev[r * count ..] = rank r's send_dev.  Enqueued on the comm's own stream, ordered after
 * everything already enqueued on the ctx stream (an event, no host sync), so the caller can launch control step k+1 on the ctx
 * stream right away: the collective overlaps it.  `slot` in [0, 4) names the event pair of this buffer (double buffering). */
int dm_gather_records(dm_ctx* ctx, dm_comm* comm, int slot, const float* send_dev, float* recv_dev, size_t count);
/* Make the ctx stream (not the host) wait for the gather last launched on `slot`; no-op when none is in flight. */
int dm_gather_wait(dm_ctx* ctx, dm_comm* comm, int slot);

/* ---- The reference's random generator, for a host that replays the reference's draw order (the N = 1 drop-in route)
 * cRand (util/Rand.h, util/Rand.cpp:6-135) behind cMathUtil::gRand (util/MathUtil.cpp:6,61-134): std::default_random_engine +
 * the <random> distributions, state kept across calls.  Implemented with the same standard-library types, so a caller that issues
 * the reference's calls in the reference's order -- cDeepMimicCore::SeedRand (DeepMimicCore.cpp:20-23 -> cMathUtil::SeedRand:
 * Seed, then one RandInt for srand), cScene::cScene (scenes/Scene.cpp:5: one RandUint), cTimer::Reset (util/Timer.cpp:55-73),
 * cGround::cGround (sim/Ground.cpp:68: one RandUint), cSceneImitate::CalcRandKinResetTime (scenes/SceneImitate.cpp:494-500) -- gets
 * the reference's reset times and episode limits for the same seed, to hand to dm_reset(kin_times, max_times).  Host only; the
 * batched device path draws from counter-based streams keyed by (seed, global env id, episode) instead. */
typedef struct dm_refrand dm_refrand;
int dm_refrand_create(unsigned long seed, dm_refrand** out);          /* cRand(seed) */
int dm_refrand_destroy(dm_refrand* r);
int dm_refrand_seed(dm_refrand* r, unsigned long seed);               /* cRand::Seed */
double dm_refrand_double(dm_refrand* r, double min, double max);      /* cRand::RandDouble(min, max): min when min == max, no draw */
double dm_refrand_exp(dm_refrand* r, double lambda);                  /* cRand::RandDoubleExp */
double dm_refrand_norm(dm_refrand* r, double mean, double stdev);     /* cRand::RandDoubleNorm */
int dm_refrand_int(dm_refrand* r);                                    /* cRand::RandInt() */
int dm_refrand_int_range(dm_refrand* r, int min, int max);            /* cRand::RandInt(min, max) */
int dm_refrand_uint(dm_refrand* r);                                   /* cRand::RandUint() */
int dm_refrand_discard(dm_refrand* r, long n);                        /* advance the engine by n raw values */
int dm_refrand_norm_state(dm_refrand* r, int set, int* avail, double* saved);    /* the second deviate std::normal_distribution keeps for its next call: read (set = 0) or write */
int dm_refrand_engine_state(dm_refrand* r, int set, unsigned long* state);       /* the engine's state (minstd_rand0: one integer): read or write -- for snapshots */

/* ---- The draw tape: the reference's draw ORDER for the draws that happen on the device (one-env drop-in, `DM_RNG=reference`).
 * The reference draws from two generators -- cMathUtil::gRand ("engine 0": every cTimer::Reset (util/Timer.cpp:55-73), cClipsController::SelectNewMotion
 * (anim/ClipsController.cpp:226-243), cSceneImitate::CalcRandKinResetTime (scenes/SceneImitate.cpp:494-500), the target sampling of strike_amp
 * (scenes/SceneStrikeAMP.cpp:336-383), the ball's rotation angle (scenes/SceneDribbleAMP.cpp:498)) and the scene's own cScene::mRand ("engine 1": goal
 * re-sampling of the task scenes (scenes/SceneHeadingAMP.cpp:168-220, SceneTargetAMP.cpp:275-285, SceneDribbleAMP.cpp:430-520), the random yaw
 * (scenes/SceneImitate.cpp:346), perturbations (scenes/SceneSimChar.cpp:205-256, 952-956), the recovery coin (SceneHeadingAMPGetup.cpp:314)) -- in an order
 * that depends on what happens inside a launch (a target timer running out, a perturbation falling due).  dm_refrand_tape tabulates, for every raw
 * engine position k < DM_TAPE_K past a generator's current state, what each distribution would return if its next call started there (computed with
 * the <random> types themselves); the kernels look their draws up in the reference's call order and advance the two positions; afterwards the host reads
 * the positions back (dm_get_draw_tape_state) and discards that many raw values from its generators (dm_refrand_discard, dm_refrand_norm_state).
 *
 * Row of env e (DM_TAPE_STRIDE doubles): header [0] raw values consumed on engine 0, [1] on engine 1, [2] / [3] engine 1's saved normal deviate
 * (available flag, value), [4] error flag (a launch ran past the tables: raised by the device, the caller must treat the launch as failed),
 * [5] 1 when the kinematic controller is a cClipsController (every reset draws a clip, even out of one), [6..8] the episode timer's
 * {min, max, exp} as cTimer holds them (annealed train-mode parameters whatever the mode; exp <= 0: uniform type), [9] the limit test mode pins
 * after the draws (< 0: none), [10..15] unused; then the tables: engine 0 uniform [K], engine 0 -log(1 - uniform) [K], engine 1 uniform [K],
 * engine 1 normal {value, saved value, raw values consumed} [3 K], engine 1 |RandInt()| {value, raw values consumed} [2 K].
 * With a tape bound, dm_reset without clip times draws the whole reset from it (timers, perturbation clock, clip time, clip, yaw, goal state), and
 * DM_AUTO_RESET is refused.  Contexts without a tape (the batched path) keep their counter-based streams. */
#define DM_TAPE_K 96
#define DM_TAPE_HDR 16
#define DM_TAPE_STRIDE (DM_TAPE_HDR + 8 * DM_TAPE_K)
int dm_refrand_tape(const dm_refrand* r, int K, double* u, double* e, double* n3, double* i2);   /* any table may be null; the generator is not advanced */
int dm_set_draw_tape(dm_ctx* ctx, const double* tape /* N x DM_TAPE_STRIDE; NULL: unbind, back to the counter-based streams */);
int dm_get_draw_tape_state(dm_ctx* ctx, double* out /* N x DM_TAPE_HDR */);
/* the rows of the listed envs only (n x DM_TAPE_STRIDE in, n x DM_TAPE_HDR out): one context serving several one-env callers, each with generators of its own
 * (the shared-owner route, deepmimic_amd/broker.py); every env a launch steps or resets must hold a current row while a tape is bound.  n = 0: bind the
 * rows the device already holds (nothing was drawn from them since they went up). */
int dm_set_draw_tape_envs(dm_ctx* ctx, const int32_t* env_ids, int n, const double* rows);
int dm_get_draw_tape_state_envs(dm_ctx* ctx, const int32_t* env_ids, int n, double* out);

/* ---- Native scene loading: cDeepMimicCore::ParseArgs (DeepMimicCore.cpp:25-44) + the ParseArgs / file loading of the scene classes the path serves,
 * in C++ inside the library (deepmimic_amd/csrc/dm_scene_load.h), so that a native host goes from the reference's own arg file to a running context
 * without Python:   dm_scene_load(args, n, data_root, test_mode, &scene);  dm_create(&info, dm_scene_get_tables(scene), &ctx);
 * `args` are command-line style tokens ("--arg_file", "args/run_humanoid3d_walk_args.txt", or the keys themselves; the first occurrence of a key
 * wins, util/ArgParser.cpp:31-120); relative paths resolve against data_root (the reference resolves them against its working directory).
 * test_mode: the episode timer pinned to time_end_lim_max (scenes/RLSceneSimChar.cpp:277-284).  The tables belong to the scene handle.
 * dm_scene_info out[8] = num_update_substeps, anneal_samples, time_end_lim_min, time_end_lim_max, time_end_lim_exp, time_lim_exp, timer type (0 uniform,
 * 1 exp: call dm_set_timer_exp(ctx, time_lim_exp)), reserved -- what a driver needs beside the tables (SetSampleCount annealing, Update substeps). */
typedef struct dm_scene dm_scene;
int dm_scene_load(const char* const* args, int n_args, const char* data_root, int test_mode, dm_scene** out);
const dm_scene_tables* dm_scene_get_tables(const dm_scene* scene);
int dm_scene_info(const dm_scene* scene, double* out);
int dm_scene_free(dm_scene* scene);

/* ---- On-device policy inference: the actor of learning/pg_agent.py:141-188 with the net of
 * learning/nets/fc_2layers_1024units.py and the normalisers of learning/normalizer.py:95-102, on the matrix cores (bf16
 * operands, fp32 accumulate), so that observation -> action -> control step stays on the GPU.  Weights are fp32 host
 * arrays in tf.layers.dense layout (kernel [in x out] row-major, bias [out]); NULL normaliser / logstd arrays mean
 * identity / 0.  s_clip <= 0: no clipping. */
typedef struct dm_policy dm_policy;
typedef struct {
    int state_dim, hidden1, hidden2, action_dim;      /* reference: S, 1024, 512, A; hidden widths multiples of 64 */
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    const float *s_mean, *s_std;                      /* normalize:   (s - mean) / std, clipped to +-s_clip */
    const float *a_mean, *a_std;                      /* unnormalize: norm_a * std + mean */
    const float *logstd;                              /* Gaussian head (learning/tf_distribution_gaussian_diag.py), A entries */
    double s_clip;
} dm_policy_params;
int dm_policy_create(int device_id, const dm_policy_params* params, dm_policy** out);
int dm_policy_destroy(dm_policy* policy);
/* actions[n x A] (and logp[n], optional) for states[n x S]; all DEVICE pointers, asynchronous on hip_stream.
 * sample = 0: the mode (pg_agent.py _mode_a_tf); 1: mean + exp(logstd) * N(0,1) with Philox4x32-10 noise keyed by
 * (seed + env_id_offset + row, step * A + j) -- the generator of deepmimic_amd/streams.py. */
int dm_policy_forward(dm_policy* policy, const float* states_dev, int n, float* actions_dev, float* logp_dev, int sample,
                      uint64_t seed, uint32_t step, int env_id_offset, void* hip_stream);
/* _decide_action of learning/pg_agent.py:214-221 for a batch, with the goal as its own input block:
 *  - goals_dev [n x goal_dim] (RecordGoal): the net's input is the concatenation [state, goal] (pg_agent.py:170-187), so the policy's
 *    state_dim counts the goal columns too (w1 has state_dim rows, s_mean / s_std cover both blocks) and states_dev is n x (state_dim - goal_dim);
 *    goal_dim = 0: as dm_policy_forward;
 *  - exp_rate in [0, 1] (exp_params_curr.rate): with sample = 1 a row takes the sampled action with probability exp_rate (its own coin:
 *    Philox4x32-10 keyed like the noise, counter (step, 1, 0, 0), uniform < exp_rate) and the mode otherwise; logp is the one of the action
 *    taken; exp_flags_dev[n] (optional, int32) = 1 for the rows that explored (the EXP flag of a path's steps, pg_agent.py:251-255). */
int dm_policy_forward_ex(dm_policy* policy, const float* states_dev, const float* goals_dev, int goal_dim, int n, float* actions_dev,
                         float* logp_dev, int32_t* exp_flags_dev, double exp_rate, int sample, uint64_t seed, uint32_t step, int env_id_offset,
                         void* hip_stream);

/* ---- Running observation statistics on the device: the Normalizer of the reference's learner (learning/normalizer.py:6-152; the
 * s_norm / g_norm / amp_obs_norm of learning/rl_agent.py:466-483, amp_agent.py:290-291) for records that stay in HBM.
 * group_ids (NULL = one NORM_GROUP_SINGLE group): per column -1 = never updated (NORM_GROUP_NONE), 0 = per element, k > 0 = the
 * columns of group k share the average of the new data (normalizer.py:141-149; dm_build_offsets_scales hands the ids out).
 * eps = smallest std (reference: 0.02); clip <= 0 or inf = no clipping in dm_norm_normalize. */
typedef struct dm_normalizer dm_normalizer;
int dm_norm_create(int device_id, int size, const int32_t* group_ids, double eps, double clip, dm_normalizer** out);
int dm_norm_destroy(dm_normalizer* norm);
/* Normalizer.record (normalizer.py:33-45): new_count += n, new_sum += column sums, new_sum_sq += column sums of squares, in fp64, of
 * x[n x size] fp32 -- a DEVICE pointer with DM_DEVICE_PTRS in flags (the record block of a control step), a host pointer otherwise.
 * Asynchronous on hip_stream; deterministic (two passes, no atomics). */
int dm_norm_record(dm_normalizer* norm, const float* x, int n, int flags, void* hip_stream);
/* the pending sums {new_count, new_sum[size], new_sum_sq[size]} as ONE device array of 1 + 2 size doubles: what MPIUtil.reduce_sum
 * (normalizer.py:48-50) adds over the workers -- all-reduce it (SUM) over the ranks before dm_norm_update */
int dm_norm_pending(dm_normalizer* norm, double** dev_ptr, int* len);
/* Normalizer.update (normalizer.py:47-73): fold the pending sums into count / mean / mean_sq / std, clear them.  Asynchronous on hip_stream. */
int dm_norm_update(dm_normalizer* norm, void* hip_stream);
/* set_mean_std (normalizer.py:79-93; host arrays): mean_sq = std^2 + mean^2; count >= 0 also sets the sample count (TFNormalizer.load), < 0 keeps it */
int dm_norm_set(dm_normalizer* norm, const double* mean, const double* std, int64_t count, void* hip_stream);
/* host copies of the statistics (any pointer may be NULL); synchronises hip_stream */
int dm_norm_get(dm_normalizer* norm, double* mean, double* std, double* mean_sq, int64_t* count, void* hip_stream);
/* Normalizer.normalize (normalizer.py:95-98) of x[n x size] into out[n x size], DEVICE pointers, fp32 */
int dm_norm_normalize(dm_normalizer* norm, const float* x_dev, int n, float* out_dev, void* hip_stream);
/* make `norm` columns [first_column, first_column + size) of the observation normaliser of `policy` (a device-to-device copy of mean and 1 / std,
 * ordered on hip_stream): the actor's next dm_policy_forward on that stream normalises with the statistics of the last dm_norm_update / dm_norm_set.
 * s_norm at column 0 and g_norm at column state_size, as the reference keeps them (learning/rl_agent.py:212-222). */
int dm_policy_bind_obs_normalizer(dm_policy* policy, dm_normalizer* norm, int first_column, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
