// Device-visible tables and per-env state for the imitate hot path.
//
// One `ModelDev` describes one scene type (skeleton + PD gains + motion clip + config); it is built on
// the host by dm_host.cpp from the raw reference-layout tables and is shared by all envs.  The tables the
// inner loops touch are packed into one `MdlLds` block that every wavefront copies into LDS at kernel entry
// (one coalesced read of ~2 KB), so that no global-memory latency sits inside the 20-update loop.
// `EnvState` is the per-env dynamic state in HBM: one record per env, fields contiguous per env
// (a wave owns exactly one env, so "lane k reads field k of my env" is the coalesced pattern).
#pragma once
#include <stdint.h>

namespace dmk {

enum { JT_REVOLUTE = 0, JT_PLANAR, JT_PRISMATIC, JT_FIXED, JT_SPHERICAL, JT_NONE };
enum { SH_NULL = 0, SH_BOX, SH_CAPSULE, SH_SPHERE, SH_CYLINDER, SH_PLANE };
enum { DK_ROOT_LIN = 0, DK_ROOT_ANG = 1, DK_SPH = 2, DK_REV = 3 };
enum { TERM_NULL = 0, TERM_FAIL = 1, TERM_SUCC = 2 };

constexpr int kWave = 64;
constexpr int kMaxRows = 64;   // constraint rows per substep (one per lane)
constexpr int kMaxLim = 4;     // joint-limit rows (revolute joints with lo <= hi): knees / elbows of the shipped characters
constexpr int kMaxRotLinks = 4; // links with a non-identity body (or joint attach) rotation, ClsLarge
constexpr int kMaxContacts = 20;   // contact slots per character (ground + self): kMaxLim/NL + 3 * 20 <= kMaxRows

// Compiled kernel classes: static bounds of the per-lane register arrays and the LDS record.
struct ClsBiped {      // humanoid3d: 15 links, 34 dof, 43 pose dims, <= 64 ground-contact candidates, no attach rotations
    static constexpr int NJ = 15, ND = 34, NP = 43, NCAP = 64, RREG = 32, RREG_PLAIN = 32, NPAIRCAP = 128, LPAD = 4; static constexpr bool ROT = false;   // RREG: rows of A kept in VGPRs
    static constexpr bool OBJ = false;      // no free rigid body next to the character
    static constexpr bool TREE = false;     // dense LL^T factor (TREE classes: branch-sparse, level-scheduled L^T L on a compiled topology)
    static constexpr bool BROAD = false;    // self collision: every link pair goes through the segment-segment test (BROAD classes: bounding-sphere cull first)
    static constexpr bool PGS_MASKSEL = false;      // sweep: lane r takes its new lambda by v_cmp + v_cndmask (MASKSEL classes: a select on a constant SGPR lane mask)
    static constexpr bool GRAM64 = false;   // 64-row Gram matrix by the readlane loop (128-VGPR budget of the one-per-wave kernel)
    static constexpr int PFD = 2;           // look-ahead of the sweep into the overflow block of A, rows
    static constexpr bool FULLD = false;    // every character stepped through this class has exactly ND dofs (the fallback class of the two-per-wave kernel: dm_host.cpp checks D == ND before a duo launch)
    static constexpr int PRIO_FLOOR = 0;    // lowest wave priority inside substep_post (s_setprio; the fallback class of the two-per-wave kernel raises it)
};
// the same character class with all 64 rows of A in VGPRs: the instantiation the two-per-wave kernel falls back to for a pair with a
// heavily contacted character (256-VGPR budget there; identical LDS record layout)
struct ClsBipedWide : ClsBiped { static constexpr int RREG = 64, RREG_PLAIN = 64; };
// the fallback class of the two-per-wave kernel by default: the narrow row file (rows 32..63 of A in the HBM / L2 overflow block), but
// the Gram matrix of a character with more than 32 rows still comes off the matrix core (64 accumulators live for the Gram only)
#ifndef DM_FB_RREG
#define DM_FB_RREG 32
#endif
// DM_FB_PRIO: a pair on the fallback is a wave the launch will wait for (one round of waves lasts as long as its slowest): it runs the two 64-lane passes above its SIMD mate throughout
// DM_DUO_YFULL: the y = L^-1 J^T loops of the two-per-wave kernel (DuoSim::substep_post, duo_rows_xd, the 64-lane fallback class) without the per-dof `k < D` tests
// (every duo launch has D == ND)
#ifndef DM_DUO_YFULL
#define DM_DUO_YFULL 1
#endif
#ifndef DM_FB_PRIO
#define DM_FB_PRIO 0
#endif
struct ClsBipedFb : ClsBiped { static constexpr int RREG = DM_FB_RREG, RREG_PLAIN = DM_FB_RREG; static constexpr bool GRAM64 = true; static constexpr int PRIO_FLOOR = DM_FB_PRIO; static constexpr bool FULLD = DM_DUO_YFULL != 0; };   // (a look-ahead of 6 rows instead of 2 measured no gain)
// the biped class plus one free rigid sphere in the world (`--scene dribble_amp`: the ball, scenes/SceneDribbleAMP.cpp:398-420); one
// character per wavefront, 2 waves / SIMD (the ball's Jacobian columns ride in six more VGPRs per row lane)
struct ClsBipedObj : ClsBiped { static constexpr bool OBJ = true; static constexpr bool PGS_MASKSEL = true; };     // (same-box A/B of the mask select: -3.6 %; ClsBiped at 128 VGPRs: +2.8 %, SGPR pressure)
struct ClsLarge {      // dog3d and anything up to 23 links / 64 dof / 83 pose dims / 128 candidates, attach rotations allowed
    static constexpr int NJ = 23, ND = 64, NP = 83, NCAP = 128, RREG = 32, RREG_PLAIN = 32, NPAIRCAP = 256, LPAD = 2; static constexpr bool ROT = true;
    static constexpr bool GRAM64 = false; static constexpr int PFD = 2; static constexpr bool OBJ = false; static constexpr bool TREE = false;
    static constexpr bool BROAD = false; static constexpr bool PGS_MASKSEL = false; static constexpr int PRIO_FLOOR = 0; static constexpr bool FULLD = false;
};

// ---- compiled skeleton topologies (the elimination program of the branch-sparse factor is generated at compile time) -------------
// A kernel class with TREE = true factors the mass matrix as H = L^T L (Featherstone, "Efficient factorization of the joint-space
// inertia matrix for branched kinematic trees", IJRR 2005: no fill-in, L_ij != 0 only when dof j is an ancestor of dof i) instead of
// the dense L L^T of sim/ImpPDController.cpp:162-188 (Eigen LDLT) / Bullet's dense solves.  The dof tree -- the dofs of a joint form a
// chain, a joint's first dof hangs off its parent joint's last -- is a compile-time table, so that every register index, every skipped
// zero block and the LEVEL SCHEDULE (all pivots of one tree depth are eliminated in one step: 22 dependent steps instead of 64 columns
// for dog3d) are immediates in the instruction stream.  The host checks the loaded skeleton against the table (dm_host.cpp) and falls
// back to the dense class when it differs.
template <int N>
struct TopoTables {
    int par[N], depth[N];
    uint64_t anc[N], desc[N];          // strict ancestors / strict descendants of dof k
    int order[N];                      // dofs sorted by depth, deepest level first, ascending inside a level
    int lev_start[N + 1];              // order[lev_start[v] .. lev_start[v + 1]) is level v of that schedule (v = 0: the deepest)
    uint64_t levmask[N];               // the dofs of level v as a lane mask (lane = dof)
    // skyline store of the factor's columns: column j keeps rows 4 floor(j / 4) .. colend[j] - 1 (colend: past its last descendant, quad
    // aligned) at word colbase[j]; quadmask[t]: the lanes whose column holds rows 4t .. 4t + 3; lwords: words to allocate (with the slack
    // that keeps colbase[j] - 4 floor(j / 4) + 63 in range for the masked row reads)
    int colbase[N], colend[N], lwords;
    uint64_t quadmask[(N + 3) / 4];
    int nlev, maxw;                    // levels; widest level
};
template <int N>
constexpr TopoTables<N> make_topo(const int (&par)[N]) {
    TopoTables<N> t{};
    for (int k = 0; k < N; ++k) {
        t.par[k] = par[k];
        t.depth[k] = par[k] < 0 ? 0 : t.depth[par[k]] + 1;
        t.anc[k] = par[k] < 0 ? 0ull : (t.anc[par[k]] | (1ull << par[k]));
    }
    for (int k = 0; k < N; ++k) for (int i = 0; i < N; ++i) if ((t.anc[i] >> k) & 1ull) t.desc[k] |= 1ull << i;
    int md = 0;
    for (int k = 0; k < N; ++k) if (t.depth[k] > md) md = t.depth[k];
    t.nlev = md + 1; t.maxw = 0;
    int o = 0;
    for (int v = 0; v <= md; ++v) {
        t.lev_start[v] = o;
        for (int k = 0; k < N; ++k) if (t.depth[k] == md - v) { t.order[o++] = k; t.levmask[v] |= 1ull << k; }
        if (o - t.lev_start[v] > t.maxw) t.maxw = o - t.lev_start[v];
    }
    for (int v = md + 1; v <= N; ++v) t.lev_start[v] = o;
    int tot = 0; t.lwords = 0;
    for (int j = 0; j < N; ++j) {
        int last = j;
        for (int i = j + 1; i < N; ++i) if ((t.desc[j] >> i) & 1ull) last = i;
        const int q0 = 4 * (j / 4);
        t.colend[j] = (last + 4) / 4 * 4; t.colbase[j] = tot; tot += t.colend[j] - q0;
        if (t.colbase[j] - q0 + 64 > t.lwords) t.lwords = t.colbase[j] - q0 + 64;
        for (int q = j / 4; q < t.colend[j] / 4; ++q) t.quadmask[q] |= 1ull << j;
    }
    if (tot > t.lwords) t.lwords = tot;
    if (t.lwords < 8 * N) t.lwords = 8 * N;       // the momentum records of dyn_mom live in the same storage (8 words per dof)
    return t;
}
// data/characters/dog3d.txt: root 6 | spine0, spine1 | neck, head | two fore legs (shoulder 3, forearm 1, hand 3, finger 3) off spine1 |
// two hind legs (upper leg 3, leg 1, foot 3, toe 3) and the tail (3 + 3) off the root
struct TopoDog3d {
    static constexpr int N = 64;
    static constexpr int PAR[64] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 11, 18, 19, 20, 21, 22, 23, 24, 25, 26, 11, 28, 29, 30,
                                    31, 32, 33, 34, 35, 36, 5, 38, 39, 40, 41, 42, 43, 44, 45, 46, 5, 48, 49, 50, 51, 52, 53, 54, 55, 56, 5, 58, 59, 60, 61, 62};
    static constexpr TopoTables<64> T = make_topo<64>(PAR);
};
// data/characters/humanoid3d.txt: root 6 | chest (neck; right / left shoulder 3 + elbow 1) | right / left hip 3, knee 1, ankle 3 off the root
struct TopoHumanoid3d {
    static constexpr int N = 34;
    static constexpr int PAR[34] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 5, 12, 13, 14, 15, 16, 17, 8, 19, 20, 21, 5, 23, 24, 25, 26, 27, 28, 8, 30, 31, 32};
    static constexpr TopoTables<34> T = make_topo<34>(PAR);
};
// the biped class on humanoid3d's compiled topology (one character per wavefront: `wave_packing 1`; the two-per-wave kernel keeps the dense
// factor -- its 31 row lanes per character have no room for the root-translation columns, which an L^T L elimination finishes LAST)
struct ClsBipedTree : ClsBiped { static constexpr bool TREE = true; typedef TopoHumanoid3d Topo; };
// the large class on dog3d's compiled topology
// rows of A in VGPRs for the plain imitate instantiation of the compiled dog3d class.  25.6 % of the dog's substeps have more than 32 rows, 1.8 % more
// than 48, none more than 56 (profiles/r03_rows_hist.txt); 48 / 56 / 64 all compile to 0 scratch at 2 waves / SIMD (239 / 247 / 256 VGPRs) and measure
// +3.3 / +3.7 / +3.4 % over 32 on one box with bit-identical outputs (profiles/r04_ab_dog_rreg.json).  The AMP / v2 instantiations keep 32 (48 spills there).
#ifndef DM_DOG_RREG
#define DM_DOG_RREG 56
#endif
struct ClsLargeTree : ClsLarge { static constexpr int RREG_PLAIN = DM_DOG_RREG; static constexpr bool TREE = true; typedef TopoDog3d Topo; static constexpr bool GRAM64 = true; static constexpr bool BROAD = true; static constexpr bool PGS_MASKSEL = true; };   // (more than 32 rows: Gram on the matrix core too)

// link_info word: parent+1 [0:4] | jtype [5:7] | depth [8:11] | pose_off [12:18] | dof_off [19:25] | arot_ident 26 | brot_ident 27 | is_ee 28 | fall 29
#define DM_LI_PARENT(i) (((i) & 31) - 1)
#define DM_LI_JTYPE(i) (((i) >> 5) & 7)
#define DM_LI_DEPTH(i) (((i) >> 8) & 15)
#define DM_LI_POFF(i) (((i) >> 12) & 127)
#define DM_LI_DOFF(i) (((i) >> 19) & 127)
#define DM_LI_AROT_ID(i) (((i) >> 26) & 1)
#define DM_LI_BROT_ID(i) (((i) >> 27) & 1)
#define DM_LI_IS_EE(i) (((i) >> 28) & 1)
#define DM_LI_FALL(i) (((i) >> 29) & 1)
// dof_info word: joint [0:7] | kind [8:9] | axis [10:11] | vidx [12:19]
#define DM_DI_JOINT(i) ((i) & 255)
#define DM_DI_KIND(i) (((i) >> 8) & 3)
#define DM_DI_AXIS(i) (((i) >> 10) & 3)
#define DM_DI_VIDX(i) (((i) >> 12) & 255)

// Hot model tables, staged in LDS.  Plain-old-data with the same layout on host and device.
template <typename Real, typename C>
struct MdlLds {
    int link_info[C::NJ];
    uint32_t subtree_mask[C::NJ];              // bit k: link k is in the subtree rooted at j (incl. j)
    uint32_t chain_lo[C::NJ], chain_hi[C::NJ]; // dofs of the joints on the path root..j (incl.): support of a point Jacobian on link j
    Real attach[C::NJ][3];                     // joint attach point in the parent joint frame
    Real battach[C::NJ][3];                    // body COM in the joint frame
    Real mass[C::NJ];
    Real inertia[2][C::NJ][3];                 // principal inertias: [0] SPD model, [1] simulator model
    Real thresh[C::NJ], torque_lim[C::NJ];
    Real cap[C::NJ][4];                        // self-collision capsule of the link: half segment c0 (body frame, c1 = -c0), radius
    // rotations (ClsLarge only), compact: rot_idx[j] = (ordinal + 1 of the link's body rotation) | (same for its joint attach
    // rotation) << 4, 0 = identity; at most kMaxRotLinks links of either kind
    int rot_idx[C::ROT ? C::NJ : 1];
    Real attach_rot[C::ROT ? kMaxRotLinks : 1][C::ROT ? 9 : 1];
    Real brot[C::ROT ? kMaxRotLinks : 1][C::ROT ? 9 : 1];
    int dof_info[C::ND];
    Real kp[C::NJ], kd[C::NJ];                 // PD gains per joint (every dof of a joint shares them; root 0)
    int lim_joint[kMaxLim]; Real lim_lo[kMaxLim], lim_hi[kMaxLim];
};

template <typename Real>
struct ModelDev {
    int J, P, D, A, S, F, NC, NL, max_depth;
    const uint32_t* mdl_blob; int mdl_words;   // MdlLds<Real, C> image
    // ---- per link / joint tables only used outside the update loop (action latch, reward, reset)
    const int* act_off;
    const Real* diffw;
    const Real* aabb_he;                     // J x 4  half extents of the collider AABB box (w = 1: sphere)
    // ---- ground contact candidates, NC entries (each lane keeps its own candidates in registers)
    const int* cand_link; const Real* cand_loc /* NC x 3, body frame */; const Real* cand_rad;
    // ---- self-collision pairs (i < j, not parent-child), NPAIR entries i | j << 8, in (i, j) lexicographic order; 0 pairs = off
    const int* pair_code; int NPAIR;
    // ---- motion clip
    const double* frame_time;                // F
    const Real* frames;                      // F x P (post-processed)
    const Real* frame_vel;                   // F x P
    double duration; int loop; Real cycle_delta[3];
    // ---- config
    Real gravity[3];
    int num_sim_substeps, solver_iters, max_contacts;
    Real friction, erp, report_dist, max_lin_vel, max_ang_vel, slerp_one;
    Real lim_max_impulse;                    // upper bound of a joint-limit row's impulse: btMultiBodyConstraint::m_maxAppliedImpulse 100 in the scaled world = 100 / world_scale^2
    int sync_root_pos, sync_root_rot, enable_fall_end, enable_contact_fall, enable_root_rot_fail, enable_rand_placement;
    int enable_phase_input, record_world_root_pos, record_world_root_rot;
    double query_period;                     // 1 / QueryRate
    double time_lim_min, time_lim_max;       // episode timer range
    double timer_exp;                        // 0: uniform timer U[min, max]; > 0: `--timer_type exp`, min(min + Exp(mean timer_exp), max) (util/Timer.cpp:55-73)
    uint64_t seed;
    double* draw_tape;                       // N x TP_STRIDE or null: the reference's generators as position-indexed tables (dm_set_draw_tape; TP_* below)
    int env_off;                             // global id of env 0 of this shard (keeps RNG streams partition-invariant)
    int physics;                             // 1: DM-physics v1, 2: v2 (persistent ground manifolds, both rows of a revolute limit); NL counts ROWS
    int NLJ;                                 // revolute joints with limits (v1: NL == NLJ, v2: NL == 2 NLJ)
    // ---- `--scene imitate_amp` (scenes/SceneImitateAMP.cpp): reward 0, fall-only termination, AMP observations
    int scene_amp, amp_local_root;
    int amp_pose_size, amp_vel_size;         // per timestep; the observation is [pose_t, pose_t-1, vel_t, vel_t-1]
    const int* amp_off;                      // J   offset of joint j's rotation features inside a pose block (j >= 1)
    const int* amp_ee;                       // J   ordinal of link j among the end effectors, -1 if it is none
    int amp_ee_base;                         // offset of the end-effector positions inside a pose block
    // ---- goal-conditioned AMP task scenes: 1 = target_amp (scenes/SceneTargetAMP.cpp), 2 = heading_amp (SceneHeadingAMP.cpp),
    //      3 = heading_amp_getup (SceneHeadingAMPGetup.cpp), 4 = strike_amp (SceneStrikeAMP.cpp)
    int scene_goal, enable_min_tar_vel, enable_rand_rot_reset;
    int goal_dim;                            // RecordGoal size: 3 (kinds 1, 2), 4 (kinds 3, 4)
    int mode_test;                           // cRLScene::eModeTest (dm_set_mode)
    double getup_time, recover_prob;         // heading_amp_getup: longest get-up clip; recover_episode_prob
    Real getup_height_root, getup_height_head; int head_id, getup_clip_mask;
    double tar_far_prob, init_hit_prob, hit_reset_time, target_min[3], target_max[3];     // strike_amp
    Real tar_near_dist, target_radius, hit_tar_speed, tar_reward_scale; int strike_mask, fail_tar_mask;
    // dribble_amp (scene_goal 5): the ball -- radius, 1/mass, 1/inertia (0.4 m r^2), friction against links and ground, ln(1 - damping)
    Real ball_radius, ball_inv_mass, ball_inv_inertia, ball_friction, ball_thresh; double ball_ln_lin, ball_ln_ang;   // ball_thresh: contact breaking threshold 0.02 r
    double obj_time_min, obj_time_max, min_tar_obj_dist, max_tar_obj_dist;
    double goal_time_min, goal_time_max;     // target timer range (rand_target_time_*)
    Real max_target_dist, target_succ_dist, tar_fail_dist, tar_speed, pos_reward_scale;
    Real max_heading_turn_rate, sharp_turn_prob, speed_change_prob, tar_speed_min, tar_speed_max, vel_reward_scale;
    // ---- multi-clip dataset (`--kin_ctrl clips`): frames / frame_time / frame_vel hold every clip back to back; the members
    // above (F, duration, loop, cycle_delta) describe clip 0.  num_clips <= 1: single clip.
    int num_clips;
    const int* clip_start;                   // num_clips + 1 (rows)
    const double* clip_dur;                  // num_clips
    const int* clip_loop;                    // num_clips
    const Real* clip_delta;                  // num_clips x 3  cycle root delta
    const double* clip_cdf;                  // num_clips  cClipsController::mClipsCDF
    // ---- random perturbations (`--enable_rand_perturbs`, scenes/SceneSimChar.cpp:92-99): a force on a random body part every U[time_min,
    // time_max] seconds, magnitude U[min, max] N, lasting U[min, max] duration; part mask 0 = any part
    int perturb_on; uint32_t perturb_part_mask;
    double perturb_time_min, perturb_time_max, perturb_min, perturb_max, perturb_dur_min, perturb_dur_max;
};
// per-env goal state row (EnvState::goal), doubles: the clocks among them must not round to fp32
enum { GS_TX = 0, GS_TY, GS_TZ, GS_HEADING, GS_SPEED, GS_TIMER, GS_TIMER_MAX, GS_PCOMX, GS_PCOMY, GS_PCOMZ, GS_PTIME, GS_DRAWS, GS_CLIP,
       GS_AUX0, GS_AUX1,        // heading_amp_getup: get-up timer | strike_amp: target hit (0 / 1), hit time
       GS_PBX, GS_PBY, GS_PBZ,  // dribble_amp: ball position at the last action (cSceneDribbleAMP::mAgentPrevTarObjPos)
       GS_OTIMER, GS_OTIMER_MAX, // dribble_amp: target-object timer
       GS_KSEED, GS_KON,        // a draw key of the env's own (dm_set_env_keys): when GS_KON != 0 every counter-based draw of this env is keyed (GS_KSEED, env 0) instead of
                                // (ctx seed, global env id) -- the stream a one-env context created with that seed would draw (the shared-owner facade, deepmimic_amd/broker.py)
       GS_WIDTH = 24 };
// The draw tape (ModelDev::draw_tape, one row of doubles per env; `DM_RNG=reference` of the one-env drop-in, include/dm_hip.h DM_TAPE_*).  The reference draws from two
// std::default_random_engine generators -- cMathUtil::gRand (engine 0: timers, clip choice, reset clip time, part of strike_amp / dribble_amp) and the
// scene's cScene::mRand (engine 1: goal re-sampling, yaw, perturbations, recovery coin) -- in an order that depends on what happens on the device.  The host
// tabulates, for every raw engine position k < TP_K past the generators' current states, what each <random> distribution would return if its next
// call started there (dm_refrand_tape: computed with the standard library's own types); the device looks its draws up in call order and advances the
// two positions, the host then discards that many raw values from its generators.  TP_ERR is raised when a launch runs past the tables.
enum { TP_POS_G = 0, TP_POS_M, TP_NAVAIL, TP_NSAVED,      // raw positions consumed on gRand / mRand; the saved second deviate of mRand's normal_distribution
       TP_ERR,
       TP_CLIPDRAW,                                       // the kinematic controller is a cClipsController: every reset draws a clip from gRand, even out of one (anim/ClipsController.cpp:36-45)
       TP_TMIN, TP_TMAX, TP_TEXP, TP_TPIN,                // the episode timer as cTimer holds it (annealed train-mode parameters whatever the mode; TEXP <= 0: uniform) and the limit test mode pins afterwards (< 0: none)
       TP_HDR = 16, TP_K = 96,
       TP_UG = TP_HDR, TP_EG = TP_UG + TP_K,              // engine 0: uniform(0, 1), -log(1 - uniform)
       TP_UM = TP_EG + TP_K, TP_NM = TP_UM + TP_K,        // engine 1: uniform(0, 1), normal {value, saved value, raw values consumed}
       TP_IM = TP_NM + 3 * TP_K,                          // engine 1: |RandInt()| {value, raw values consumed}
       TP_STRIDE = TP_IM + 2 * TP_K };
// per-env perturbation state (EnvState::pert), doubles: cSceneSimChar::tPerturbParams::mTimer / mNextTime, the draw counter of stream 5,
// and the active tPerturb entries of cWorld's cPerturbManager (link < 0: free slot)
enum { PT_TIMER = 0, PT_NEXT, PT_DRAWS, PT_SLOT0, PT_LINK = 0, PT_FX, PT_FY, PT_FZ, PT_DUR, PT_TIME, PT_SLOT_W = 6, PT_SLOTS = 2, PT_KSEED = 15 /* own draw key + 1 (0: the ctx's), see GS_KSEED */, PT_WIDTH = 16 };
// the free body's record (EnvState::obj, OBJ classes): position, rotation (w, x, y, z), linear and angular velocity
enum { OB_PX = 0, OB_QW = 3, OB_VX = 7, OB_WX = 10, OB_WIDTH = 16 };

template <typename Real>
struct EnvState {
    int N;
    Real* pose;      // N x P   sim character pose (reference layout)
    Real* vel;       // N x P
    Real* tar;       // N x P   PD targets (root slots unused)
    Real* tau;       // N x D   latched SPD torque (generalized-velocity layout)
    Real* kin;       // N x 8   kin origin pos(3) + origin rot(4) + pad
    double* clock;   // N x 6   kin_time, ctrl_time, init_time_offset, timer_time, timer_max, pad
    int* flag;       // N x 4   need_new_action, contact_mask, episode_count, valid
    Real* aovf;      // N x (64 - RREG) x 64  overflow rows of the constraint-space matrix (null when the class keeps all 64 in VGPRs)
    Real* hist;      // N x 2P  pose | vel at the last action latch (cSceneImitateAMP::mPrevPose / mPrevVel); null unless imitate_amp
    Real* obj;       // N x OB_WIDTH  the free body of an OBJ class (dribble_amp's ball); null otherwise
    double* goal;    // N x GS_WIDTH  goal state of the task scenes + the clip the env was reset to; null unless a goal scene / multi-clip dataset
    double* pert;    // N x PT_WIDTH  random-perturbation clock and active forces; null unless enable_rand_perturbs
    Real* manif;     // N x J x MF_STRIDE  physics 2: per link [count, 4 x (point on the link in body coordinates (3), point on the plane x, z, distance)]; null otherwise
};
enum { MF_STRIDE = 32, MF_PT = 6 };

// Per-call I/O of the batched step (device pointers; any may be null)
template <typename Real>
struct StepIO {
    const float* actions;   // N x A (set before the first update of the call when non-null)
    float* states;          // N x S
    float* rewards;         // N
    int* terminate;         // N   eTerminate
    int* valid;             // N   CheckValidEpisode
    int* episode_end;       // N   IsEpisodeEnd
    int n_updates;          // scene updates per call (20 = one control step)
    double dt;              // update timestep (1/600)
    int auto_reset;         // reset envs whose episode ended, after the outputs are written
    int emit;               // write states / rewards / flags at the end of the call
    int open_loop;          // ignore `actions`; track the reference clip (stream A1 of SURVEY 8d)
    float* amp_obs;         // N x amp size  RecordAMPObsAgent at the end of the call (imitate_amp scenes only)
    int end_early;          // stop an env's updates at the update after which its episode is over (DM_END_EPISODE_EARLY)
    float* goals;           // N x goal_dim  RecordGoal at the end of the call (goal scenes only)
    const int* env_ids;     // non-null (one-per-wave kernels, dm_step_envs): workgroup b steps env env_ids[b] -- `actions` rows are indexed by b (compact), every output by the env id
};

// Debug taps for component parity tests (device pointers, null when unused)
template <typename Real>
struct DebugTaps {
    Real* H;        // N x D x D  mass matrix of the last dynamics pass
    Real* C;        // N x D      bias force of the last dynamics pass
    Real* vstar;    // N x D      unconstrained velocity of the last substep
    Real* lambda;   // N x 64     constraint impulses of the last substep
    int* rows;      // N x 2      (num rows, num contacts) of the last substep
    Real* kin_pose; // N x P      kin pose / vel used by the last reward evaluation
    Real* kin_vel;  // N x P
    Real* reward_terms; // N x 5
    Real* links;    // N x J x 21 (com3, Rb9, vcom3, w3, joint3)
    long long* prof; // N x 16 accumulated shader-clock cycles per phase (profiling build of the step kernel only)
};

}  // namespace dmk
