// Device-visible tables and per-env state for the imitate hot path.
//
// One `ModelDev` describes one scene type (skeleton + PD gains + motion clip + config); it is built on
// the host by dm_host.cpp from the raw reference-layout tables and is shared by all envs.
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
constexpr int kMaxCand = 64;   // ground-contact candidate points per character (one per lane)

template <typename Real>
struct ModelDev {
    int J, P, D, A, S, F, NC, NL, max_depth;
    // ---- per link / joint (index j), J entries
    const int* parent; const int* jtype; const int* pose_off; const int* dof_off; const int* ndof;
    const int* depth; const int* act_off; const int* is_ee; const int* fall; const int* brot_ident; const int* arot_ident;
    const uint32_t* subtree_mask;            // bit k: link k is in the subtree rooted at j (incl. j)
    const Real* attach;                      // J x 3  joint attach point in the parent joint frame
    const Real* attach_rot;                  // J x 9  joint attach rotation
    const Real* battach;                     // J x 3  body COM in the joint frame
    const Real* brot;                        // J x 9  body frame in the joint frame
    const Real* mass;                        // J
    const Real* inertia;                     // 2 x J x 3  principal inertias: [0] SPD model, [1] simulator model
    const Real* torque_lim; const Real* lim_lo; const Real* lim_hi; const Real* diffw; const Real* thresh;
    const Real* aabb_he;                     // J x 4  half extents of the collider AABB box (w = 1: sphere)
    // ---- per generalized velocity (index i), D entries: root lin 0..2, root ang 3..5, then joints
    const int* dof_joint; const int* dof_kind; const int* dof_axis; const int* dof_vidx;
    const uint64_t* dof_anc;                 // bit k: dof k belongs to an ancestor-or-self joint of dof i's joint
    const Real* kp; const Real* kd;
    // ---- ground contact candidates, NC entries
    const int* cand_link; const Real* cand_loc /* NC x 3, body frame */; const Real* cand_rad;
    // ---- joint-limit rows, NL entries (revolute joints with lo <= hi)
    const int* lim_joint;
    // ---- motion clip
    const double* frame_time;                // F
    const Real* frames;                      // F x P (post-processed)
    const Real* frame_vel;                   // F x P
    double duration; int loop; Real cycle_delta[3];
    // ---- config
    Real gravity[3];
    int num_sim_substeps, solver_iters, max_contacts;
    Real friction, erp, report_dist, max_lin_vel, max_ang_vel, slerp_one;
    int sync_root_pos, sync_root_rot, enable_fall_end, enable_contact_fall, enable_root_rot_fail, enable_rand_placement;
    int enable_phase_input, record_world_root_pos, record_world_root_rot;
    double query_period;                     // 1 / QueryRate
    double time_lim_min, time_lim_max;       // episode timer range (uniform)
    uint64_t seed;
    int env_off;                             // global id of env 0 of this shard (keeps RNG streams partition-invariant)
};

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
};

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
