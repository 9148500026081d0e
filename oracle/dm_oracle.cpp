// TEST INFRASTRUCTURE ONLY -- CPU oracle for the DeepMimic imitate hot path.
//
// C entry points (ctypes) over orc::Scene.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load the library built from this file; the product
// (deepmimic_amd/) never does.  Parity status: DeepMimic-side functions are restated from the
// reference sources cited in orc_*.h and pinned by the closed-form known-answer tests in
// tests/test_oracle_kat.py; the rigid-body step replaces un-vendored Bullet 2.88 and is
// "parity unpinned" against real Bullet (no reference tests or golden vectors exist).
#include "orc_scene.h"
#include <chrono>

using namespace orc;

extern "C" {

// cfg[] layout (doubles) -- keep in sync with tests/oracle_lib.py
enum {
    CFG_NUM_SIM_SUBSTEPS = 0, CFG_WORLD_SCALE, CFG_GRAV_X, CFG_GRAV_Y, CFG_GRAV_Z,
    CFG_SYNC_ROOT_POS, CFG_SYNC_ROOT_ROT, CFG_ENABLE_FALL_END, CFG_ENABLE_CONTACT_FALL, CFG_ENABLE_ROOT_ROT_FAIL,
    CFG_ENABLE_RAND_PLACEMENT, CFG_ENABLE_PHASE_INPUT, CFG_RECORD_WORLD_ROOT_POS, CFG_RECORD_WORLD_ROOT_ROT,
    CFG_QUERY_RATE, CFG_FRICTION, CFG_ERP, CFG_SOLVER_ITERS, CFG_MAX_CONTACTS, CFG_SELF_COLLISION, CFG_SCENE_AMP, CFG_AMP_LOCAL_ROOT, CFG_COUNT
};

int orc_cfg_count() { return CFG_COUNT; }
int orc_real_bytes() { return (int)sizeof(real); }

void orc_cfg_default(double* c) {
    SceneCfg d;
    c[CFG_NUM_SIM_SUBSTEPS] = d.num_sim_substeps; c[CFG_WORLD_SCALE] = d.world_scale;
    c[CFG_GRAV_X] = d.gravity.x; c[CFG_GRAV_Y] = d.gravity.y; c[CFG_GRAV_Z] = d.gravity.z;
    c[CFG_SYNC_ROOT_POS] = d.sync_char_root_pos; c[CFG_SYNC_ROOT_ROT] = d.sync_char_root_rot;
    c[CFG_ENABLE_FALL_END] = d.enable_fall_end; c[CFG_ENABLE_CONTACT_FALL] = d.enable_char_contact_fall;
    c[CFG_ENABLE_ROOT_ROT_FAIL] = d.enable_root_rot_fail; c[CFG_ENABLE_RAND_PLACEMENT] = d.enable_rand_char_placement;
    c[CFG_ENABLE_PHASE_INPUT] = d.enable_phase_input; c[CFG_RECORD_WORLD_ROOT_POS] = d.record_world_root_pos;
    c[CFG_RECORD_WORLD_ROOT_ROT] = d.record_world_root_rot; c[CFG_QUERY_RATE] = d.query_rate;
    c[CFG_FRICTION] = d.friction; c[CFG_ERP] = d.erp; c[CFG_SOLVER_ITERS] = d.solver_iters; c[CFG_MAX_CONTACTS] = d.max_contacts; c[CFG_SELF_COLLISION] = d.enable_self_collision;
    c[CFG_SCENE_AMP] = d.scene_amp; c[CFG_AMP_LOCAL_ROOT] = d.enable_amp_obs_local_root;
}

void* orc_create(const double* jm, const double* bd, int J, const double* pd, const double* frames, int F, int loop,
                 const int* fall_mask, const double* c) {
    SceneCfg cfg;
    cfg.num_sim_substeps = (int)c[CFG_NUM_SIM_SUBSTEPS]; cfg.world_scale = c[CFG_WORLD_SCALE];
    cfg.gravity = V3((real)c[CFG_GRAV_X], (real)c[CFG_GRAV_Y], (real)c[CFG_GRAV_Z]);
    cfg.sync_char_root_pos = c[CFG_SYNC_ROOT_POS] != 0; cfg.sync_char_root_rot = c[CFG_SYNC_ROOT_ROT] != 0;
    cfg.enable_fall_end = c[CFG_ENABLE_FALL_END] != 0; cfg.enable_char_contact_fall = c[CFG_ENABLE_CONTACT_FALL] != 0;
    cfg.enable_root_rot_fail = c[CFG_ENABLE_ROOT_ROT_FAIL] != 0; cfg.enable_rand_char_placement = c[CFG_ENABLE_RAND_PLACEMENT] != 0;
    cfg.enable_phase_input = c[CFG_ENABLE_PHASE_INPUT] != 0; cfg.record_world_root_pos = c[CFG_RECORD_WORLD_ROOT_POS] != 0;
    cfg.record_world_root_rot = c[CFG_RECORD_WORLD_ROOT_ROT] != 0; cfg.query_rate = c[CFG_QUERY_RATE];
    cfg.friction = c[CFG_FRICTION]; cfg.erp = c[CFG_ERP]; cfg.solver_iters = (int)c[CFG_SOLVER_ITERS]; cfg.max_contacts = (int)c[CFG_MAX_CONTACTS]; cfg.enable_self_collision = c[CFG_SELF_COLLISION] != 0;
    cfg.scene_amp = c[CFG_SCENE_AMP] != 0; cfg.enable_amp_obs_local_root = c[CFG_AMP_LOCAL_ROOT] != 0;
    Scene* s = new Scene();
    s->init(jm, bd, J, pd, frames, F, loop != 0, fall_mask, cfg);
    return s;
}
void orc_destroy(void* h) { delete (Scene*)h; }

void orc_dims(void* h, int* out /*J,P,A,S,F*/) {
    Scene* s = (Scene*)h; out[0] = s->sk.J; out[1] = s->sk.P; out[2] = s->A; out[3] = s->S; out[4] = s->mo.F;
}
double orc_motion_duration(void* h) { return ((Scene*)h)->mo.duration(); }

void orc_reset(void* h, double kin_time, double max_time) { ((Scene*)h)->reset(kin_time, max_time); }
void orc_set_action(void* h, const double* a) { ((Scene*)h)->set_action(a); }
void orc_update(void* h, double dt) { ((Scene*)h)->update(dt); }
int orc_need_new_action(void* h) { return ((Scene*)h)->need_new_action ? 1 : 0; }
void orc_record_state(void* h, double* out) { ((Scene*)h)->record_state(out); }
int orc_amp_obs_size(void* h) { return ((Scene*)h)->amp_obs_size(); }
void orc_amp_obs_agent(void* h, double* out) { ((Scene*)h)->amp_obs_agent(out); }
void orc_amp_obs_expert(void* h, double t, double* out) { ((Scene*)h)->amp_obs_expert(t, out); }
void orc_prev_state(void* h, double* pose, double* vel) {
    Scene* s = (Scene*)h;
    for (int i = 0; i < s->sk.P; ++i) { pose[i] = s->prev_pose[i]; vel[i] = s->prev_vel[i]; }
}
double orc_calc_reward(void* h) { return ((Scene*)h)->calc_reward(); }
double orc_calc_reward_terms(void* h, double* terms) { return ((Scene*)h)->calc_reward(terms); }
int orc_check_terminate(void* h) { return ((Scene*)h)->check_terminate(); }
int orc_is_episode_end(void* h) { return ((Scene*)h)->is_episode_end() ? 1 : 0; }
int orc_check_valid_episode(void* h) { return ((Scene*)h)->check_valid_episode() ? 1 : 0; }
double orc_time(void* h) { return ((Scene*)h)->timer_time; }
double orc_kin_time(void* h) { return ((Scene*)h)->kin.time; }
double orc_phase(void* h) { return ((Scene*)h)->ctrl_phase(); }

static void copy_out(const Vec& v, double* out) { for (size_t i = 0; i < v.size(); ++i) out[i] = (double)v[i]; }
static Vec copy_in(const double* in, int n) { Vec v(n); for (int i = 0; i < n; ++i) v[i] = (real)in[i]; return v; }

void orc_get_sim_state(void* h, double* pose, double* vel) { Scene* s = (Scene*)h; copy_out(s->pose, pose); copy_out(s->vel, vel); }
void orc_set_sim_state(void* h, const double* pose, const double* vel) {
    Scene* s = (Scene*)h; s->set_sim_state(copy_in(pose, s->sk.P), copy_in(vel, s->sk.P));
}
void orc_get_kin_state(void* h, double* pose, double* vel, double* origin /*3+4*/) {
    Scene* s = (Scene*)h; copy_out(s->kin.pose, pose); copy_out(s->kin.vel, vel);
    origin[0] = s->kin.origin.x; origin[1] = s->kin.origin.y; origin[2] = s->kin.origin.z;
    origin[3] = s->kin.origin_rot.w; origin[4] = s->kin.origin_rot.x; origin[5] = s->kin.origin_rot.y; origin[6] = s->kin.origin_rot.z;
}
void orc_get_tar_pose(void* h, double* out) { copy_out(((Scene*)h)->tar_pose, out); }
void orc_get_tau(void* h, double* out) { copy_out(((Scene*)h)->tau, out); }
void orc_get_contacts(void* h, int* in_contact) { Scene* s = (Scene*)h; for (int j = 0; j < s->sk.J; ++j) in_contact[j] = s->in_contact[j]; }
int orc_dbg_num_rows(void* h) { return ((Scene*)h)->dbg_num_rows; }
int orc_dbg_num_contacts(void* h) { return (int)((Scene*)h)->dbg_contacts.size(); }
// out: n x 9 doubles (link, link_b, dist, x3, n3)
void orc_dbg_contacts(void* h, double* out) { const auto& cs = ((Scene*)h)->dbg_contacts; for (size_t i = 0; i < cs.size(); ++i) { double* o = out + 9 * i; o[0] = cs[i].link; o[1] = cs[i].link_b; o[2] = (double)cs[i].dist; o[3] = (double)cs[i].x.x; o[4] = (double)cs[i].x.y; o[5] = (double)cs[i].x.z; o[6] = (double)cs[i].n.x; o[7] = (double)cs[i].n.y; o[8] = (double)cs[i].n.z; } }
int orc_dbg_num_self_contacts(void* h) { int n = 0; for (const auto& c : ((Scene*)h)->dbg_contacts) n += c.link_b >= 0; return n; }

// ---- component taps (used by the known-answer and component parity tests) ----
// kinematic pose/vel at time t for the current origin (cKinCharacter::CalcPose/CalcVel)
void orc_kin_eval(void* h, double t, double* pose, double* vel) {
    Scene* s = (Scene*)h; Vec p, v; s->kin.calc_pose(t, p); s->kin.calc_vel(t, v); copy_out(p, pose); copy_out(v, vel);
}
void orc_motion_frame(void* h, int f, double* frame, double* frame_vel, double* time) {
    Scene* s = (Scene*)h; copy_out(s->mo.frames[f], frame); copy_out(s->mo.frame_vel[f], frame_vel); *time = s->mo.times[f];
}
// mass matrix H [P x P] and bias force C [P]; which = 0: SPD model (DeepMimic inertias, reference cj),
// 1: simulator model (Bullet inertias, exact cj)
void orc_mass_bias(void* h, int which, const double* pose, const double* vel, double* H, double* C) {
    Scene* s = (Scene*)h; RBDModel& m = which ? s->rbd_sim : s->rbd_ctrl;
    m.update(copy_in(pose, s->sk.P), copy_in(vel, s->sk.P));
    for (size_t i = 0; i < m.H.size(); ++i) H[i] = (double)m.H[i];
    copy_out(m.C, C);
}
// SPD torque for the current state and targets (after clamp), pose layout
void orc_spd_tau(void* h, double dt, double* out) { Scene* s = (Scene*)h; Vec t; s->calc_spd_tau(dt, t); copy_out(t, out); }
// one rigid-body substep with the currently latched torque
void orc_set_tau(void* h, const double* tau) { Scene* s = (Scene*)h; s->tau = copy_in(tau, s->sk.P); }
void orc_substep(void* h, double hstep) { ((Scene*)h)->substep(hstep); }
void orc_get_vstar(void* h, double* out) { copy_out(((Scene*)h)->dbg_vstar, out); }
// link world states: per link com(3) rot-matrix(9) lin vel(3) ang vel(3) joint pos(3) = 21 doubles
void orc_get_links(void* h, double* out) {
    Scene* s = (Scene*)h;
    for (int j = 0; j < s->sk.J; ++j) {
        const LinkState& l = s->links[j]; double* o = out + j * 21;
        o[0] = l.com.x; o[1] = l.com.y; o[2] = l.com.z;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) o[3 + a * 3 + b] = l.Rb.m[a][b];
        o[12] = l.vcom.x; o[13] = l.vcom.y; o[14] = l.vcom.z; o[15] = l.w.x; o[16] = l.w.y; o[17] = l.w.z;
        o[18] = l.joint.t.x; o[19] = l.joint.t.y; o[20] = l.joint.t.z;
    }
}
void orc_calc_com(void* h, const double* pose, const double* vel, double* com, double* com_vel) {
    Scene* s = (Scene*)h; V3 c, v; calc_com(s->sk, copy_in(pose, s->sk.P), copy_in(vel, s->sk.P), c, v);
    com[0] = c.x; com[1] = c.y; com[2] = c.z; com_vel[0] = v.x; com_vel[1] = v.y; com_vel[2] = v.z;
}
// the action that encodes a pose as PD targets (stream A1 of SURVEY 8d): spherical -> exp map
// (cMathUtil::QuaternionToExpMap, MathUtil.cpp:607-615), revolute -> angle
void orc_pose_to_action(void* h, const double* pose, double* action) {
    Scene* s = (Scene*)h;
    for (int j = 1; j < s->sk.J; ++j) {
        int off = s->sk.offset(j), ao = s->act_off[j];
        if (s->sk.type(j) == JT_SPHERICAL) {
            V3 e = quat_to_exp_map(Q4((real)pose[off], (real)pose[off + 1], (real)pose[off + 2], (real)pose[off + 3]));
            action[ao] = e.x; action[ao + 1] = e.y; action[ao + 2] = e.z;
        } else for (int k = 0; k < s->sk.size(j); ++k) action[ao + k] = pose[off + k];
    }
}

// Fixed-action rollout used for the CPU baseline: `steps` control steps of `updates_per_step` updates.
// actions: steps x A (or NULL -> open-loop mocap tracking, stream A1).  Returns wall seconds.
double orc_rollout(void* h, int steps, int updates_per_step, double dt, const double* actions, double* rewards, double* states) {
    Scene* s = (Scene*)h;
    std::vector<double> a(s->A, 0.0), kp(s->sk.P), kv(s->sk.P);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) {
        if (actions) s->set_action(actions + (size_t)k * s->A);
        else { orc_kin_eval(h, s->kin.time, kp.data(), kv.data()); orc_pose_to_action(h, kp.data(), a.data()); s->set_action(a.data()); }
        for (int u = 0; u < updates_per_step; ++u) s->update(dt);
        if (rewards) rewards[k] = s->calc_reward();
        if (states) s->record_state(states + (size_t)k * s->S);
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
