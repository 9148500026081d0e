// TEST INFRASTRUCTURE ONLY -- CPU oracle for the DeepMimic imitate hot path.
//
// C entry points (ctypes) over orc::Scene.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load the library built from this file; the product
// (deepmimic_amd/) never does.  Parity status: DeepMimic-side functions are restated from the
// reference sources cited in orc_*.h and PINNED to the reference's own code -- oracle/_ref compiles the unmodified
// MathUtil / SpAlg / KinTree / Motion / KinCharacter / RBDUtil / RBDModel / CtCtrlUtil / DynamicTimeWarper sources against a
// minimal Eigen-API shim (oracle/build_ref.sh) and tests/test_oracle_vs_ref.py holds this restatement against it on 1 000 random
// states per character (H and C bit-identical, SPD torque 1e-11); reference-generated golden vectors: tests/golden/ref_vectors.npz.
// Closed-form known-answer tests: tests/test_oracle_kat.py.  The rigid-body step replaces un-vendored Bullet 2.88 and stays
// "parity unpinned" against real Bullet (no reference tests or golden vectors exist for it); the goal-scene draws are specified on
// the device's counter-based generator, not on the reference's std::default_random_engine.
#include "orc_scene.h"
#include <chrono>
#include <cstdint>

using namespace orc;

extern "C" {

// cfg[] layout (doubles) -- keep in sync with tests/oracle_lib.py
enum {
    CFG_NUM_SIM_SUBSTEPS = 0, CFG_WORLD_SCALE, CFG_GRAV_X, CFG_GRAV_Y, CFG_GRAV_Z,
    CFG_SYNC_ROOT_POS, CFG_SYNC_ROOT_ROT, CFG_ENABLE_FALL_END, CFG_ENABLE_CONTACT_FALL, CFG_ENABLE_ROOT_ROT_FAIL,
    CFG_ENABLE_RAND_PLACEMENT, CFG_ENABLE_PHASE_INPUT, CFG_RECORD_WORLD_ROOT_POS, CFG_RECORD_WORLD_ROOT_ROT,
    CFG_QUERY_RATE, CFG_FRICTION, CFG_ERP, CFG_SOLVER_ITERS, CFG_MAX_CONTACTS, CFG_SELF_COLLISION, CFG_SCENE_AMP, CFG_AMP_LOCAL_ROOT,
    CFG_SCENE_GOAL, CFG_RAND_ROT_RESET, CFG_TAR_TIME_MIN, CFG_TAR_TIME_MAX, CFG_MAX_TAR_DIST, CFG_TAR_SUCC_DIST, CFG_TAR_FAIL_DIST, CFG_TAR_SPEED, CFG_POS_REWARD_SCALE,
    CFG_MIN_TAR_VEL, CFG_MAX_TURN_RATE, CFG_SHARP_TURN_PROB, CFG_SPEED_CHANGE_PROB, CFG_TAR_SPEED_MIN, CFG_TAR_SPEED_MAX, CFG_VEL_REWARD_SCALE,
    CFG_MODE_TEST, CFG_GETUP_TIME, CFG_GETUP_HEIGHT_ROOT, CFG_GETUP_HEIGHT_HEAD, CFG_HEAD_ID, CFG_RECOVER_PROB, CFG_GETUP_CLIP_MASK,
    CFG_TAR_NEAR_DIST, CFG_TAR_FAR_PROB, CFG_TARGET_RADIUS, CFG_HIT_RESET_TIME, CFG_INIT_HIT_PROB, CFG_HIT_TAR_SPEED, CFG_TAR_REWARD_SCALE,
    CFG_TMIN_X, CFG_TMIN_Y, CFG_TMIN_Z, CFG_TMAX_X, CFG_TMAX_Y, CFG_TMAX_Z, CFG_STRIKE_MASK, CFG_FAIL_TAR_MASK,
    CFG_OBJ_TIME_MIN, CFG_OBJ_TIME_MAX, CFG_MIN_OBJ_DIST, CFG_MAX_OBJ_DIST, CFG_BALL_RADIUS, CFG_BALL_MASS, CFG_BALL_FRICTION, CFG_BALL_LIN_DAMP, CFG_BALL_ANG_DAMP,
    CFG_PERTURB_ON, CFG_PERTURB_TIME_MIN, CFG_PERTURB_TIME_MAX, CFG_PERTURB_MIN, CFG_PERTURB_MAX, CFG_PERTURB_DUR_MIN, CFG_PERTURB_DUR_MAX, CFG_PERTURB_PART_MASK, CFG_PHYSICS, CFG_COUNT
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
    c[CFG_SCENE_GOAL] = d.scene_goal; c[CFG_RAND_ROT_RESET] = d.enable_rand_rot_reset; c[CFG_TAR_TIME_MIN] = d.rand_target_time_min; c[CFG_TAR_TIME_MAX] = d.rand_target_time_max;
    c[CFG_MAX_TAR_DIST] = d.max_target_dist; c[CFG_TAR_SUCC_DIST] = d.target_succ_dist; c[CFG_TAR_FAIL_DIST] = d.tar_fail_dist; c[CFG_TAR_SPEED] = d.tar_speed;
    c[CFG_POS_REWARD_SCALE] = d.pos_reward_scale; c[CFG_MIN_TAR_VEL] = d.enable_min_tar_vel; c[CFG_MAX_TURN_RATE] = d.max_heading_turn_rate;
    c[CFG_SHARP_TURN_PROB] = d.sharp_turn_prob; c[CFG_SPEED_CHANGE_PROB] = d.speed_change_prob; c[CFG_TAR_SPEED_MIN] = d.tar_speed_min; c[CFG_TAR_SPEED_MAX] = d.tar_speed_max;
    c[CFG_VEL_REWARD_SCALE] = d.vel_reward_scale;
    c[CFG_MODE_TEST] = d.mode_test; c[CFG_GETUP_TIME] = d.getup_time; c[CFG_GETUP_HEIGHT_ROOT] = d.getup_height_root; c[CFG_GETUP_HEIGHT_HEAD] = d.getup_height_head;
    c[CFG_HEAD_ID] = d.head_id; c[CFG_RECOVER_PROB] = d.recover_episode_prob; c[CFG_GETUP_CLIP_MASK] = d.getup_clip_mask;
    c[CFG_TAR_NEAR_DIST] = d.tar_near_dist; c[CFG_TAR_FAR_PROB] = d.tar_far_prob; c[CFG_TARGET_RADIUS] = d.target_radius; c[CFG_HIT_RESET_TIME] = d.target_hit_reset_time;
    c[CFG_INIT_HIT_PROB] = d.init_hit_prob; c[CFG_HIT_TAR_SPEED] = d.hit_tar_speed; c[CFG_TAR_REWARD_SCALE] = d.tar_reward_scale;
    for (int k = 0; k < 3; ++k) { c[CFG_TMIN_X + k] = d.target_min[k]; c[CFG_TMAX_X + k] = d.target_max[k]; }
    c[CFG_STRIKE_MASK] = d.strike_mask; c[CFG_FAIL_TAR_MASK] = d.fail_tar_mask;
    c[CFG_OBJ_TIME_MIN] = d.tar_obj_time_min; c[CFG_OBJ_TIME_MAX] = d.tar_obj_time_max; c[CFG_MIN_OBJ_DIST] = d.min_tar_obj_dist; c[CFG_MAX_OBJ_DIST] = d.max_tar_obj_dist;
    c[CFG_BALL_RADIUS] = d.ball_radius; c[CFG_BALL_MASS] = d.ball_mass; c[CFG_BALL_FRICTION] = d.ball_friction; c[CFG_BALL_LIN_DAMP] = d.ball_lin_damping; c[CFG_BALL_ANG_DAMP] = d.ball_ang_damping;
    c[CFG_PERTURB_ON] = d.enable_rand_perturbs; c[CFG_PERTURB_TIME_MIN] = d.perturb_time_min; c[CFG_PERTURB_TIME_MAX] = d.perturb_time_max; c[CFG_PERTURB_MIN] = d.min_perturb;
    c[CFG_PERTURB_MAX] = d.max_perturb; c[CFG_PERTURB_DUR_MIN] = d.min_perturb_duration; c[CFG_PERTURB_DUR_MAX] = d.max_perturb_duration; c[CFG_PERTURB_PART_MASK] = d.perturb_part_mask; c[CFG_PHYSICS] = d.physics;
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
    cfg.scene_goal = (int)c[CFG_SCENE_GOAL]; cfg.enable_rand_rot_reset = c[CFG_RAND_ROT_RESET] != 0; cfg.rand_target_time_min = c[CFG_TAR_TIME_MIN]; cfg.rand_target_time_max = c[CFG_TAR_TIME_MAX];
    cfg.max_target_dist = c[CFG_MAX_TAR_DIST]; cfg.target_succ_dist = c[CFG_TAR_SUCC_DIST]; cfg.tar_fail_dist = c[CFG_TAR_FAIL_DIST]; cfg.tar_speed = c[CFG_TAR_SPEED];
    cfg.pos_reward_scale = c[CFG_POS_REWARD_SCALE]; cfg.enable_min_tar_vel = c[CFG_MIN_TAR_VEL] != 0; cfg.max_heading_turn_rate = c[CFG_MAX_TURN_RATE];
    cfg.sharp_turn_prob = c[CFG_SHARP_TURN_PROB]; cfg.speed_change_prob = c[CFG_SPEED_CHANGE_PROB]; cfg.tar_speed_min = c[CFG_TAR_SPEED_MIN]; cfg.tar_speed_max = c[CFG_TAR_SPEED_MAX];
    cfg.vel_reward_scale = c[CFG_VEL_REWARD_SCALE];
    cfg.mode_test = c[CFG_MODE_TEST] != 0; cfg.getup_time = c[CFG_GETUP_TIME]; cfg.getup_height_root = c[CFG_GETUP_HEIGHT_ROOT]; cfg.getup_height_head = c[CFG_GETUP_HEIGHT_HEAD];
    cfg.head_id = (int)c[CFG_HEAD_ID]; cfg.recover_episode_prob = c[CFG_RECOVER_PROB]; cfg.getup_clip_mask = (uint32_t)c[CFG_GETUP_CLIP_MASK];
    cfg.tar_near_dist = c[CFG_TAR_NEAR_DIST]; cfg.tar_far_prob = c[CFG_TAR_FAR_PROB]; cfg.target_radius = c[CFG_TARGET_RADIUS]; cfg.target_hit_reset_time = c[CFG_HIT_RESET_TIME];
    cfg.init_hit_prob = c[CFG_INIT_HIT_PROB]; cfg.hit_tar_speed = c[CFG_HIT_TAR_SPEED]; cfg.tar_reward_scale = c[CFG_TAR_REWARD_SCALE];
    for (int k = 0; k < 3; ++k) { cfg.target_min[k] = c[CFG_TMIN_X + k]; cfg.target_max[k] = c[CFG_TMAX_X + k]; }
    cfg.strike_mask = (uint32_t)c[CFG_STRIKE_MASK]; cfg.fail_tar_mask = (uint32_t)c[CFG_FAIL_TAR_MASK];
    cfg.tar_obj_time_min = c[CFG_OBJ_TIME_MIN]; cfg.tar_obj_time_max = c[CFG_OBJ_TIME_MAX]; cfg.min_tar_obj_dist = c[CFG_MIN_OBJ_DIST]; cfg.max_tar_obj_dist = c[CFG_MAX_OBJ_DIST];
    cfg.ball_radius = c[CFG_BALL_RADIUS]; cfg.ball_mass = c[CFG_BALL_MASS]; cfg.ball_friction = c[CFG_BALL_FRICTION]; cfg.ball_lin_damping = c[CFG_BALL_LIN_DAMP]; cfg.ball_ang_damping = c[CFG_BALL_ANG_DAMP];
    cfg.enable_rand_perturbs = c[CFG_PERTURB_ON] != 0; cfg.perturb_time_min = c[CFG_PERTURB_TIME_MIN]; cfg.perturb_time_max = c[CFG_PERTURB_TIME_MAX]; cfg.min_perturb = c[CFG_PERTURB_MIN];
    cfg.max_perturb = c[CFG_PERTURB_MAX]; cfg.min_perturb_duration = c[CFG_PERTURB_DUR_MIN]; cfg.max_perturb_duration = c[CFG_PERTURB_DUR_MAX]; cfg.perturb_part_mask = (uint32_t)c[CFG_PERTURB_PART_MASK]; cfg.physics = (int)c[CFG_PHYSICS];
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
// physics 2: the persistent ground manifolds, J x 25 = {count, 4 x (lp (3), bx, bz, dist)} -- the layout of the product's dm_get_manifolds
void orc_get_manifolds(void* h, double* out) {
    Scene* s = (Scene*)h;
    for (int j = 0; j < s->sk.J; ++j) {
        double* o = out + (size_t)j * 25; for (int k = 0; k < 25; ++k) o[k] = 0;
        if (j >= (int)s->manifolds.size()) continue;
        const auto& mf = s->manifolds[j];
        o[0] = (double)mf.size();
        for (size_t i = 0; i < mf.size() && i < 4; ++i) { o[1 + 6 * i] = mf[i].lp.x; o[2 + 6 * i] = mf[i].lp.y; o[3 + 6 * i] = mf[i].lp.z; o[4 + 6 * i] = mf[i].bx; o[5 + 6 * i] = mf[i].bz; o[6 + 6 * i] = mf[i].dist; }
    }
}
void orc_set_manifolds(void* h, const double* in) {
    Scene* s = (Scene*)h;
    s->manifolds.assign(s->sk.J, std::vector<Scene::ManifoldPt>());
    for (int j = 0; j < s->sk.J; ++j) {
        const double* o = in + (size_t)j * 25;
        for (int i = 0; i < (int)o[0] && i < 4; ++i) {
            Scene::ManifoldPt p; p.lp.x = (real)o[1 + 6 * i]; p.lp.y = (real)o[2 + 6 * i]; p.lp.z = (real)o[3 + 6 * i]; p.bx = (real)o[4 + 6 * i]; p.bz = (real)o[5 + 6 * i]; p.dist = (real)o[6 + 6 * i];
            s->manifolds[j].push_back(p);
        }
    }
}
void orc_get_kin_state(void* h, double* pose, double* vel, double* origin /*3+4*/) {
    Scene* s = (Scene*)h; copy_out(s->kin.pose, pose); copy_out(s->kin.vel, vel);
    origin[0] = s->kin.origin.x; origin[1] = s->kin.origin.y; origin[2] = s->kin.origin.z;
    origin[3] = s->kin.origin_rot.w; origin[4] = s->kin.origin_rot.x; origin[5] = s->kin.origin_rot.y; origin[6] = s->kin.origin_rot.z;
}
void orc_get_tar_pose(void* h, double* out) { copy_out(((Scene*)h)->tar_pose, out); }
// (tests: hand the AMP pose history in, as the device keeps it in EnvState::hist)
void orc_set_prev_state(void* h, const double* pose, const double* vel) {
    Scene* s = (Scene*)h;
    s->prev_pose = copy_in(pose, s->sk.P); s->prev_vel = copy_in(vel, s->sk.P);
}
void orc_set_tar_pose(void* h, const double* in) { Scene* s = (Scene*)h; s->tar_pose = copy_in(in, s->sk.P); }
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


// ---- component exports mirrored by oracle/ref_glue.cpp (same names with the ref_ prefix, same op codes): used by
// tests/test_oracle_vs_ref.py to hold this restatement against the reference's own compiled sources -------------
static Q4 qin_(const double* p) { return Q4((real)p[0], (real)p[1], (real)p[2], (real)p[3]); }
static void qout_(const Q4& q, double* p) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
static V3 v3in_(const double* p) { return V3((real)p[0], (real)p[1], (real)p[2]); }
static void v3out_(const V3& v, double* p) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static void m3out_(const M3& m, double* p) { for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) p[a * 3 + b] = m.m[a][b]; }
static void xfout_(const Xf& x, double* p) { m3out_(x.R, p); p[9] = x.t.x; p[10] = x.t.y; p[11] = x.t.z; }

int orc_math_op(int op, const double* in, double* out) {
    switch (op) {
        case 0: qout_(exp_map_to_quat(v3in_(in)), out); return 4;
        case 1: v3out_(quat_to_exp_map(qin_(in)), out); return 3;
        case 2: out[0] = quat_diff_theta(qin_(in), qin_(in + 4)); return 1;
        case 3: v3out_(quat_vel(qin_(in), qin_(in + 4), (real)in[8]), out); return 3;
        case 4: v3out_(quat_vel_rel(qin_(in), qin_(in + 4), (real)in[8]), out); return 3;
        case 5: { Q4 q = qin_(in); v3out_(qrot(q, V3(0, 1, 0)), out); v3out_(qrot(q, V3(1, 0, 0)), out + 3); return 6; }   // as Scene::record_state
        case 6: out[0] = normalize_angle((real)in[0]); return 1;
        case 7: qout_(slerp(qin_(in), (real)in[8], qin_(in + 4)), out); return 4;
        case 8: out[0] = check_next_interval(in[0], in[1], in[2]) ? 1 : 0; return 1;
        case 9: out[0] = calc_heading(qin_(in)); return 1;
        case 10: { V3 ax; real th; quat_to_axis_angle(qin_(in), ax, th); v3out_(ax, out); out[3] = th; return 4; }
        case 11: qout_(quat_axis_angle(v3in_(in), (real)in[3]), out); return 4;
        case 12: v3out_(qrot(qin_(in), v3in_(in + 4)), out); return 3;
        case 13: m3out_(rot_quat(qin_(in)), out); return 9;
        case 14: m3out_(rot_euler(v3in_(in)), out); return 9;
        case 15: qout_(quat_euler(v3in_(in)), out); return 4;
        case 16: qout_(standardize(qin_(in)), out); return 4;
        case 17: qout_(quat_diff(qin_(in), qin_(in + 4)), out); return 4;
        case 18: m3out_(rot_axis(v3in_(in), (real)in[3]), out); return 9;
        case 19: qout_(quat_from_rot(rot_quat(qin_(in))), out); return 4;
        case 20: qout_(quat_axis_angle(V3(0, 1, 0), -calc_heading(qin_(in))), out); return 4;   // KinTree.cpp:1637-1643
        case 21: {   // cMotion::CalcPhase (Motion.cpp:32-47)
            double time = in[0], period = in[1], phase_offset = in[2]; bool loop = in[3] != 0;
            double phase = time / period + phase_offset;
            if (loop) { phase -= std::floor(phase); } else { phase = std::min(std::max(phase, 0.0), 1.0); }
            out[0] = phase; return 1;
        }
    }
    return -1;
}

struct SkelH { Skeleton sk; RBDModel rbd; };
void* orc_skel_create(const double* jm, const double* bd, int J, const double* gravity) {
    SkelH* s = new SkelH();
    s->sk.init(jm, bd, J);
    RBDOpts o;   // DeepMimic inertias, the reference's own root Cj
    s->rbd.init(&s->sk, o, V3((real)gravity[0], (real)gravity[1], (real)gravity[2]));
    return s;
}
void orc_skel_destroy(void* h) { delete (SkelH*)h; }
int orc_skel_num_dof(void* h) { return ((SkelH*)h)->sk.P; }
void orc_skel_lerp_poses(void* h, const double* p0, const double* p1, double t, double* out) {
    SkelH* s = (SkelH*)h; Vec o; lerp_poses(s->sk, copy_in(p0, s->sk.P), copy_in(p1, s->sk.P), (real)t, o); copy_out(o, out);
}
void orc_skel_calc_vel(void* h, const double* p0, const double* p1, double dt, double* out) {
    SkelH* s = (SkelH*)h; Vec o; calc_vel(s->sk, copy_in(p0, s->sk.P), copy_in(p1, s->sk.P), (real)dt, o); copy_out(o, out);
}
void orc_skel_vel_to_pose_diff(void* h, const double* pose, const double* vel, double* out) {
    SkelH* s = (SkelH*)h; Vec o; vel_to_pose_diff(s->sk, copy_in(pose, s->sk.P), copy_in(vel, s->sk.P), o); copy_out(o, out);
}
void orc_skel_post_process_pose(void* h, double* pose) {
    SkelH* s = (SkelH*)h; Vec p = copy_in(pose, s->sk.P); post_process_pose(s->sk, p); copy_out(p, pose);
}
void orc_skel_pose_errs(void* h, const double* p0, const double* p1, const double* v0, const double* v1, double* out) {
    SkelH* s = (SkelH*)h; const Skeleton& sk = s->sk;
    Vec a = copy_in(p0, sk.P), b = copy_in(p1, sk.P), c = copy_in(v0, sk.P), d = copy_in(v1, sk.P);
    // the per-joint terms exactly as Scene::calc_reward forms them
    real th = quat_diff_theta(root_rot(a), root_rot(b));
    out[0] = th * th; out[1] = norm2(root_ang_vel(d) - root_ang_vel(c)); out[2] = 0; out[2 + sk.J] = 0;
    for (int j = 1; j < sk.J; ++j) {
        int off = sk.offset(j), sz = sk.size(j); real pe = 0, ve = 0;
        if (sk.type(j) == JT_SPHERICAL) { real t = quat_theta(quat_diff(joint_quat(a, off), joint_quat(b, off))); pe = t * t; }
        else for (int k = 0; k < sz; ++k) { real e = b[off + k] - a[off + k]; pe += e * e; }
        for (int k = 0; k < sz; ++k) { real e = d[off + k] - c[off + k]; ve += e * e; }
        out[2 + j] = pe; out[2 + sk.J + j] = ve;
    }
}
void orc_skel_world_trans(void* h, const double* pose, double* out_joint, double* out_body) {
    SkelH* s = (SkelH*)h; const Skeleton& sk = s->sk; Vec p = copy_in(pose, sk.P);
    for (int j = 0; j < sk.J; ++j) {
        Xf jw = joint_world_trans(sk, p, j);
        xfout_(jw, out_joint + 12 * j);
        if (sk.valid_body(j)) xfout_(jw * body_joint_trans(sk, j), out_body + 12 * j);
        else for (int k = 0; k < 12; ++k) out_body[12 * j + k] = 0;
    }
}
void orc_skel_link_vel(void* h, const double* pose, const double* vel, double* out) {
    SkelH* s = (SkelH*)h; std::vector<LinkState> L; calc_links(s->sk, copy_in(pose, s->sk.P), copy_in(vel, s->sk.P), L);
    for (int j = 0; j < s->sk.J; ++j) {
        if (s->sk.valid_body(j)) v3out_(L[j].vcom, out + 6 * j); else v3out_(V3(), out + 6 * j);
        v3out_(L[j].w, out + 6 * j + 3);
    }
}
void orc_skel_mass_bias(void* h, const double* pose, const double* vel, double* H, double* C) {
    SkelH* s = (SkelH*)h; s->rbd.update(copy_in(pose, s->sk.P), copy_in(vel, s->sk.P));
    for (size_t i = 0; i < s->rbd.H.size(); ++i) H[i] = (double)s->rbd.H[i];
    copy_out(s->rbd.C, C);
}
void orc_skel_inv_dyna(void* h, const double* pose, const double* vel, const double* acc, double* tau) {
    SkelH* s = (SkelH*)h; s->rbd.update(copy_in(pose, s->sk.P), copy_in(vel, s->sk.P));
    Vec t; s->rbd.solve_inv_dyna(copy_in(acc, s->sk.P), t); copy_out(t, tau);
}
void orc_skel_com(void* h, const double* pose, const double* vel, double* com, double* com_vel) {
    SkelH* s = (SkelH*)h; V3 c, v; calc_com(s->sk, copy_in(pose, s->sk.P), copy_in(vel, s->sk.P), c, v); v3out_(c, com); v3out_(v, com_vel);
}
void orc_skel_origin_trans(void* h, const double* pose, double* out12) { xfout_(origin_trans(copy_in(pose, ((SkelH*)h)->sk.P)), out12); }
double orc_skel_total_mass(void* h) { SkelH* s = (SkelH*)h; double m = 0; for (int j = 0; j < s->sk.J; ++j) if (s->sk.valid_body(j)) m += s->sk.mass(j); return m; }
void orc_skel_inertia(void* h, int j, double* out36) {
    SM I = ((SkelH*)h)->rbd.moment_inertia(j);
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out36[a * 6 + b] = (double)I.m[a][b];
}
// unclamped SPD torque (pose layout) for explicit gains: the arithmetic of Scene::calc_spd_tau before the clamp
void orc_skel_spd_tau(void* h, const double* pose_, const double* vel_, const double* tar_, const double* kp, const double* kd, double dt, double* out) {
    SkelH* s = (SkelH*)h; const Skeleton& sk = s->sk; const int P = sk.P; real t = (real)dt;
    Vec rp = copy_in(pose_, P), vel = copy_in(vel_, P), tar = copy_in(tar_, P);
    s->rbd.update(rp, vel);
    std::vector<real> M = s->rbd.H;
    for (int i = 0; i < P; ++i) M[(size_t)i * P + i] += t * (real)kd[i];
    Vec inc; vel_to_pose_diff(sk, rp, vel, inc);
    for (int i = 0; i < P; ++i) inc[i] = rp[i] + t * inc[i];
    post_process_pose(sk, inc);
    Vec pose_err; calc_vel(sk, inc, tar, 1, pose_err);
    Vec acc(P, 0);
    for (int i = 0; i < P; ++i) acc[i] = (real)kp[i] * pose_err[i] + (real)kd[i] * (0 - vel[i]) - s->rbd.C[i];
    Vec sol; ldlt_solve(M, P, acc, sol);
    for (int i = 0; i < P; ++i) out[i] = (double)((real)kp[i] * pose_err[i] + (real)kd[i] * ((0 - vel[i]) - t * sol[i]));
}
void orc_set_kin_origin(void* h, const double* pos3, const double* rot4) {
    Scene* s = (Scene*)h; s->kin.origin = v3in_(pos3); s->kin.origin_rot = qin_(rot4);
}
// (tests / bench.py's parity check: put the oracle where the DEVICE is -- the arrays of the product's dm_get_state: pose, vel, PD targets, kinematic
//  origin {pos, rot wxyz}, clocks {kin time, controller time, init time offset, episode timer, its limit}, flags {need_new_action, ground-contact
//  mask, episode counter (unused here), -}; the kinematic pose is re-posed from (time, origin) as cKinCharacter::Pose does)
void orc_set_full_state(void* h, const double* pose, const double* vel, const double* tar, const double* origin7, const double* clocks5, const int* flags4) {
    Scene* s = (Scene*)h;
    s->set_sim_state(copy_in(pose, s->sk.P), copy_in(vel, s->sk.P));
    s->tar_pose = copy_in(tar, s->sk.P);
    s->kin.origin = v3in_(origin7); s->kin.origin_rot = qin_(origin7 + 3);
    s->kin.time = clocks5[0]; s->kin.do_pose();
    s->ctrl_time = clocks5[1]; s->init_time_offset = clocks5[2]; s->timer_time = clocks5[3]; s->timer_max = clocks5[4];
    s->need_new_action = flags4[0] != 0;
    for (int j = 0; j < s->sk.J; ++j) s->in_contact[j] = (flags4[1] >> j) & 1;
}
void orc_kin_set_time(void* h, double t) { Scene* s = (Scene*)h; s->kin.time = t; s->kin.do_pose(); }
void orc_kin_set_root_pos(void* h, const double* p3) { ((Scene*)h)->kin.set_root_pos_(v3in_(p3)); }
void orc_kin_rotate_root(void* h, const double* q4) { ((Scene*)h)->kin.rotate_root(qin_(q4)); }
void orc_motion_eval(void* h, double t, double* frame, double* vel) {
    Scene* s = (Scene*)h; Vec f, v; s->mo.calc_frame(s->sk, t, f); s->mo.calc_frame_vel(t, v); copy_out(f, frame); copy_out(v, vel);
}
int orc_kin_cycle(void* h, double t) { return ((Scene*)h)->mo.cycle_count(t); }

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

// ---- goal scenes / multi-clip datasets
// frames: the concatenation of the clips (total rows x (1 + P)); clip c owns rows starts[c] .. starts[c+1]-1
void orc_set_clips(void* h, const double* frames, const int* starts, const int* loops, const double* weights, int n) {
    Scene* s = (Scene*)h;
    s->clips.resize(n); s->clip_cdf.assign(n, 0);
    double w = 0;
    for (int c = 0; c < n; ++c) {
        s->clips[c].load(s->sk, frames + (size_t)starts[c] * (s->sk.P + 1), starts[c + 1] - starts[c], s->sk.P, loops[c] != 0);
        w += weights[c]; s->clip_cdf[c] = w;
    }
    for (int c = 0; c < n; ++c) s->clip_cdf[c] /= w;
}
int orc_num_clips(void* h) { return (int)((Scene*)h)->clips.size(); }
double orc_clip_duration(void* h, int c) { Scene* s = (Scene*)h; return s->clips.empty() ? s->mo.duration() : s->clips[c].duration(); }
// random perturbations: the same 16-double row as the device (include/dm_hip.h dm_get_perturb_state); entries beyond two are not representable
void orc_perturb_state(void* h, double* o) {
    Scene* s = (Scene*)h;
    for (int k = 0; k < 16; ++k) o[k] = 0;
    o[0] = s->pert_timer; o[1] = s->pert_next; o[2] = (double)s->pert_draws;
    for (size_t i = 0; i < s->perts.size() && i < 2; ++i) { const Scene::Perturb& p = s->perts[i]; double* q = o + 3 + 6 * i; q[0] = p.link + 1; q[1] = p.f.x; q[2] = p.f.y; q[3] = p.f.z; q[4] = p.dur; q[5] = p.time; }
}
int orc_num_perturbs(void* h) { return (int)((Scene*)h)->perts.size(); }
void orc_set_perturb_state(void* h, const double* o) {
    Scene* s = (Scene*)h;
    s->pert_timer = o[0]; s->pert_next = o[1]; s->pert_draws = (uint64_t)o[2]; s->perts.clear();
    for (int i = 0; i < 2; ++i) { const double* q = o + 3 + 6 * i; if (q[0] > 0) { Scene::Perturb p; p.link = (int)q[0] - 1; p.f.x = q[1]; p.f.y = q[2]; p.f.z = q[3]; p.dur = q[4]; p.time = q[5]; s->perts.push_back(p); } }
}
void orc_goal_rng(void* h, uint64_t seed, uint64_t env_id, uint64_t draws) { Scene* s = (Scene*)h; s->rng_seed = seed; s->rng_env = env_id; s->goal_draws = draws; }
// reset to a named clip / clip time / yaw (what the device draws with streams 3, 0, 4 of its reset generator)
void orc_reset_ex(void* h, double kin_time, double max_time, int clip, double yaw) { ((Scene*)h)->reset(kin_time, max_time, clip, (real)yaw); }
int orc_draw_clip(void* h, double u) { Scene* s = (Scene*)h; int c = 0; int n = (int)s->clip_cdf.size(); while (c < n - 1 && !(u < s->clip_cdf[c])) ++c; return c; }
void orc_record_goal(void* h, double* out) { ((Scene*)h)->record_goal(out); }
void orc_goal_state(void* h, double* o) {
    Scene* s = (Scene*)h;
    o[0] = s->tar_pos.x; o[1] = s->tar_pos.y; o[2] = s->tar_pos.z; o[3] = s->tar_heading; o[4] = s->tar_speed; o[5] = s->tar_timer; o[6] = s->tar_timer_max;
    o[7] = s->prev_action_com.x; o[8] = s->prev_action_com.y; o[9] = s->prev_action_com.z; o[10] = s->prev_action_time; o[11] = (double)s->goal_draws;
    o[12] = (double)s->cur_clip;
    if (s->cfg.scene_goal == 3) { o[13] = s->getup_timer; o[14] = -1.0; }
    else { o[13] = s->target_hit ? 1.0 : 0.0; o[14] = s->target_hit_time; }
}
int orc_goal_dim(void* h) { return ((Scene*)h)->goal_dim(); }
// dribble_amp: ball pos(3), rot wxyz(4), vel(3), ang vel(3), ball pos at the last action(3), target-object timer time / max
void orc_ball_state(void* h, double* o) {
    Scene* s = (Scene*)h;
    o[0] = s->ball_pos.x; o[1] = s->ball_pos.y; o[2] = s->ball_pos.z; o[3] = s->ball_rot.w; o[4] = s->ball_rot.x; o[5] = s->ball_rot.y; o[6] = s->ball_rot.z;
    o[7] = s->ball_vel.x; o[8] = s->ball_vel.y; o[9] = s->ball_vel.z; o[10] = s->ball_w.x; o[11] = s->ball_w.y; o[12] = s->ball_w.z;
    o[13] = s->prev_ball_pos.x; o[14] = s->prev_ball_pos.y; o[15] = s->prev_ball_pos.z; o[16] = s->obj_timer; o[17] = s->obj_timer_max;
}
void orc_set_ball(void* h, const double* o) {
    Scene* s = (Scene*)h;
    s->ball_pos = V3((real)o[0], (real)o[1], (real)o[2]); s->ball_rot = Q4((real)o[3], (real)o[4], (real)o[5], (real)o[6]);
    s->ball_vel = V3((real)o[7], (real)o[8], (real)o[9]); s->ball_w = V3((real)o[10], (real)o[11], (real)o[12]);
}
int orc_maybe_recovery_reset(void* h, double max_time) { return ((Scene*)h)->maybe_recovery_reset(max_time) ? 1 : 0; }
void orc_amp_obs_expert_clip(void* h, int clip, double t, double ground_h, double* out) { ((Scene*)h)->amp_obs_expert(t, out, clip, ground_h); }

// Counter-based reset draws of the device path (dm_rand01 in deepmimic_amd/csrc/dm_device.h, host mirror
// deepmimic_amd/streams.py reset_rand01): splitmix64 of (seed, global env id, episode, stream)
static double rand01_(uint64_t seed, uint64_t env, uint64_t episode, uint64_t stream) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (env * 0x100000001B3ull + episode * 0xD6E8FEB86659FD93ull + stream + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
// The throughput workload of bench.py on the CPU: open-loop tracking WITH auto-reset (an env whose episode ended -- fall, motion
// end, timer -- restarts at a drawn clip time), the same episode mixture the GPU line runs.  Returns wall seconds;
// stats[0] = resets, stats[1] = reward sum, stats[2] = live steps.
double orc_rollout_auto_reset(void* h, int steps, int updates_per_step, double dt, uint64_t seed, int env_id, double t0,
                              double time_lim_min, double time_lim_max, double* stats) {
    Scene* s = (Scene*)h;
    std::vector<double> a(s->A, 0.0), kp(s->sk.P), kv(s->sk.P);
    uint64_t ep = 0; double resets = 0, rsum = 0, live = 0;
    auto draw_timer = [&](uint64_t e) { return (time_lim_max > time_lim_min) ? time_lim_min + (time_lim_max - time_lim_min) * rand01_(seed, (uint64_t)env_id, e, 1) : time_lim_max; };
    s->reset(t0, draw_timer(ep)); ++ep;
    auto t_beg = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) {
        orc_kin_eval(h, s->kin.time, kp.data(), kv.data()); orc_pose_to_action(h, kp.data(), a.data()); s->set_action(a.data());
        for (int u = 0; u < updates_per_step; ++u) { s->update(dt); if (s->is_episode_end() || !s->check_valid_episode()) break; }   // the driver ends an episode at the update where it is over or invalid (DeepMimic.py:62-80)
        double r = s->calc_reward(); rsum += r; live += (r != 0.0);
        double st[512]; s->record_state(st);
        if (s->is_episode_end() || !s->check_valid_episode()) { s->reset(s->mo.duration() * rand01_(seed, (uint64_t)env_id, ep, 0), draw_timer(ep)); ++ep; resets += 1; }
    }
    auto t_end = std::chrono::steady_clock::now();
    if (stats) { stats[0] = resets; stats[1] = rsum; stats[2] = live; }
    return std::chrono::duration<double>(t_end - t_beg).count();
}

}  // extern "C"
