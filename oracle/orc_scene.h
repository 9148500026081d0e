// TEST INFRASTRUCTURE ONLY -- CPU oracle (see orc_math.h header).
//
// orc_scene.h: the imitate scene -- controller, SPD torques, rigid-body step, reward,
// observation, termination, reset.  DeepMimic-side functions are restated from the cited
// reference lines.  The rigid-body step stands in for Bullet 2.88 (absent, un-vendored):
// it is the "DM-physics v1" specification written down in DESIGN.md section 4, which follows
// the published behaviour of btMultiBodyDynamicsWorld as recorded in SURVEY.md Appendix C
// ([EXT-BULLET]); trajectory-level parity against real Bullet is UNPINNED.
#pragma once
#include <cstdint>
#include "orc_rbd.h"

namespace orc {

struct SceneCfg {
    int num_sim_substeps = 2;            // args/run_humanoid3d_walk_args.txt:4
    double world_scale = 4.0;            // :5  (only shrinks Bullet's absolute tolerances)
    V3 gravity = V3(0, (real)-9.8, 0);   // util/MathUtil.h:25
    bool sync_char_root_pos = true, sync_char_root_rot = false;
    bool enable_fall_end = true, enable_char_contact_fall = true, enable_root_rot_fail = false;
    bool enable_rand_char_placement = true;
    bool enable_phase_input = false, record_world_root_pos = false, record_world_root_rot = false;
    double query_rate = 30.0;            // sim/CtController.cpp:8,165
    // --- `--scene imitate_amp` (scenes/SceneImitateAMP.cpp): reward 0, terminate on fall only, AMP observations
    bool scene_amp = false;
    bool enable_amp_obs_local_root = false;   // SceneImitateAMP.cpp:30,42
    // --- goal-conditioned AMP task scenes: 1 = target_amp (scenes/SceneTargetAMP.cpp), 2 = heading_amp (scenes/SceneHeadingAMP.cpp),
    //     3 = heading_amp_getup (scenes/SceneHeadingAMPGetup.cpp), 4 = strike_amp (scenes/SceneStrikeAMP.cpp)
    int scene_goal = 0;
    bool mode_test = false;                   // cRLScene::eModeTest (getup: falls start a get-up instead of ending; strike: test reward)
    // heading_amp_getup (SceneHeadingAMPGetup.cpp:58-83): getup_time = longest get-up clip (CalcGetupTime :266-291)
    double getup_time = 0, getup_height_root = 0.5, getup_height_head = 0.5, recover_episode_prob = 0; int head_id = 0; uint32_t getup_clip_mask = 0;
    // strike_amp (SceneStrikeAMP.cpp:189-231)
    double tar_near_dist = 1.4, tar_far_prob = 0.4, target_radius = 0.2, target_hit_reset_time = 2, init_hit_prob = 0, hit_tar_speed = 1.5, tar_reward_scale = 2;
    double target_min[3] = {-0.5, 1.2, 0.6}, target_max[3] = {0.5, 1.4, 1.1}; uint32_t strike_mask = 0, fail_tar_mask = 0;
    // 5 = dribble_amp (scenes/SceneDribbleAMP.cpp:124-149, 398-420): a free rigid sphere ("DM-physics v1 + one free body")
    double tar_obj_time_min = 100, tar_obj_time_max = 200, min_tar_obj_dist = 0.5, max_tar_obj_dist = 10;
    double ball_radius = 0.2, ball_mass = 0.43, ball_friction = 0.4 * 0.9, ball_lin_damping = 0.4, ball_ang_damping = 0.4;
    bool enable_rand_rot_reset = false;       // SceneImitate.cpp:131,145,184-189
    double rand_target_time_min = 1, rand_target_time_max = 5, max_target_dist = 3, target_succ_dist = 0.5;
    double tar_fail_dist = std::numeric_limits<double>::infinity(), tar_speed = 1, pos_reward_scale = 1;
    bool enable_min_tar_vel = false;
    double max_heading_turn_rate = 0.15, sharp_turn_prob = 0.025, speed_change_prob = 0.1, tar_speed_min = 1, tar_speed_max = 1, vel_reward_scale = 1;
    // --- random perturbations (cSceneSimChar::tPerturbParams, scenes/SceneSimChar.cpp:41-51; keys :92-99); part mask 0 = any body part
    bool enable_rand_perturbs = false;
    double perturb_time_min = std::numeric_limits<double>::infinity(), perturb_time_max = std::numeric_limits<double>::infinity();
    double min_perturb = 50, max_perturb = 100, min_perturb_duration = 0.1, max_perturb_duration = 0.5; uint32_t perturb_part_mask = 0;
    // --- DM-physics v1 constants [EXT-BULLET, SURVEY App. C] ---
    double friction = 0.9 * 0.9;         // link 0.9 (SimCharacter.cpp:26) x ground 0.9 (Ground.cpp:14-27)
    double erp = 0.2;                    // btContactSolverInfo::m_erp2
    int solver_iters = 10;               // btContactSolverInfo::m_numIterations
    int max_contacts = 20;               // manifold reduction: deepest-first cap
    bool enable_self_collision = true;   // btMultiBody::m_hasSelfCollision default; parent-child pairs excluded (SimCharacter.cpp:857,873,919)
    double contact_report_dist = 0.001;  // sim/ContactManager.cpp:80 (0.001*scale in scaled units)
    double breaking_factor = 0.02;       // gContactBreakingThreshold x angular-motion disc
    double max_coord_vel = 100.0;        // btMultiBody::m_maxCoordinateVelocity (scaled units)
    // --- DM-physics v2 (physics = 2; [EXT-BULLET, recalled] SURVEY App. C items 4 and 7; v1 stays the default):
    //   * both unilateral rows of every revolute limit (btMultiBodyJointLimitConstraint: q - lo and hi - q), not only the nearer bound;
    //   * link-vs-ground contacts through a PERSISTENT manifold per link (btPersistentManifold): every narrowphase call (= substep)
    //     refreshes the cached points (dropped beyond the breaking threshold in distance or in lateral drift), then adds ONE new point,
    //     the support point of the hull along -n (btConvexPlaneCollisionAlgorithm), replacing the cached point it coincides with or,
    //     when four are cached, the one the area heuristic picks (sortCachedPoints, deepest point kept).  A box settling on a face
    //     therefore gathers its four corners over four substeps instead of starting with all of them.
    //   Collision margins (0.04 scaled): sphere / capsule margins are their radii and btBoxShape's support vertex is the corner of the
    //   full extents, so link-vs-plane contacts do not see them; self collision stays v1 (capsule pairs, one point per pair).
    int physics = 1;
};

struct LinkState { Xf joint; V3 com; M3 Rb; V3 w; V3 vj; V3 vcom; };

// World-frame link kinematics from joint-space (pose, vel): what the reference reads back from Bullet in
// cSimCharacter::PostUpdate (SimCharacter.cpp:112-122,1219-1256) and cSimObj::GetPos/GetRotation.
static inline void calc_links(const Skeleton& sk, const Vec& pose, const Vec& vel, std::vector<LinkState>& L) {
    L.resize(sk.J);
    for (int j = 0; j < sk.J; ++j) {
        int par = sk.parent(j);
        Xf cp = child_parent_trans(sk, pose, j);
        L[j].joint = (par == -1) ? cp : L[par].joint * cp;
        Xf bw = L[j].joint * body_joint_trans(sk, j);
        L[j].com = bw.t; L[j].Rb = bw.R;
        int off = sk.offset(j);
        if (par == -1) { L[j].w = root_ang_vel(vel); L[j].vj = root_vel(vel); }
        else {
            V3 wl;
            if (sk.type(j) == JT_SPHERICAL) wl = V3(vel[off], vel[off + 1], vel[off + 2]);
            else if (sk.type(j) == JT_REVOLUTE) wl = V3(0, 0, vel[off]);
            L[j].w = L[par].w + L[j].joint.R * wl;
            L[j].vj = L[par].vj + cross(L[par].w, L[j].joint.t - L[par].joint.t);
        }
        L[j].vcom = L[j].vj + cross(L[j].w, L[j].com - L[j].joint.t);
    }
}

struct LDLT {
    int n = 0, m = 0; std::vector<int> idx; std::vector<real> L, D;
    void factor(const std::vector<real>& A, int n_) {
        n = n_; idx.clear();
        for (int i = 0; i < n; ++i) if (A[(size_t)i * n + i] != 0) idx.push_back(i);
        m = (int)idx.size(); L.assign((size_t)m * m, 0); D.assign(m, 0);
        for (int i = 0; i < m; ++i) for (int j = 0; j <= i; ++j) {
            real s = A[(size_t)idx[i] * n + idx[j]];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * m + k] * L[(size_t)j * m + k] * D[k];
            if (i == j) { D[i] = s; L[(size_t)i * m + i] = 1; } else L[(size_t)i * m + j] = s / D[j];
        }
    }
    void solve(const Vec& b, Vec& x) const {
        Vec y(m, 0);
        for (int i = 0; i < m; ++i) { real s = b[idx[i]]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * m + k] * y[k]; y[i] = s; }
        for (int i = 0; i < m; ++i) y[i] /= D[i];
        for (int i = m - 1; i >= 0; --i) { real s = y[i]; for (int k = i + 1; k < m; ++k) s -= L[(size_t)k * m + i] * y[k]; y[i] = s; }
        x.assign(n, 0);
        for (int i = 0; i < m; ++i) x[idx[i]] = y[i];
    }
};

struct V3d { double x = 0, y = 0, z = 0; };
struct ContactPt { int link; V3 x; real dist; int link_b = -1; V3 n = V3(0, 1, 0); int ball = 0; };   // link_b >= 0: self contact, n points from link_b to link; ball 1: body b is the ball (link vs ball), 2: body a is the ball (ball vs ground, link unused)
struct Row { Vec J; Vec W; real b, lo, hi, lam; int normal_row; real mu; real Jb[6] = {0, 0, 0, 0, 0, 0}, Wb[6] = {0, 0, 0, 0, 0, 0}; };   // Jb / Wb: the ball's 3 linear + 3 angular velocities

enum Terminate { TERM_NULL = 0, TERM_FAIL = 1, TERM_SUCC = 2 };   // scenes/RLScene.h:18-24

struct Scene {
    Skeleton sk; Motion mo; SceneCfg cfg; KinChar kin;
    RBDModel rbd_ctrl;   // SPD model: DeepMimic inertias, reference root cj
    RBDModel rbd_sim;    // simulator model: Bullet inertias, exact root cj
    Vec Kp, Kd;          // per pose slot (ImpPDController.cpp:99-120)
    std::vector<int> fall_mask;
    Vec joint_w;         // normalised DiffWeight (SceneImitate.cpp:236-248)
    std::vector<int> act_off, act_size;  // cCtController::BuildCtrlParamOffset (CtController.cpp:183-195)
    int A = 0, S = 0;

    // dynamic state
    Vec pose, vel;       // sim character (joint space, reference pose/vel layout)
    Vec tar_pose;        // PD targets (root slots unused)
    Vec tau;             // last SPD torque (pose layout)
    Vec prev_pose, prev_vel;   // cSceneImitateAMP::mPrevPose / mPrevVel (SceneImitateAMP.cpp:152-171)
    double ctrl_time = 0, init_time_offset = 0;   // cDeepMimicCharController::mTime, cCtController::mInitTimeOffset
    bool need_new_action = true;
    double timer_time = 0, timer_max = std::numeric_limits<double>::infinity();
    std::vector<int> in_contact;         // per link: ground contact within contact_report_dist
    std::vector<LinkState> links;
    // debug taps for component-level parity tests
    std::vector<ContactPt> dbg_contacts; int dbg_num_rows = 0; Vec dbg_vstar;
    // physics 2: per link the cached manifold points -- the point on the link in body coordinates, the point on the plane (x, z)
    struct ManifoldPt { V3 lp; real bx, bz; real dist; };
    std::vector<std::vector<ManifoldPt>> manifolds;
    // ---- goal scenes (cSceneTargetAMP / cSceneHeadingAMP members) and multi-clip datasets (cClipsController)
    std::vector<Motion> clips; std::vector<double> clip_cdf; int cur_clip = 0;
    V3d tar_pos; double tar_heading = 0, tar_speed = 1, tar_timer = 0, tar_timer_max = 0;
    V3d prev_action_com; double prev_action_time = 0;      // cDeepMimicCharController::mPrevActionCOM / mPrevActionTime
    uint64_t rng_seed = 0, rng_env = 0, goal_draws = 0;    // the device path's counter-based generator (dm_rand01, stream 2)
    double getup_timer = 0;                                // cSceneHeadingAMPGetup::mGetupTimer (time; max = cfg.getup_time)
    bool target_hit = false; double target_hit_time = -1;  // cSceneStrikeAMP::mTargetHit / mTargetHitTime (gInvalidHitTime = -1)
    // random perturbations: tPerturbParams::mTimer / mNextTime, the entries of cWorld's cPerturbManager (sim/Perturb.h: force at the part's COM)
    struct Perturb { int link; V3d f; double dur, time; };
    double pert_timer = 0, pert_next = std::numeric_limits<double>::infinity(); uint64_t pert_draws = 0; std::vector<Perturb> perts;
    // dribble_amp: the ball (cSimSphere), cSceneDribbleAMP::mAgentPrevTarObjPos, mTarObjTimer
    V3 ball_pos, ball_vel, ball_w; Q4 ball_rot; V3d prev_ball_pos; double obj_timer = 0, obj_timer_max = 0;

    void init(const double* jm, const double* bd, int J, const double* pd /*J x 2*/, const double* frames, int F, bool loop,
              const int* fall, const SceneCfg& c) {
        cfg = c;
        sk.init(jm, bd, J);
        mo.load(sk, frames, F, sk.P, loop);
        kin.sk = &sk; kin.mo = &mo; kin.pose.assign(sk.P, 0); kin.vel.assign(sk.P, 0);
        RBDOpts o0; o0.inertia_model = 0; o0.exact_root_cj = false; o0.world_scale = cfg.world_scale;
        RBDOpts o1; o1.inertia_model = 1; o1.exact_root_cj = true; o1.world_scale = cfg.world_scale;
        rbd_ctrl.init(&sk, o0, cfg.gravity); rbd_sim.init(&sk, o1, cfg.gravity);
        Kp.assign(sk.P, 0); Kd.assign(sk.P, 0);
        for (int j = 1; j < J; ++j)   // cExpPDController::Init skips the root (ExpPDController.cpp:22-32)
            for (int k = 0; k < sk.size(j); ++k) { Kp[sk.offset(j) + k] = (real)pd[j * 2]; Kd[sk.offset(j) + k] = (real)pd[j * 2 + 1]; }
        fall_mask.assign(fall, fall + J);
        joint_w.assign(J, 0);
        real sum = 0; for (int j = 0; j < J; ++j) { joint_w[j] = (real)sk.jd(j, JD_DIFF_W); sum += std::fabs(joint_w[j]); }
        for (int j = 0; j < J; ++j) joint_w[j] /= sum;
        act_off.assign(J, 0); act_size.assign(J, 0);
        int off = 0;
        for (int j = 0; j < J; ++j) {
            int sz = 0;
            if (j != 0) sz = (sk.type(j) == JT_SPHERICAL) ? 3 : sk.size(j);   // CtCtrlUtil.cpp:10-35
            act_off[j] = off; act_size[j] = sz; off += sz;
        }
        A = off;
        S = (cfg.enable_phase_input ? 1 : 0) + (J * 9 + 1) + J * 6;   // CtController.cpp:41-46,300-314
        if (cfg.scene_goal == 5) S += 15;                             // cSceneDribbleAMP::GetTaskStateSize (:541-545)
        pose.assign(sk.P, 0); vel.assign(sk.P, 0); tau.assign(sk.P, 0);
        tar_pose.assign(sk.P, 0);
        for (int j = 1; j < J; ++j) if (sk.type(j) == JT_SPHERICAL) tar_pose[sk.offset(j)] = 1;  // PDController.cpp:425-443
        in_contact.assign(J, 0);
        reset(0.0, std::numeric_limits<double>::infinity());
    }

    // ------------------------------------------------------------------ reset
    // cSceneSimChar::ResetScene (SceneSimChar.cpp:628-644) + cSceneImitate::ResetCharacters (SceneImitate.cpp:320-368)
    // clip / yaw: multi-clip datasets (cClipsController::Reset -> SelectNewMotion) and enable_rand_rot_reset (ResetKinChar, :331-349)
    void reset(double kin_time, double max_time, int clip = 0, real yaw = 0) {
        // cSceneDribbleAMP::Reset (:160-167): mTarObjTimer.Reset(); ResetTarObjs() -- around the root of the PREVIOUS episode's last
        // state -- ; ResetAgentTarObjRecord(); then cSceneTargetAMP::Reset
        if (cfg.scene_goal == 5) { obj_timer_reset(); reset_tar_objs(); }
        timer_time = 0; timer_max = max_time;                       // cTimer::Reset (Timer.cpp:55-73)
        if (cfg.enable_rand_perturbs) { reset_rand_perturb(); perts.clear(); }   // ResetScene (SceneSimChar.cpp:628-644): ResetRandPertrub; ResetWorld -> cWorld::Reset -> mPerturbManager.Clear()
        if (!clips.empty()) { cur_clip = clip; kin.mo = &clips[clip]; }
        // ResetKinChar: origin rot/pos reset, time := rand_time, Pose(t)
        kin.origin_rot = Q4(); kin.origin = V3(); kin.time = kin_time; kin.do_pose();
        if (yaw != 0) kin.rotate_origin(quat_axis_angle(V3(0, 1, 0), yaw));   // RotateOrigin(EulerToQuaternion(0, theta, 0))
        // SyncCharacters: sim.SetPose/SetVel(kin); ctrl.SetInitTime(kin_time)
        set_sim_state(kin.pose, kin.vel);
        ctrl_time = kin_time; init_time_offset = -kin_time;         // CtController.cpp:144-150
        need_new_action = true;                                      // DeepMimicCharController.cpp:220-229
        std::fill(tau.begin(), tau.end(), (real)0);
        std::fill(in_contact.begin(), in_contact.end(), 0);          // cWorld::Reset -> contact manager reset
        manifolds.assign(sk.J, std::vector<ManifoldPt>());           // cWorld::Reset clears the broadphase pair cache (World.cpp:81-90): manifolds start empty
        // InitCharacterPos -> SetCharRandPlacement on a plane: root x,z := 0, y kept, rot kept (Ground.cpp:154-159)
        if (cfg.enable_rand_char_placement) { pose[0] = 0; pose[2] = 0; }
        // ResolveCharGroundIntersect (SceneSimChar.cpp:542-583): lift so that min link-AABB >= ground + 1 mm
        calc_links(sk, pose, vel, links);
        real min_violation = 0;
        for (int j = 0; j < sk.J; ++j) if (sk.valid_body(j)) min_violation = std::min(min_violation, link_aabb_min_y(j) - (real)0.001);
        if (min_violation < 0) pose[1] += -min_violation;
        // SyncKinCharRoot (SceneImitate.cpp:401-418)
        if (cfg.sync_char_root_rot) {
            real dh = calc_heading(root_rot(pose)) - calc_heading(root_rot(kin.pose));
            kin.rotate_root(quat_axis_angle(V3(0, 1, 0), dh));
        }
        kin.set_root_pos_(root_pos(pose));
        calc_links(sk, pose, vel, links);
        init_hist();
        if (cfg.scene_goal) goal_reset();
    }
    // cSceneImitateAMP::InitHist (SceneImitateAMP.cpp:152-164): history := kin character one control period before the
    // controller time, origin transform included (cKinCharacter::CalcPose / CalcVel)
    void init_hist() {
        double prev_time = ctrl_time - 1.0 / cfg.query_rate;
        kin.calc_pose(prev_time, prev_pose);
        kin.calc_vel(prev_time, prev_vel);
    }
    // what cSimCharacter::SetPose/SetVel followed by BuildPose/BuildVel leave in mPose/mVel
    void set_sim_state(const Vec& p, const Vec& v) {
        pose = p; vel = v;
        post_process_pose(sk, pose);
        for (int j = 1; j < sk.J; ++j) if (sk.type(j) == JT_SPHERICAL) {
            set_joint_quat(pose, sk.offset(j), standardize(joint_quat(pose, sk.offset(j))));   // SimBodyJoint.cpp:374-383
            vel[sk.offset(j) + 3] = 0;
        }
        vel[6] = 0;
        calc_links(sk, pose, vel, links);
    }
    // [EXT-BULLET] btCollisionShape::getAabb of the link collider, lower y bound (cSimObj::CalcAABB, SimObj.cpp:224-236)
    real link_aabb_min_y(int j) const {
        const LinkState& l = links[j];
        real p0 = (real)sk.bdv(j, BD_P0), p1 = (real)sk.bdv(j, BD_P1), p2 = (real)sk.bdv(j, BD_P2);
        V3 he;
        switch (sk.shape(j)) {
            case SH_SPHERE: return l.com.y - (real)0.5 * p0;
            case SH_BOX: he = V3((real)0.5 * p0, (real)0.5 * p1, (real)0.5 * p2); break;
            case SH_CAPSULE: he = V3((real)0.5 * p0, (real)0.5 * p0 + (real)0.5 * p1, (real)0.5 * p0); break;
            default: assert(false);
        }
        return l.com.y - (std::fabs(l.Rb.m[1][0]) * he.x + std::fabs(l.Rb.m[1][1]) * he.y + std::fabs(l.Rb.m[1][2]) * he.z);
    }

    // ------------------------------------------------------------------ action
    // cCtPDController::ApplyAction -> SetPDTargets -> ConvertActionToTargetPose (CtPDController.cpp:97-166)
    void set_action(const double* a) {
        const real max_len = 2 * kPi;   // cCtCtrlUtil::gMaxPDExpVal
        for (int j = 1; j < sk.J; ++j) {
            int off = sk.offset(j), ao = act_off[j];
            if (sk.type(j) == JT_SPHERICAL) {
                V3 e((real)a[ao], (real)a[ao + 1], (real)a[ao + 2]);
                real len = norm(e);
                if (len > max_len) e = e * (max_len / len);
                Q4 q = exp_map_to_quat(e);
                real n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;   // cPDController::PostProcessTargetPose
                if (n2 == 0) q = Q4(1, 0, 0, 0); else q = qnormalized(q);
                set_joint_quat(tar_pose, off, q);
            } else {
                for (int k = 0; k < sk.size(j); ++k) tar_pose[off + k] = (real)a[ao + k];
            }
        }
    }

    // pose as the controller / reward read it: revolute angles normalised (SimBodyJoint.cpp:350-352)
    Vec reported_pose() const {
        Vec p = pose;
        for (int j = 1; j < sk.J; ++j) if (sk.type(j) == JT_REVOLUTE) p[sk.offset(j)] = normalize_angle(p[sk.offset(j)]);
        return p;
    }

    // ------------------------------------------------------------------ SPD
    // cImpPDController::CalcControlForces (ImpPDController.cpp:136-195) + joint torque clamp (SimBodyJoint.cpp:299-307,636-695)
    void calc_spd_tau(double dt_, Vec& out_tau) {
        const int P = sk.P; real t = (real)dt_;
        Vec rp = reported_pose();
        rbd_ctrl.update(rp, vel);
        std::vector<real> M = rbd_ctrl.H;
        for (int i = 0; i < P; ++i) M[(size_t)i * P + i] += t * Kd[i];
        Vec inc; vel_to_pose_diff(sk, rp, vel, inc);
        for (int i = 0; i < P; ++i) inc[i] = rp[i] + t * inc[i];
        post_process_pose(sk, inc);
        Vec pose_err; calc_vel(sk, inc, tar_pose, 1, pose_err);
        Vec acc(P, 0);
        for (int i = 0; i < P; ++i) acc[i] = Kp[i] * pose_err[i] + Kd[i] * (0 - vel[i]) - rbd_ctrl.C[i];
        Vec sol; ldlt_solve(M, P, acc, sol);
        out_tau.assign(P, 0);
        for (int i = 0; i < P; ++i) out_tau[i] = Kp[i] * pose_err[i] + Kd[i] * ((0 - vel[i]) - t * sol[i]);
        // cSimCharacter::ApplyControlForces skips the root; joint.ApplyTau clamps the torque *norm*
        for (int i = 0; i < 7; ++i) out_tau[i] = 0;
        for (int j = 1; j < sk.J; ++j) {
            int off = sk.offset(j); real lim = (real)sk.jd(j, JD_TORQUE_LIM);
            if (sk.type(j) == JT_SPHERICAL) {
                real mag = std::sqrt(out_tau[off] * out_tau[off] + out_tau[off + 1] * out_tau[off + 1] + out_tau[off + 2] * out_tau[off + 2]);
                if (mag > lim) for (int k = 0; k < 3; ++k) out_tau[off + k] *= lim / mag;
                out_tau[off + 3] = 0;
            } else if (sk.type(j) == JT_REVOLUTE) {
                real mag = std::fabs(out_tau[off]);
                if (mag > lim) out_tau[off] *= lim / mag;
            }
        }
    }

    // ------------------------------------------------------------------ rigid-body substep (DM-physics v1)
    void contact_candidates(std::vector<ContactPt>& out) const {
        out.clear();
        for (int j = 0; j < sk.J; ++j) {
            if (!sk.valid_body(j)) continue;
            const LinkState& l = links[j];
            real p0 = (real)sk.bdv(j, BD_P0), p1 = (real)sk.bdv(j, BD_P1), p2 = (real)sk.bdv(j, BD_P2);
            switch (sk.shape(j)) {
                case SH_SPHERE: { real r = (real)0.5 * p0; ContactPt c; c.link = j; c.x = l.com - V3(0, r, 0); c.dist = c.x.y; out.push_back(c); break; }
                case SH_CAPSULE: {
                    real r = (real)0.5 * p0, hh = (real)0.5 * p1;
                    for (int s = 0; s < 2; ++s) {
                        V3 e = l.com + l.Rb * V3(0, s ? -hh : hh, 0);
                        ContactPt c; c.link = j; c.x = e - V3(0, r, 0); c.dist = c.x.y; out.push_back(c);
                    }
                    break;
                }
                case SH_BOX: {
                    for (int s = 0; s < 8; ++s) {
                        V3 loc((s & 1) ? (real)-0.5 * p0 : (real)0.5 * p0, (s & 2) ? (real)-0.5 * p1 : (real)0.5 * p1, (s & 4) ? (real)-0.5 * p2 : (real)0.5 * p2);
                        ContactPt c; c.link = j; c.x = l.com + l.Rb * loc; c.dist = c.x.y; out.push_back(c);
                    }
                    break;
                }
                default: assert(false);
            }
        }
    }
    // ---- physics 2: persistent link-vs-ground manifolds [EXT-BULLET, recalled: btPersistentManifold::refreshContactPoints,
    // btConvexPlaneCollisionAlgorithm::collideSingleContact, btPersistentManifold::getCacheEntry / sortCachedPoints (2.88)]
    void manifold_update(const std::vector<ContactPt>& cand) {
        if ((int)manifolds.size() != sk.J) manifolds.assign(sk.J, std::vector<ManifoldPt>());
        for (int j = 0; j < sk.J; ++j) {
            if (!sk.valid_body(j)) continue;
            std::vector<ManifoldPt>& mf = manifolds[j];
            const real thr = breaking_threshold(j);
            const LinkState& l = links[j];
            // refresh: world position of the point on the link, distance along n = +y, lateral drift against the point on the plane
            for (int i = (int)mf.size() - 1; i >= 0; --i) {
                V3 xa = l.com + l.Rb * mf[i].lp;
                mf[i].dist = xa.y;
                const real dx = mf[i].bx - xa.x, dz = mf[i].bz - xa.z;
                if (!(mf[i].dist <= thr) || dx * dx + dz * dz > thr * thr) mf.erase(mf.begin() + i);
            }
            // the new point: support vertex of the hull along -n = the deepest candidate of the link (ties: the first)
            int best = -1;
            for (size_t c = 0; c < cand.size(); ++c) if (cand[c].link == j && (best < 0 || cand[c].dist < cand[best].dist)) best = (int)c;
            if (best < 0 || !(cand[best].dist < thr)) continue;
            ManifoldPt np; np.lp = transpose(l.Rb) * (cand[best].x - l.com); np.bx = cand[best].x.x; np.bz = cand[best].x.z; np.dist = cand[best].dist;
            // getCacheEntry: the cached point nearest to the new one in body coordinates, if closer than the breaking threshold
            int slot = -1; real shortest = thr * thr;
            for (size_t i = 0; i < mf.size(); ++i) { V3 d = mf[i].lp - np.lp; real d2 = dot(d, d); if (d2 < shortest) { shortest = d2; slot = (int)i; } }
            if (slot >= 0) mf[slot] = np;
            else if (mf.size() < 4) mf.push_back(np);
            else mf[manifold_sort(mf, np)] = np;
        }
    }
    // sortCachedPoints: which of the four cached points the new one replaces -- never the deepest; of the others the one whose removal
    // leaves the largest quadrilateral (squared cross product of the diagonals)
    static int manifold_sort(const std::vector<ManifoldPt>& mf, const ManifoldPt& np) {
        int deepest = -1; real maxpen = np.dist;
        for (int i = 0; i < 4; ++i) if (mf[i].dist < maxpen) { deepest = i; maxpen = mf[i].dist; }
        real res[4] = {0, 0, 0, 0};
        auto area = [&](const V3& a, const V3& b) { V3 c = cross(a, b); return dot(c, c); };
        if (deepest != 0) res[0] = area(np.lp - mf[1].lp, mf[3].lp - mf[2].lp);
        if (deepest != 1) res[1] = area(np.lp - mf[0].lp, mf[3].lp - mf[2].lp);
        if (deepest != 2) res[2] = area(np.lp - mf[0].lp, mf[3].lp - mf[1].lp);
        if (deepest != 3) res[3] = area(np.lp - mf[0].lp, mf[2].lp - mf[1].lp);
        int best = 0;                                         // btVector4::closestAxis4: index of the largest absolute value, first on ties
        for (int i = 1; i < 4; ++i) if (res[i] > res[best]) best = i;
        return best;
    }
    // the manifolds' points as contact candidates, in (link, slot) order
    void manifold_points(std::vector<ContactPt>& out) const {
        for (int j = 0; j < sk.J; ++j) {
            if (!sk.valid_body(j)) continue;
            const LinkState& l = links[j];
            for (const ManifoldPt& m : manifolds[j]) { ContactPt c; c.link = j; c.x = l.com + l.Rb * m.lp; c.dist = m.dist; out.push_back(c); }
        }
    }
    // [EXT-BULLET] contact breaking threshold = 0.02 x getAngularMotionDisc() of the convex shape
    real breaking_threshold(int j) const {
        real p0 = (real)sk.bdv(j, BD_P0), p1 = (real)sk.bdv(j, BD_P1), p2 = (real)sk.bdv(j, BD_P2);
        real rad = 0;
        switch (sk.shape(j)) {
            case SH_SPHERE: rad = (real)0.5 * p0; break;
            case SH_CAPSULE: rad = (real)0.5 * p0 + (real)0.5 * p1; break;
            case SH_BOX: rad = (real)0.5 * std::sqrt(p0 * p0 + p1 * p1 + p2 * p2); break;
            default: assert(false);
        }
        return (real)cfg.breaking_factor * rad;
    }
    // ---- self collision, capsule model: every link is a segment (body frame, c0 -> c1) swept by a sphere of radius r.
    // sphere: point; capsule: its own axis; box: inscribed capsule along the longest extent.
    void link_capsule(int j, V3& c0, V3& c1, real& r) const {
        real p0 = (real)sk.bdv(j, BD_P0), p1 = (real)sk.bdv(j, BD_P1), p2 = (real)sk.bdv(j, BD_P2);
        c0 = V3(0, 0, 0); c1 = V3(0, 0, 0); r = 0;
        switch (sk.shape(j)) {
            case SH_SPHERE: r = (real)0.5 * p0; break;
            case SH_CAPSULE: r = (real)0.5 * p0; c0 = V3(0, (real)0.5 * p1, 0); c1 = V3(0, (real)-0.5 * p1, 0); break;
            case SH_BOX: {
                real e[3] = { p0, p1, p2 };
                int a = 0; if (e[1] > e[a]) a = 1; if (e[2] > e[a]) a = 2;
                r = (real)0.5 * std::min(e[(a + 1) % 3], e[(a + 2) % 3]);
                real hl = std::max((real)0, (real)0.5 * e[a] - r);
                real v[3] = { 0, 0, 0 }; v[a] = hl;
                c0 = V3(v[0], v[1], v[2]); c1 = V3(-v[0], -v[1], -v[2]);
                break;
            }
            default: assert(false);
        }
    }
    // closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9)
    static void closest_segment_points(const V3& p1, const V3& q1, const V3& p2, const V3& q2, V3& c1, V3& c2) {
        const real eps = (real)1e-12;
        V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
        real a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), s, t;
        if (a <= eps && e <= eps) { s = t = 0; }
        else if (a <= eps) { s = 0; t = std::min((real)1, std::max((real)0, f / e)); }
        else {
            real c = dot(d1, r);
            if (e <= eps) { t = 0; s = std::min((real)1, std::max((real)0, -c / a)); }
            else {
                real b = dot(d1, d2), denom = a * e - b * b;
                s = (denom > eps) ? std::min((real)1, std::max((real)0, (b * f - c * e) / denom)) : 0;
                t = (b * s + f) / e;
                if (t < 0) { t = 0; s = std::min((real)1, std::max((real)0, -c / a)); }
                else if (t > 1) { t = 1; s = std::min((real)1, std::max((real)0, (b - c) / a)); }
            }
        }
        c1 = p1 + s * d1; c2 = p2 + t * d2;
    }
    void self_contacts(std::vector<ContactPt>& out) const {
        out.clear();
        for (int i = 0; i < sk.J; ++i) for (int j = i + 1; j < sk.J; ++j) {
            if (sk.parent(j) == i || sk.parent(i) == j) continue;          // disableParentCollision
            if (!sk.valid_body(i) || !sk.valid_body(j)) continue;
            V3 a0, a1, b0, b1; real ra, rb;
            link_capsule(i, a0, a1, ra); link_capsule(j, b0, b1, rb);
            V3 pa0 = links[i].com + links[i].Rb * a0, pa1 = links[i].com + links[i].Rb * a1;
            V3 pb0 = links[j].com + links[j].Rb * b0, pb1 = links[j].com + links[j].Rb * b1;
            V3 ca, cb; closest_segment_points(pa0, pa1, pb0, pb1, ca, cb);
            V3 dlt = ca - cb; real d = norm(dlt);
            real dist = d - ra - rb;
            if (!(dist < std::min(breaking_threshold(i), breaking_threshold(j)))) continue;
            V3 n = (d > (real)1e-9) ? ((real)1 / d) * dlt : V3(0, 1, 0);
            ContactPt c; c.link = i; c.link_b = j; c.n = n; c.dist = dist;
            c.x = (real)0.5 * ((ca - ra * n) + (cb + rb * n));
            out.push_back(c);
        }
    }
    // btPlaneSpace1
    static void plane_space(const V3& n, V3& p, V3& q) {
        if (std::fabs(n.z) > (real)0.7071067811865475244) {
            real a = n.y * n.y + n.z * n.z, k = (real)1 / std::sqrt(a);
            p = V3(0, -n.z * k, n.y * k); q = V3(a * k, -n.x * p.z, n.x * p.y);
        } else {
            real a = n.x * n.x + n.y * n.y, k = (real)1 / std::sqrt(a);
            p = V3(-n.y * k, n.x * k, 0); q = V3(-n.z * p.y, n.z * p.x, a * k);
        }
    }
    // row of a contact along d: the contact point moves with `link`; for a self contact minus the same point moving with link_b
    void contact_jacobian(const ContactPt& cp, const V3& d, Vec& Jr) const {
        if (cp.link < 0) { Jr.assign(sk.P, 0); return; }             // ball vs ground: no character dof moves the point
        point_jacobian(cp.link, cp.x, d, Jr);
        if (cp.link_b >= 0) { Vec Jb; point_jacobian(cp.link_b, cp.x, d, Jb); for (int i = 0; i < sk.P; ++i) Jr[i] -= Jb[i]; }
    }
    // generalized-velocity Jacobian row (pose layout) of d . (velocity of world point x rigidly attached to `link`)
    void point_jacobian(int link, const V3& x, const V3& d, Vec& Jr) const {
        Jr.assign(sk.P, 0);
        Jr[0] = d.x; Jr[1] = d.y; Jr[2] = d.z;
        V3 r0 = x - links[0].joint.t;
        V3 rxd = cross(r0, d);             // (e_k x r).d = e_k . (r x d)
        Jr[3] = rxd.x; Jr[4] = rxd.y; Jr[5] = rxd.z;
        for (int j = link; j > 0; j = sk.parent(j)) {
            int off = sk.offset(j);
            V3 m = cross(x - links[j].joint.t, d);
            const M3& R = links[j].joint.R;
            if (sk.type(j) == JT_SPHERICAL) for (int k = 0; k < 3; ++k) Jr[off + k] = R.m[0][k] * m.x + R.m[1][k] * m.y + R.m[2][k] * m.z;
            else if (sk.type(j) == JT_REVOLUTE) Jr[off] = R.m[0][2] * m.x + R.m[1][2] * m.y + R.m[2][2] * m.z;
        }
    }
    void clamp_coord_vel(Vec& v) const {
        real lin = (real)(cfg.max_coord_vel / cfg.world_scale), ang = (real)cfg.max_coord_vel;
        for (int i = 0; i < sk.P; ++i) { real m = (i < 3) ? lin : ang; v[i] = std::max(-m, std::min(m, v[i])); }
    }
    void substep(double h_) {
        const int P = sk.P; const real h = (real)h_;
        calc_links(sk, pose, vel, links);
        rbd_sim.update(pose, vel);
        LDLT fac; fac.factor(rbd_sim.H, P);
        Vec rhs(P, 0), acc;
        for (int i = 0; i < P; ++i) rhs[i] = tau[i] - rbd_sim.C[i];
        // external forces of the perturbations (btMultiBody::addLinkForce: world-frame force at the link's centre of mass, held over the
        // substeps of one stepSimulation call): generalized force J^T f
        for (const Perturb& pt : perts) {
            Vec Jr; point_jacobian(pt.link, links[pt.link].com, V3((real)pt.f.x, (real)pt.f.y, (real)pt.f.z), Jr);
            for (int i = 0; i < P; ++i) rhs[i] += Jr[i];
        }
        fac.solve(rhs, acc);
        Vec vstar(P, 0);
        for (int i = 0; i < P; ++i) vstar[i] = vel[i] + h * acc[i];
        clamp_coord_vel(vstar);
        dbg_vstar = vstar;
        // the ball's unconstrained velocity: btRigidBody::applyDamping then the gravity impulse
        const bool ball = cfg.scene_goal == 5;
        V3 bv, bw;
        if (ball) {
            bv = ball_vel * (real)std::pow(1.0 - cfg.ball_lin_damping, (double)h) + h * cfg.gravity;
            bw = ball_w * (real)std::pow(1.0 - cfg.ball_ang_damping, (double)h);
        }

        // collision detection at the start-of-substep pose
        std::vector<ContactPt> cand; contact_candidates(cand);
        std::fill(in_contact.begin(), in_contact.end(), 0);
        std::vector<int> act;
        if (cfg.physics == 2) { manifold_update(cand); cand.clear(); manifold_points(cand); }
        for (size_t i = 0; i < cand.size(); ++i) {
            if (cand[i].dist <= (real)cfg.contact_report_dist) in_contact[cand[i].link] = 1;
            if (cfg.physics == 2 || cand[i].dist < breaking_threshold(cand[i].link)) act.push_back((int)i);
        }
        // manifold reduction: keep the max_contacts deepest (ties -> lower candidate index)
        std::stable_sort(act.begin(), act.end(), [&](int a, int b) { return cand[a].dist < cand[b].dist; });
        if ((int)act.size() > cfg.max_contacts) act.resize(cfg.max_contacts);
        std::sort(act.begin(), act.end());
        std::vector<ContactPt> contacts;
        for (size_t i = 0; i < act.size(); ++i) contacts.push_back(cand[act[i]]);
        // self contacts: capsule model of every link, non-adjacent pairs in (i < j) order; they take the
        // slots the ground contacts left, in pair order
        if (cfg.enable_self_collision) {
            std::vector<ContactPt> self; self_contacts(self);
            for (size_t i = 0; i < self.size() && (int)contacts.size() < cfg.max_contacts; ++i) contacts.push_back(self[i]);
        }
        // ball contacts take the slots that are left: ball vs ground, then the links in index order (capsule models, as link vs link)
        if (ball) {
            const real rb = (real)cfg.ball_radius, thr_b = (real)cfg.breaking_factor * rb;
            if ((int)contacts.size() < cfg.max_contacts && ball_pos.y - rb < thr_b) {
                ContactPt c; c.link = -1; c.ball = 2; c.n = V3(0, 1, 0); c.dist = ball_pos.y - rb; c.x = ball_pos - V3(0, rb, 0);
                contacts.push_back(c);
            }
            for (int j = 0; j < sk.J && (int)contacts.size() < cfg.max_contacts; ++j) {
                if (!sk.valid_body(j)) continue;
                V3 a0, a1; real ra; link_capsule(j, a0, a1, ra);
                V3 pa0 = links[j].com + links[j].Rb * a0, pa1 = links[j].com + links[j].Rb * a1;
                V3 ca, cb; closest_segment_points(pa0, pa1, ball_pos, ball_pos, ca, cb);
                V3 dlt = ca - ball_pos; real d = norm(dlt), dist = d - ra - rb;
                if (!(dist < std::min(breaking_threshold(j), thr_b))) continue;
                V3 n = (d > (real)1e-9) ? ((real)1 / d) * dlt : V3(0, 1, 0);
                ContactPt c; c.link = j; c.ball = 1; c.n = n; c.dist = dist;
                c.x = (real)0.5 * ((ca - ra * n) + (ball_pos + rb * n));
                contacts.push_back(c);
            }
        }
        dbg_contacts = contacts;

        std::vector<Row> rows;
        const real big = (real)1e30;
        // joint-limit rows (btMultiBodyJointLimitConstraint; revolute only, SimCharacter.cpp:948-973): v1 the nearer bound, v2 both
        for (int j = 1; j < sk.J; ++j) {
            if (sk.type(j) != JT_REVOLUTE) continue;
            real lo = (real)sk.jd(j, JD_LL0), hi = (real)sk.jd(j, JD_LH0);
            if (lo > hi) continue;
            int off = sk.offset(j); real th = pose[off];
            real pen_lo = th - lo, pen_hi = hi - th;
            for (int side = 0; side < 2; ++side) {
                if (cfg.physics != 2 && side != ((pen_lo <= pen_hi) ? 0 : 1)) continue;
                // btMultiBodyConstraint::m_maxAppliedImpulse = 100 [EXT-BULLET, recalled; SURVEY App. C item 4]: an angular impulse in the scaled world's
                // units (lengths x world_scale, masses as they are) = 100 / world_scale^2 here
                Row r; r.J.assign(P, 0); r.normal_row = -1; r.mu = 0; r.lo = 0; r.hi = (real)(100.0 / (cfg.world_scale * cfg.world_scale)); r.lam = 0;
                const real pen = side ? pen_hi : pen_lo;
                r.J[off] = side ? (real)-1 : (real)1;
                r.b = (pen > 0) ? -pen / h : (real)-cfg.erp * pen / h;
                rows.push_back(r);
            }
        }
        int n_lim = (int)rows.size(), nc = (int)contacts.size();
        for (int c = 0; c < nc; ++c) {
            const ContactPt& cp = contacts[c];
            Row r; contact_jacobian(cp, cp.n, r.J); ball_jacobian(cp, cp.n, r.Jb); r.normal_row = -1; r.mu = 0; r.lo = 0; r.hi = big; r.lam = 0;
            r.b = (cp.dist > 0) ? -cp.dist / h : (real)-cfg.erp * cp.dist / h;
            rows.push_back(r);
        }
        for (int c = 0; c < nc; ++c) for (int d = 0; d < 2; ++d) {
            const ContactPt& cp = contacts[c];
            V3 t1, t2; plane_space(cp.n, t1, t2);      // btPlaneSpace1: (-1,0,0), (0,0,1) for the ground normal
            Row r; contact_jacobian(cp, d ? t2 : t1, r.J); ball_jacobian(cp, d ? t2 : t1, r.Jb); r.normal_row = n_lim + c;
            r.mu = cp.ball ? (real)cfg.ball_friction : (real)cfg.friction; r.lo = r.hi = 0; r.lam = 0; r.b = 0;
            rows.push_back(r);
        }
        const int R = (int)rows.size(); dbg_num_rows = R;
        std::vector<real> Amat((size_t)R * R, 0); Vec cvec(R, 0);
        for (int r = 0; r < R; ++r) fac.solve(rows[r].J, rows[r].W);
        if (ball) {
            const real im = (real)1 / (real)cfg.ball_mass, ii = (real)1 / ((real)0.4 * (real)cfg.ball_mass * (real)cfg.ball_radius * (real)cfg.ball_radius);
            for (int r = 0; r < R; ++r) for (int k = 0; k < 6; ++k) rows[r].Wb[k] = (k < 3 ? im : ii) * rows[r].Jb[k];
        }
        for (int r = 0; r < R; ++r) {
            for (int s = 0; s < R; ++s) {
                real a = 0; for (int i = 0; i < P; ++i) a += rows[r].J[i] * rows[s].W[i];
                if (ball) for (int k = 0; k < 6; ++k) a += rows[r].Jb[k] * rows[s].Wb[k];
                Amat[(size_t)r * R + s] = a;
            }
            real c = 0; for (int i = 0; i < P; ++i) c += rows[r].J[i] * vstar[i];
            if (ball) c += rows[r].Jb[0] * bv.x + rows[r].Jb[1] * bv.y + rows[r].Jb[2] * bv.z + rows[r].Jb[3] * bw.x + rows[r].Jb[4] * bw.y + rows[r].Jb[5] * bw.z;
            cvec[r] = c;
        }
        // projected Gauss-Seidel: per iteration limits, normals, frictions
        for (int it = 0; it < cfg.solver_iters; ++it) {
            for (int r = 0; r < R; ++r) {
                Row& row = rows[r];
                real u = cvec[r]; for (int s = 0; s < R; ++s) u += Amat[(size_t)r * R + s] * rows[s].lam;
                real d = (row.b - u) / Amat[(size_t)r * R + r];
                real lo = row.lo, hi = row.hi;
                if (row.normal_row >= 0) { hi = row.mu * rows[row.normal_row].lam; lo = -hi; }
                real nl = row.lam + d;
                if (nl < lo) nl = lo; else if (nl > hi) nl = hi;
                row.lam = nl;
            }
        }
        Vec vnew = vstar;
        for (int r = 0; r < R; ++r) for (int i = 0; i < P; ++i) vnew[i] += rows[r].W[i] * rows[r].lam;
        clamp_coord_vel(vnew);
        vel = vnew;
        integrate(h);
        if (ball) {
            for (int r = 0; r < R; ++r) {
                bv = bv + rows[r].lam * V3(rows[r].Wb[0], rows[r].Wb[1], rows[r].Wb[2]);
                bw = bw + rows[r].lam * V3(rows[r].Wb[3], rows[r].Wb[4], rows[r].Wb[5]);
            }
            ball_vel = bv; ball_w = bw;
            ball_pos = ball_pos + h * bv;
            ball_rot = qnormalized(quat_exp(h * bw) * ball_rot);
        }
    }
    // the ball's part of a contact row along d: the contact point moving with the ball, negative when the ball is body b
    void ball_jacobian(const ContactPt& cp, const V3& d, real* Jb) const {
        for (int k = 0; k < 6; ++k) Jb[k] = 0;
        if (!cp.ball) return;
        const real sg = (cp.ball == 2) ? (real)1 : (real)-1;
        V3 rxd = cross(cp.x - ball_pos, d);
        Jb[0] = sg * d.x; Jb[1] = sg * d.y; Jb[2] = sg * d.z; Jb[3] = sg * rxd.x; Jb[4] = sg * rxd.y; Jb[5] = sg * rxd.z;
    }
    static Q4 quat_exp(const V3& rv) {     // exact exponential map of a rotation vector
        real th = norm(rv), half = (real)0.5 * th;
        real s = (th < (real)1e-6) ? ((real)0.5 - th * th / 48) : std::sin(half) / th;
        return Q4(std::cos(half), s * rv.x, s * rv.y, s * rv.z);
    }
    // [EXT-BULLET] btMultiBody::stepPositionsMultiDof: semi-implicit Euler, exponential map on rotations
    void integrate(real h) {
        for (int k = 0; k < 3; ++k) pose[k] += h * vel[k];
        set_root_rot(pose, qnormalized(quat_exp(h * root_ang_vel(vel)) * root_rot(pose)));   // world-frame omega
        for (int j = 1; j < sk.J; ++j) {
            int off = sk.offset(j);
            if (sk.type(j) == JT_SPHERICAL) {
                Q4 q = qnormalized(joint_quat(pose, off) * quat_exp(h * V3(vel[off], vel[off + 1], vel[off + 2])));   // child-frame omega
                set_joint_quat(pose, off, standardize(q));
            } else if (sk.type(j) == JT_REVOLUTE) pose[off] += h * vel[off];
        }
    }

    // ------------------------------------------------------------------ one scene update
    void update(double dt) {
        // cRLSceneSimChar::PreUpdate (RLSceneSimChar.cpp:263-275) -> cSceneImitateAMP::NewActionUpdate -> UpdateHist (:166-171)
        if (need_new_action) { prev_pose = reported_pose(); prev_vel = vel; }
        if (need_new_action && cfg.scene_goal == 5) { prev_ball_pos.x = ball_pos.x; prev_ball_pos.y = ball_pos.y; prev_ball_pos.z = ball_pos.z; }   // cSceneDribbleAMP::NewActionUpdate (:337-341)
        if (need_new_action && cfg.scene_goal) {                     // cDeepMimicCharController::HandleNewAction (DeepMimicCharController.cpp:262-267)
            calc_links(sk, pose, vel, links);
            V3 c = sim_com(); prev_action_com.x = c.x; prev_action_com.y = c.y; prev_action_com.z = c.z;
            prev_action_time = ctrl_time + dt;                       // mTime has been advanced by this update when UpdateCalcTau latches it
        }
        timer_time += dt;                                            // cScene::Update -> UpdateTimers
        if (cfg.scene_goal == 3) getup_timer += dt;                  // cSceneHeadingAMPGetup::UpdateTimers (:163-167)
        if (cfg.enable_rand_perturbs) update_rand_perturb(dt);       // cSceneSimChar::Update (:145-148), before PreUpdate
        // 4a UpdateKinChar (SceneImitate.cpp:306-318)
        double prev_phase = kin.phase();
        kin.update(dt);
        double curr_phase = kin.phase();
        if (curr_phase < prev_phase) sync_kin_new_cycle();
        // 4b sim_char->Update: controller clock, latch, SPD
        ctrl_time += dt;                                             // DeepMimicCharController.cpp:71-78
        if (need_new_action) need_new_action = false;                // HandleNewAction
        calc_spd_tau(dt, tau);
        // 5 cWorld::Update: stepSimulation(dt, n, dt/n) (World.cpp:93-104)
        // cPerturbManager::Update (PerturbManager.cpp:39-53) at the top of cWorld::Update: expired entries leave, the others advance and act
        { size_t k = 0; for (size_t i = 0; i < perts.size(); ++i) if (!(perts[i].time >= perts[i].dur)) { perts[i].time += dt; perts[k++] = perts[i]; } perts.resize(k); }
        double h = dt / cfg.num_sim_substeps;
        for (int s = 0; s < cfg.num_sim_substeps; ++s) substep(h);
        // 7 PostUpdate
        calc_links(sk, pose, vel, links);
        need_new_action = check_next_interval(dt, ctrl_time + init_time_offset, 1.0 / cfg.query_rate);   // CtController.cpp:221-227
        if (cfg.scene_goal) goal_update(dt);                         // cSceneTargetAMP::Update after cSceneImitate::Update (:137-146)
        // cSceneHeadingAMPGetup::Update (:99-107) -> UpdateTestGetup (:244-253): in test mode a fall starts a get-up
        if (cfg.scene_goal == 3 && cfg.mode_test && has_fallen_contact() && !getting_up()) getup_timer = 0;
    }

    // ------------------------------------------------------------------ random perturbations (scenes/SceneSimChar.cpp:205-256, 618-626, 952-956)
    // Draws: dm_rand01(seed, global env id, draw counter, stream 5) in the reference's call order (part, direction x y z, magnitude, duration,
    // next time); the reference's own generator is the scene's cRand.
    double pert_u01() { return rand01(rng_seed, rng_env, pert_draws++, 5); }
    double pert_uniform(double lo, double hi) { const double u = pert_u01(); return (hi > lo && hi < 1e300) ? lo + (hi - lo) * u : hi; }   // cRand::RandDouble
    void reset_rand_perturb() { pert_timer = 0; pert_next = pert_uniform(cfg.perturb_time_min, cfg.perturb_time_max); }               // ResetRandPertrub
    void update_rand_perturb(double dt) {                                                                                            // UpdateRandPerturb
        pert_timer += dt;
        if (!(pert_timer >= pert_next)) return;
        // ApplyRandForce(char 0): GetRandPerturbPartID, then a uniformly drawn direction, magnitude and duration
        int n = 0; for (int j = 0; j < sk.J; ++j) if (!cfg.perturb_part_mask || ((cfg.perturb_part_mask >> j) & 1u)) ++n;
        int idx = (int)(pert_u01() * n); if (idx >= n) idx = n - 1;                                                                  // cRand::RandInt(0, n)
        int part = -1; for (int j = 0, k = 0; j < sk.J; ++j) if (!cfg.perturb_part_mask || ((cfg.perturb_part_mask >> j) & 1u)) { if (k++ == idx) { part = j; break; } }
        const double dx = pert_uniform(-1, 1), dy = pert_uniform(-1, 1), dz = pert_uniform(-1, 1);
        const double mag = pert_uniform(cfg.min_perturb, cfg.max_perturb), dur = pert_uniform(cfg.min_perturb_duration, cfg.max_perturb_duration);
        const double sc = mag / std::sqrt(dx * dx + dy * dy + dz * dz);
        Perturb pt; pt.link = part; pt.f.x = sc * dx; pt.f.y = sc * dy; pt.f.z = sc * dz; pt.dur = dur; pt.time = 0;
        if (part >= 0 && sk.valid_body(part)) perts.push_back(pt);                                                                   // cPerturbManager::AddPerturb: valid ones only
        reset_rand_perturb();
    }

    // ------------------------------------------------------------------ goal-conditioned task scenes
    // Draws: the reference uses the scene's cRand (std::default_random_engine); this path is specified on the device's counter-based
    // generator -- dm_rand01(seed, global env id, draw counter, stream 2) -- so that oracle and device consume identical numbers.
    static double rand01(uint64_t seed, uint64_t env, uint64_t episode, uint64_t stream) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (env * 0x100000001B3ull + episode * 0xD6E8FEB86659FD93ull + stream + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z = z ^ (z >> 31);
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
    double goal_u01() { return rand01(rng_seed, rng_env, goal_draws++, 2); }
    double goal_uniform(double lo, double hi) { return lo + (hi - lo) * goal_u01(); }
    double goal_normal(double mean, double stdev) { double u1 = 1.0 - goal_u01(), u2 = goal_u01(); return mean + stdev * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
    V3 sim_com() const {                                             // cSimCharacter::CalcCOM (SimCharacter.cpp:398-416)
        V3 c; real tm = 0;
        for (int j = 0; j < sk.J; ++j) if (sk.valid_body(j)) { c += (real)sk.mass(j) * links[j].com; tm += (real)sk.mass(j); }
        return c / tm;
    }
    bool target_like() const { return cfg.scene_goal == 1 || cfg.scene_goal == 4 || cfg.scene_goal == 5; }
    // ---- dribble_amp: the ball as target object (SceneDribbleAMP.cpp:422-440, 493-508)
    void obj_timer_reset() { obj_timer = 0; obj_timer_max = goal_uniform(cfg.tar_obj_time_min, cfg.tar_obj_time_max); }
    void reset_tar_objs() {
        const double r = goal_uniform(cfg.min_tar_obj_dist, cfg.max_tar_obj_dist), theta = goal_uniform(-3.141592653589793, 3.141592653589793);
        const double px = (double)pose[0] + r * std::cos(theta), pz = (double)pose[2] + r * std::sin(theta);
        // SetTarObjPos: on the ground, a random orientation, at rest
        double ax = goal_uniform(-1, 1), ay = goal_uniform(-1, 1), az = goal_uniform(-1, 1);
        const double an = std::sqrt(ax * ax + ay * ay + az * az), th = goal_uniform(-3.141592653589793, 3.141592653589793);
        ball_pos = V3((real)px, (real)cfg.ball_radius, (real)pz);
        ball_rot = quat_axis_angle(V3((real)(ax / an), (real)(ay / an), (real)(az / an)), (real)th);
        ball_vel = V3(); ball_w = V3();
        prev_ball_pos.x = ball_pos.x; prev_ball_pos.y = ball_pos.y; prev_ball_pos.z = ball_pos.z;      // ResetAgentTarObjRecord
    }
    bool tar_obj_dist_fail() const {                                 // CheckTarObjDistFail (:351-364)
        const real dx = (real)tar_pos.x - ball_pos.x, dz = (real)tar_pos.z - ball_pos.z, thr = 2 * (real)cfg.max_target_dist;
        return dx * dx + dz * dz > thr * thr;
    }
    bool char_obj_dist_fail() const {                                // CheckCharObjDistFail (:366-379)
        const real dx = ball_pos.x - pose[0], dz = ball_pos.z - pose[2], thr = 2 * (real)cfg.max_tar_obj_dist;
        return dx * dx + dz * dz > thr * thr;
    }
    bool dribble_succ() const {                                      // CheckTargetSucc (:454-464)
        const real dx = (real)tar_pos.x - ball_pos.x, dz = (real)tar_pos.z - ball_pos.z;
        return dx * dx + dz * dz < (real)cfg.target_succ_dist * (real)cfg.target_succ_dist;
    }
    bool heading_like() const { return cfg.scene_goal == 2 || cfg.scene_goal == 3; }
    int goal_dim() const { return (cfg.scene_goal == 3 || cfg.scene_goal == 4) ? 4 : 3; }
    bool getting_up() const { return cfg.scene_goal == 3 && !(getup_timer >= cfg.getup_time); }   // CheckGettingUp (:301-304): !mGetupTimer.IsEnd()
    void goal_reset_target_pos() {
        if (cfg.scene_goal == 4) {                                   // cSceneStrikeAMP::ResetTargetPos / ...Far / ...Near (:318-374)
            const bool far = goal_u01() < cfg.tar_far_prob;
            const double theta = far ? goal_uniform(-3.141592653589793, 3.141592653589793) : goal_uniform(cfg.target_min[0], cfg.target_max[0]);
            const double hgt = goal_uniform(cfg.target_min[1], cfg.target_max[1]);
            const double dist = far ? goal_uniform(cfg.target_min[2], cfg.max_target_dist) : goal_uniform(cfg.target_min[2], cfg.target_max[2]);
            tar_pos.x = dist * std::cos(theta) + (double)pose[0]; tar_pos.y = hgt; tar_pos.z = dist * -std::sin(theta) + (double)pose[2];
            set_target_hit(false);
            return;
        }
        if (cfg.scene_goal == 5) {                                   // cSceneDribbleAMP::SampleRandTargetPos (:510-524): around the ball
            const double r = goal_uniform(cfg.ball_radius, cfg.max_target_dist), theta = goal_uniform(-3.141592653589793, 3.141592653589793);
            tar_pos.x = (double)ball_pos.x + r * std::cos(theta); tar_pos.y = 0; tar_pos.z = (double)ball_pos.z + r * std::sin(theta);
            return;
        }
        // cSceneTargetAMP::SampleRandTargetPos (:285-299)
        double dist = goal_uniform(0.0, cfg.max_target_dist), theta = goal_uniform(0.0, 6.283185307179586);
        tar_pos.x = (double)pose[0] + dist * std::cos(theta); tar_pos.y = 0; tar_pos.z = (double)pose[2] + dist * std::sin(theta);
    }
    void set_target_hit(bool hit) { if (!target_hit && hit) target_hit_time = timer_time; target_hit = hit; }   // SetTargetHit (:249-258), GetTime() = scene timer
    void goal_timer_reset() { tar_timer = 0; tar_timer_max = goal_uniform(cfg.rand_target_time_min, cfg.rand_target_time_max); }   // cTimer::Reset
    void goal_reset() {                                              // cSceneTargetAMP::Reset (:130-135) / cSceneHeadingAMP::ResetTarget (:230-239)
        goal_timer_reset();
        goal_reset_target_pos();
        if (heading_like()) { tar_heading = 0; tar_speed = std::min(std::max(goal_uniform(cfg.tar_speed_min, cfg.tar_speed_max), cfg.tar_speed_min), cfg.tar_speed_max); }
        else tar_speed = cfg.tar_speed;
        prev_action_com = V3d(); prev_action_time = ctrl_time;       // cCtController::SetInitTime, cDeepMimicCharController::ResetParams
        if (cfg.scene_goal == 3) {                                   // ResetTimers -> ResetGetupTimer (ended), then SyncGetupTimer (:190-207)
            getup_timer = cfg.getup_time;
            if ((cfg.getup_clip_mask >> cur_clip) & 1u) getup_timer = kin.time;
        }
        if (cfg.scene_goal == 4) {                                   // cSceneStrikeAMP::ResetTarget (:301-316) after the base ResetTarget
            if (!cfg.mode_test && cfg.init_hit_prob > 0) set_target_hit(goal_u01() < cfg.init_hit_prob);      // ResetTargetHit (:376-383)
            target_hit_time = target_hit ? goal_uniform(timer_time - cfg.target_hit_reset_time, timer_time) : -1.0;
        }
    }
    // cSceneHeadingAMPGetup::Reset (:109-121): in train mode an episode that ended in a fall continues, with probability
    // recover_episode_prob, as a recovery episode -- ResetRecoveryEpisode (:40-56): timers and controller only, the characters stay
    // where they are.  (`bool mIsRecoveryEpisode = ...` declares a local there, so the member stays false: recovery episodes chain.)
    bool maybe_recovery_reset(double max_time) {
        if (cfg.scene_goal != 3 || cfg.mode_test || !(cfg.recover_episode_prob > 0)) return false;
        if (check_terminate() != TERM_FAIL) return false;
        if (!(goal_u01() < cfg.recover_episode_prob)) return false;
        timer_time = 0; timer_max = max_time;                        // ResetTimers
        getup_timer = 0;                                             // ResetGetupTimer then BeginGetup
        ctrl_time = 0; init_time_offset = 0; need_new_action = true; // cCtController::ResetParams (CtController.cpp:103-108)
        std::fill(tau.begin(), tau.end(), (real)0);
        prev_action_com = V3d(); prev_action_time = 0;
        return true;
    }
    void goal_update(double dt) {                                    // cSceneTargetAMP::UpdateTarget (:253-268) + cSceneHeadingAMP::UpdateTarget (:214-228)
        if (cfg.scene_goal == 5) {                                   // cSceneDribbleAMP::UpdateObjs (:310-319) inside the scene update
            obj_timer += dt;
            const bool oend = obj_timer >= obj_timer_max;
            if (oend) { reset_tar_objs(); obj_timer_reset(); }
        }
        tar_timer += dt;
        const bool end = tar_timer >= tar_timer_max;
        if (end && cfg.scene_goal != 4) goal_reset_target_pos();     // cSceneStrikeAMP::CheckTargetReset is false (:385-388)
        if (cfg.scene_goal == 4 && !target_hit) set_target_hit(check_target_hit());   // cSceneStrikeAMP::UpdateTarget (:290-299)
        if (heading_like() && end) {
            const bool sharp = goal_u01() < cfg.sharp_turn_prob;
            tar_heading += sharp ? goal_uniform(-3.141592653589793, 3.141592653589793) : goal_normal(0.0, cfg.max_heading_turn_rate);
            if (goal_u01() < cfg.speed_change_prob) tar_speed = std::min(std::max(goal_uniform(cfg.tar_speed_min, cfg.tar_speed_max), cfg.tar_speed_min), cfg.tar_speed_max);
        }
        if (end) goal_timer_reset();
    }
    // cSceneStrikeAMP::CheckTargetHit (:441-478): a strike body inside the target sphere moving towards it fast enough
    bool check_target_hit() const {
        V3 tp((real)tar_pos.x, (real)tar_pos.y, (real)tar_pos.z);
        V3 d(tp.x - pose[0], 0, tp.z - pose[2]);
        real n = norm(d); V3 dir; if (n > (real)1e-5) dir = d / n;
        for (int j = 0; j < sk.J; ++j) if ((cfg.strike_mask >> j) & 1u) {
            if (norm2(tp - links[j].com) < (real)(cfg.target_radius * cfg.target_radius)) {
                real speed = dot(dir, links[j].vcom);
                if (speed >= (real)cfg.hit_tar_speed || cfg.hit_tar_speed == 0.0) return true;
            }
        }
        return false;
    }
    bool tar_contact_fail() const {                                  // CheckTarContactFail (:485-503)
        V3 tp((real)tar_pos.x, (real)tar_pos.y, (real)tar_pos.z);
        for (int j = 0; j < sk.J; ++j) if ((cfg.fail_tar_mask >> j) & 1u) if (norm2(tp - links[j].com) < (real)(cfg.target_radius * cfg.target_radius)) return true;
        return false;
    }
    bool tar_hit_succ() const { return target_hit && (timer_time - target_hit_time) >= cfg.target_hit_reset_time; }   // CheckTarHitSucc (:505-520)
    double hit_phase() const {                                       // CalcHitPhase (:390-401)
        if (!target_hit) return 0;
        return std::min(std::max((timer_time - target_hit_time) / cfg.target_hit_reset_time, 0.0), 1.0);
    }
    bool goal_dist_fail() const {                                    // cSceneTargetAMP::CheckTarDistFail (:306-317); cSceneHeadingAMP: false
        if (!target_like()) return false;
        real dx = pose[0] - (real)tar_pos.x, dz = pose[2] - (real)tar_pos.z;
        return dx * dx + dz * dz > (real)cfg.tar_fail_dist * (real)cfg.tar_fail_dist;
    }
    void record_goal(double* out) const {
        Vec rp = reported_pose();
        real heading = calc_heading(root_rot(rp));
        if (cfg.scene_goal == 5) {                                   // cSceneDribbleAMP::RecordGoal (:276-302): ball -> target in the origin frame
            V3 rel((real)tar_pos.x - ball_pos.x, 0, (real)tar_pos.z - ball_pos.z);
            real d = norm(rel);
            V3 r(1, 0, 0);
            if (d > (real)0.0001) r = (origin_trans(rp).R * rel) / d;
            out[0] = r.x; out[1] = r.z; out[2] = d;
        } else if (cfg.scene_goal == 1) {                            // cSceneTargetAMP::RecordGoal (:195-223)
            V3 rel((real)tar_pos.x - rp[0], 0, (real)tar_pos.z - rp[2]);
            real d = norm(rel);
            V3 r(1, 0, 0);
            if (d > (real)0.0001) r = (rot_axis(V3(0, 1, 0), -heading) * rel) / d;
            out[0] = r.x; out[1] = r.z; out[2] = d;
        } else if (cfg.scene_goal == 4) {                            // cSceneStrikeAMP::RecordGoal (:414-434): target in the origin frame, hit phase
            V3 t = xf_point(origin_trans(rp), V3((real)tar_pos.x, (real)tar_pos.y, (real)tar_pos.z));
            out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = hit_phase();
        } else {                                                     // cSceneHeadingAMP::RecordGoal (:150-166)
            real th = (real)tar_heading - heading;
            out[0] = std::cos(th); out[1] = -std::sin(th); out[2] = tar_speed;
            // cSceneHeadingAMPGetup::RecordGoal (:123-130): + CalcGetupPhase (:293-299)
            if (cfg.scene_goal == 3) out[3] = std::min(std::max(1.0 - getup_timer / cfg.getup_time, 0.0), 1.0);
        }
    }
    double calc_goal_reward() const {
        if (cfg.scene_goal == 3 && getting_up()) {                   // cSceneHeadingAMPGetup::CalcReward / CalcRewardGetup (:4-38)
            real root_h = pose[1], head_h = links[cfg.head_id].com.y;
            real nr = std::min(std::max(root_h / (real)cfg.getup_height_root, (real)0), (real)1), nh = std::min(std::max(head_h / (real)cfg.getup_height_head, (real)0), (real)1);
            return (real)0.2 * nr + (real)0.8 * nh;
        }
        if (cfg.scene_goal == 4) return calc_strike_reward();
        if (cfg.scene_goal == 5) return calc_dribble_reward();
        if (has_fallen()) return 0;
        V3 com = sim_com();
        V3 dcom(com.x - (real)prev_action_com.x, com.y - (real)prev_action_com.y, com.z - (real)prev_action_com.z);
        real step_dur = (real)(ctrl_time - prev_action_time);
        real ts = (real)tar_speed;
        if (cfg.scene_goal == 1) {                                   // cSceneTargetAMP::CalcReward (:3-81)
            if (goal_dist_fail()) return 0;
            Vec rp = reported_pose();
            real dx = (real)tar_pos.x - rp[0], dz = (real)tar_pos.z - rp[2], dist_sq = dx * dx + dz * dz;
            real pos_reward = std::exp(-(real)cfg.pos_reward_scale * dist_sq), vel_reward = 0;
            if (dist_sq < (real)cfg.target_succ_dist * (real)cfg.target_succ_dist) vel_reward = 1;
            else {
                V3 ct((real)tar_pos.x - com.x, 0, (real)tar_pos.z - com.z);
                real cd = norm(ct);
                V3 dir; if (cd > (real)0.0001) dir = ct / cd;
                real avg_vel = dot(dir, dcom) / step_dur, vel_err = ts - avg_vel;
                if (!(avg_vel < 0)) { if (cfg.enable_min_tar_vel) vel_err = std::max(vel_err, (real)0); vel_reward = std::exp(-((real)4 / (ts * ts)) * vel_err * vel_err); }
            }
            return (real)0.6 * pos_reward + (real)0.4 * vel_reward;
        }
        V3 av = dcom / step_dur; av.y = 0;                           // cSceneHeadingAMP::CalcReward (:3-43)
        real avg_speed = std::cos((real)tar_heading) * av.x - std::sin((real)tar_heading) * av.z;
        if (!(avg_speed > 0)) return 0;
        real vel_err = ts - avg_speed;
        if (cfg.enable_min_tar_vel) vel_err = std::max(vel_err, (real)0);
        return std::exp(-(real)cfg.vel_reward_scale * vel_err * vel_err);
    }
    // cSceneDribbleAMP::CalcReward (:6-122)
    double calc_dribble_reward() const {
        if (cfg.mode_test) {
            if (is_episode_end() && check_terminate() == TERM_SUCC) return timer_max - timer_time;
            return 0;
        }
        if (has_fallen()) return 0;
        const real ts = (real)tar_speed, te = (real)(ctrl_time - prev_action_time);
        V3 com = sim_com();
        V3 pac((real)prev_action_com.x, (real)prev_action_com.y, (real)prev_action_com.z), pbp((real)prev_ball_pos.x, (real)prev_ball_pos.y, (real)prev_ball_pos.z);
        V3 tp((real)tar_pos.x, (real)tar_pos.y, (real)tar_pos.z);
        V3 com_delta = com - pac, com_ball_delta = ball_pos - pac; com_delta.y = 0; com_ball_delta.y = 0;
        real cbn = norm(com_ball_delta);
        V3 com_ball_dir = com_ball_delta / cbn;                      // Eigen normalized(): no guard in the reference either
        real com_ball_dist = norm2(com_ball_delta);
        real com_ball_vel = dot(com_ball_dir, com_delta) / te;
        real cbv_err = std::min((real)0, com_ball_vel - ts); cbv_err *= cbv_err;
        V3 ball_delta = ball_pos - pbp, ball_target_delta = tp - pbp; ball_delta.y = 0; ball_target_delta.y = 0;
        V3 ball_target_dir = ball_target_delta / norm(ball_target_delta);
        real btv = dot(ball_target_dir, ball_delta) / te;
        real tv_err = std::min((real)0, btv - ts); tv_err *= tv_err;
        real cur_bt_dist = norm2(tp - ball_pos), tp_err = std::sqrt(cur_bt_dist);
        real r0 = std::exp(-(real)1.5 * cbv_err), r1 = std::exp(-(real)0.5 * com_ball_dist), r2 = std::exp(-(real)1 * tv_err), r3 = std::exp(-(real)0.5 * tp_err);
        if (cur_bt_dist < (real)cfg.target_succ_dist * (real)cfg.target_succ_dist && com_ball_dist < (real)4) r0 = r1 = r2 = r3 = 1;
        return (real)0.1 * r0 + (real)0.1 * r1 + (real)0.3 * r2 + (real)0.5 * r3;
    }
    // cSceneStrikeAMP::CalcReward (:9-187)
    double calc_strike_reward() const {
        if (cfg.mode_test) {                                         // CalcRewardTest (:57-72)
            if (is_episode_end() && check_terminate() == TERM_SUCC) return timer_max - timer_time;
            return 0;
        }
        const real far_w = (real)0.3, near_w = (real)0.3, hit_w = (real)0.4;
        V3 tp((real)tar_pos.x, (real)tar_pos.y, (real)tar_pos.z);
        V3 rd(tp.x - pose[0], 0, tp.z - pose[2]);
        const real dist_sq = rd.x * rd.x + rd.z * rd.z, near = (real)cfg.tar_near_dist;
        if (target_hit) return far_w + near_w + hit_w;
        if (dist_sq < near * near) {                                 // CalcRewardTargetNear (:74-112)
            real n = norm(rd); V3 dir; if (n > (real)1e-5) dir = rd / n;
            real best = 0;
            for (int j = 0; j < sk.J; ++j) if ((cfg.strike_mask >> j) & 1u) {
                real dr = std::exp(-(real)cfg.tar_reward_scale * norm2(tp - links[j].com));
                real vr = std::min(std::max(dot(dir, links[j].vcom) / (real)cfg.hit_tar_speed, (real)0), (real)1); vr *= vr;
                best = std::max(best, (real)0.2 * dr + (real)0.8 * vr);
            }
            return far_w + near_w * best;
        }
        // CalcRewardTargetFar (:114-187)
        if (has_fallen()) return 0;
        const real ts = (real)tar_speed, rt = std::sqrt(dist_sq), de = std::max(rt - near, (real)0);
        real pos_reward = std::exp(-(real)cfg.pos_reward_scale * de * de), vel_reward = 0;
        if (rt < near) vel_reward = 1;
        else {
            V3 com = sim_com();
            V3 ct(tp.x - com.x, 0, tp.z - com.z);
            real cd = norm(ct); V3 dir; if (cd > (real)0.0001) dir = ct / cd;
            V3 dcom(com.x - (real)prev_action_com.x, com.y - (real)prev_action_com.y, com.z - (real)prev_action_com.z);
            real avg_vel = dot(dir, dcom) / (real)(ctrl_time - prev_action_time), vel_err = ts - avg_vel;
            if (!(avg_vel < 0)) { if (cfg.enable_min_tar_vel) vel_err = std::max(vel_err, (real)0); vel_reward = std::exp(-((real)4 / (ts * ts)) * vel_err * vel_err); }
        }
        return far_w * ((real)0.7 * pos_reward + (real)0.3 * vel_reward);
    }
    // cSceneImitate::SyncKinCharNewCycle (SceneImitate.cpp:420-444)
    void sync_kin_new_cycle() {
        if (cfg.sync_char_root_rot) {
            real dh = calc_heading(root_rot(pose)) - calc_heading(root_rot(kin.pose));
            kin.rotate_root(quat_axis_angle(V3(0, 1, 0), dh));
        }
        if (cfg.sync_char_root_pos) {
            V3 sp = root_pos(pose), kp = root_pos(kin.pose);
            kp.x = sp.x; kp.z = sp.z;
            real dh = kp.y - kin.origin.y;
            kp.y = 0 + dh;   // ground height of the plane
            kin.set_root_pos_(kp);
        }
    }

    // ------------------------------------------------------------------ termination
    // cSceneSimChar::HasFallenContact (SceneSimChar.cpp:838-841); cSceneHeadingAMPGetup's override (:255-264): never while getting up
    bool has_fallen_contact() const {
        if (getting_up()) return false;
        if (cfg.enable_char_contact_fall) for (int j = 0; j < sk.J; ++j) if (fall_mask[j] && in_contact[j]) return true;
        return false;
    }
    bool has_fallen() const {
        bool f = has_fallen_contact();
        if (cfg.scene_goal == 5) f = f || tar_obj_dist_fail() || char_obj_dist_fail();   // cSceneDribbleAMP::HasFallen (:343-349)
        if (cfg.enable_root_rot_fail) f |= quat_diff_theta(root_rot(pose), root_rot(kin.pose)) > (real)0.5 * kPi;   // SceneImitate.cpp:484-492
        return f;
    }
    int check_terminate() const {
        bool fail = cfg.enable_fall_end && has_fallen();
        // cSceneImitateAMP::CheckTerminate (SceneImitateAMP.cpp:184-188) keeps only the fall test of cRLSceneSimChar (:187-197)
        if (!fail && !cfg.scene_amp && kin.mo->is_over(kin.time)) fail = true;   // SceneImitate.cpp:193-205
        if (!fail && cfg.scene_goal && goal_dist_fail()) fail = true;            // cSceneTargetAMP::CheckTerminate (:319-345)
        if (!fail && cfg.scene_goal == 5 && dribble_succ()) return TERM_SUCC;    // cSceneDribbleAMP::CheckTerminateTarget (:466-475)
        if (!fail && cfg.scene_goal == 4) {                                       // cSceneStrikeAMP::CheckTerminateTarget (:522-541)
            if (tar_contact_fail()) fail = true;
            else if (tar_hit_succ()) return TERM_SUCC;
        }
        return fail ? TERM_FAIL : TERM_NULL;
    }
    bool is_episode_end() const { return timer_time >= timer_max || check_terminate() != TERM_NULL; }
    bool check_valid_episode() const {                               // SimCharacter.cpp:571-586
        for (int j = 0; j < sk.J; ++j) {
            const LinkState& l = links[j];
            real m = std::max(std::max(std::fabs(l.vcom.x), std::fabs(l.vcom.y)), std::fabs(l.vcom.z));
            m = std::max(m, std::max(std::max(std::fabs(l.w.x), std::fabs(l.w.y)), std::fabs(l.w.z)));
            if (m > 100) return false;
        }
        return true;
    }

    // ------------------------------------------------------------------ reward (SceneImitate.cpp:7-127,163-175)
    double calc_reward(double* terms /*5 errors, optional*/ = nullptr) const {
        // cSceneImitateAMP::CalcReward (SceneImitateAMP.cpp:173-182): 0 outside the test-mode time-warp score
        if (cfg.scene_goal) { if (terms) for (int i = 0; i < 5; ++i) terms[i] = 0; return calc_goal_reward(); }
        if (cfg.scene_amp) { if (terms) for (int i = 0; i < 5; ++i) terms[i] = 0; return 0; }
        if (has_fallen()) return 0;
        const int J = sk.J;
        const real pose_w = 0.5, vel_w = 0.05, end_eff_w = 0.15, root_w = 0.2, com_w = 0.1;
        const real total_w = pose_w + vel_w + end_eff_w + root_w + com_w;
        const real pose_scale = (real)2.0 / 15 * J, vel_scale = (real)0.1 / 15 * J, end_eff_scale = 10, root_scale = 5, com_scale = 10;
        Vec pose0 = reported_pose(); const Vec& vel0 = vel; const Vec& pose1 = kin.pose; const Vec& vel1 = kin.vel;
        Xf origin0 = origin_trans(pose0), origin1 = origin_trans(pose1);
        // sim COM velocity: mass-weighted link velocities (SimCharacter.cpp:418-436)
        V3 com_vel0; real tm = 0;
        for (int j = 0; j < J; ++j) if (sk.valid_body(j)) { com_vel0 += (real)sk.mass(j) * links[j].vcom; tm += (real)sk.mass(j); }
        com_vel0 = com_vel0 / tm;
        V3 com1, com_vel1; calc_com(sk, pose1, vel1, com1, com_vel1);
        V3 root_pos0 = root_pos(pose0), root_pos1 = root_pos(pose1);
        Q4 root_rot0 = root_rot(pose0), root_rot1 = root_rot(pose1);
        real pose_err = 0, vel_err = 0, end_eff_err = 0;
        real th = quat_diff_theta(root_rot0, root_rot1);
        pose_err += joint_w[0] * th * th;
        vel_err += joint_w[0] * norm2(root_ang_vel(vel1) - root_ang_vel(vel0));
        for (int j = 1; j < J; ++j) {
            real w = joint_w[j]; int off = sk.offset(j), sz = sk.size(j);
            real pe = 0, ve = 0;
            if (sk.type(j) == JT_SPHERICAL) { real t = quat_theta(quat_diff(joint_quat(pose0, off), joint_quat(pose1, off))); pe = t * t; }
            else for (int k = 0; k < sz; ++k) { real d = pose1[off + k] - pose0[off + k]; pe += d * d; }
            for (int k = 0; k < sz; ++k) { real d = vel1[off + k] - vel0[off + k]; ve += d * d; }
            pose_err += w * pe; vel_err += w * ve;
            if (sk.is_end_eff(j)) {
                V3 p0 = links[j].joint.t;                       // cSimCharacter::CalcJointPos (SimCharacter.cpp:308-331)
                V3 p1 = joint_world_pos(sk, pose1, j);
                real gh0 = 0, gh1 = kin.origin.y;
                V3 rel0 = p0 - root_pos0, rel1 = p1 - root_pos1;
                rel0.y = p0.y - gh0; rel1.y = p1.y - gh1;
                rel0 = origin0.R * rel0; rel1 = origin1.R * rel1;  // w = 0: rotation only
                end_eff_err += norm2(rel1 - rel0);
            }
        }
        V3 rp0 = root_pos0, rp1 = root_pos1;
        rp0.y -= 0; rp1.y -= kin.origin.y;
        real root_pos_err = norm2(rp0 - rp1);
        real root_rot_err = th * th;
        real root_vel_err = norm2(root_vel(vel1) - root_vel(vel0));
        real root_ang_vel_err = norm2(root_ang_vel(vel1) - root_ang_vel(vel0));
        real root_err = root_pos_err + (real)0.1 * root_rot_err + (real)0.01 * root_vel_err + (real)0.001 * root_ang_vel_err;
        real com_err = (real)0.1 * norm2(com_vel1 - com_vel0);
        if (terms) { terms[0] = pose_err; terms[1] = vel_err; terms[2] = end_eff_err; terms[3] = root_err; terms[4] = com_err; }
        real r = pose_w / total_w * std::exp(-pose_scale * pose_err) + vel_w / total_w * std::exp(-vel_scale * vel_err)
               + end_eff_w / total_w * std::exp(-end_eff_scale * end_eff_err) + root_w / total_w * std::exp(-root_scale * root_err)
               + com_w / total_w * std::exp(-com_scale * com_err);
        return r;
    }

    // ------------------------------------------------------------------ observation (CtController.cpp:281-478)
    double ctrl_phase() const {                                      // CtController.cpp:152-159
        double ph = std::fmod(ctrl_time / mo.duration(), 1.0);
        return (ph < 0) ? (1 + ph) : ph;
    }
    void record_state(double* out) const {
        const int J = sk.J;
        int idx = 0;
        if (cfg.enable_phase_input) out[idx++] = ctrl_phase();
        Vec rp = reported_pose();
        Xf ot = origin_trans(rp);
        Q4 oq = quat_from_rot(ot.R);
        V3 rpos = root_pos(rp);
        real ground_h = 0;
        V3 rrel = rpos; rrel.y -= ground_h; rrel = xf_point(ot, rrel);
        out[idx++] = rrel.y;
        for (int i = 0; i < J; ++i) {
            V3 p = links[i].com; p.y -= ground_h;
            if (!cfg.record_world_root_pos || i != 0) { p = xf_point(ot, p); p = p - rrel; }
            out[idx] = p.x; out[idx + 1] = p.y; out[idx + 2] = p.z;
            Q4 q = quat_from_rot(links[i].Rb);
            if (!cfg.record_world_root_rot || i != 0) q = oq * q;
            V3 nrm = qrot(q, V3(0, 1, 0)), tan = qrot(q, V3(1, 0, 0));
            out[idx + 3] = nrm.x; out[idx + 4] = nrm.y; out[idx + 5] = nrm.z;
            out[idx + 6] = tan.x; out[idx + 7] = tan.y; out[idx + 8] = tan.z;
            idx += 9;
        }
        for (int i = 0; i < J; ++i) {
            V3 v = links[i].vcom, w = links[i].w;
            if (!cfg.record_world_root_rot || i != 0) { v = ot.R * v; w = ot.R * w; }
            out[idx] = v.x; out[idx + 1] = v.y; out[idx + 2] = v.z; out[idx + 3] = w.x; out[idx + 4] = w.y; out[idx + 5] = w.z;
            idx += 6;
        }
        if (cfg.scene_goal == 5) {                                   // cSceneDribbleAMP::RecordTaskState (:554-590)
            V3 p = ball_pos; p.y -= ground_h; p = xf_point(ot, p);
            Q4 q = oq * ball_rot;
            V3 nrm = qrot(q, V3(0, 1, 0)), tan = qrot(q, V3(1, 0, 0)), v = qrot(oq, ball_vel), w = qrot(oq, ball_w);
            const real t15[15] = { p.x, p.y, p.z, nrm.x, nrm.y, nrm.z, tan.x, tan.y, tan.z, v.x, v.y, v.z, w.x, w.y, w.z };
            for (int k = 0; k < 15; ++k) out[idx++] = t15[k];
        }
        assert(idx == S);
    }

    // ------------------------------------------------------------------ AMP observations (SceneImitateAMP.cpp:76-138,213-371)
    int amp_pose_size() const {                                      // GetAMPObsPoseSize (:213-241)
        int size = 1 + 6, n_ee = 0;
        for (int j = 0; j < sk.J; ++j) if (sk.is_end_eff(j)) ++n_ee;
        size += 3 * n_ee;
        for (int j = 1; j < sk.J; ++j) size += (sk.type(j) == JT_SPHERICAL) ? 6 : sk.size(j);
        return size;
    }
    int amp_vel_size() const { return sk.P - 7 + 6; }                // GetAMPObsVelSize (:243-257)
    int amp_obs_size() const { return 2 * (amp_pose_size() + amp_vel_size()); }
    int record_amp_pose(const Vec& p, real ground_h, const Q4& ref_rot, double* out) const {   // RecordAMPObsPose (:279-338)
        int idx = 0;
        V3 rpos = root_pos(p); Q4 rrot = root_rot(p);
        out[idx++] = rpos.y - ground_h;
        if (cfg.enable_amp_obs_local_root) rrot = ref_rot * rrot;
        V3 nrm = qrot(rrot, V3(0, 1, 0)), tan = qrot(rrot, V3(1, 0, 0));   // cMathUtil::CalcNormalTangent (MathUtil.cpp:617-623)
        out[idx] = nrm.x; out[idx + 1] = nrm.y; out[idx + 2] = nrm.z; out[idx + 3] = tan.x; out[idx + 4] = tan.y; out[idx + 5] = tan.z;
        idx += 6;
        for (int j = 1; j < sk.J; ++j) {
            int off = sk.offset(j);
            if (sk.type(j) == JT_SPHERICAL) {
                Q4 q = joint_quat(p, off);
                V3 n = qrot(q, V3(0, 1, 0)), t = qrot(q, V3(1, 0, 0));
                out[idx] = n.x; out[idx + 1] = n.y; out[idx + 2] = n.z; out[idx + 3] = t.x; out[idx + 4] = t.y; out[idx + 5] = t.z;
                idx += 6;
            } else for (int k = 0; k < sk.size(j); ++k) out[idx++] = p[off + k];
        }
        for (int j = 0; j < sk.J; ++j) if (sk.is_end_eff(j)) {
            // cKinTree::CalcBodyPartPos (KinTree.cpp:272-281): joint world transform x body attach point
            V3 bp = xf_point(joint_world_trans(sk, p, j), sk.body_attach_pt(j));
            V3 rel = qrot(ref_rot, bp - rpos);
            out[idx] = rel.x; out[idx + 1] = rel.y; out[idx + 2] = rel.z; idx += 3;
        }
        return idx;
    }
    int record_amp_vel(const Vec& v, const Q4& ref_rot, double* out) const {   // RecordAMPObsVel (:340-371)
        V3 rv = root_vel(v), rw = root_ang_vel(v);
        if (cfg.enable_amp_obs_local_root) { rv = qrot(ref_rot, rv); rw = qrot(ref_rot, rw); }
        out[0] = rv.x; out[1] = rv.y; out[2] = rv.z; out[3] = rw.x; out[4] = rw.y; out[5] = rw.z;
        for (int i = 7; i < sk.P; ++i) out[6 + i - 7] = v[i];
        return 6 + sk.P - 7;
    }
    void build_amp_obs(const Vec& pp, const Vec& pv, const Vec& p, const Vec& v, real ground_h, double* out) const {   // BuildAMPObs (:259-277)
        real heading = calc_heading(root_rot(p));
        Q4 ref_rot = quat_axis_angle(V3(0, 1, 0), -heading);          // cKinTree::CalcHeadingRot (KinTree.cpp:1629-1635)
        int idx = 0;
        idx += record_amp_pose(p, ground_h, ref_rot, out + idx);
        idx += record_amp_pose(pp, ground_h, ref_rot, out + idx);
        idx += record_amp_vel(v, ref_rot, out + idx);
        idx += record_amp_vel(pv, ref_rot, out + idx);
        assert(idx == amp_obs_size());
    }
    void amp_obs_agent(double* out) const {                          // RecordAMPObsAgent (:101-113); plane: ground height 0
        build_amp_obs(prev_pose, prev_vel, reported_pose(), vel, 0, out);
    }
    // RecordAMPObsExpert (:115-138) with the random clip time passed in: raw cMotion::CalcFrame / CalcFrameVel (no origin,
    // no cycle offset) at t and t - 1/query_rate, ground height := kin origin y
    void amp_obs_expert(double t, double* out, int clip = -1, double ground_h = std::numeric_limits<double>::quiet_NaN()) const {
        const Motion& mm = (clip >= 0 && !clips.empty()) ? clips[clip] : mo;      // SampleExpertMotion with a cClipsController
        Vec p, v, pp, pv;
        mm.calc_frame(sk, t, p); mm.calc_frame_vel(t, v);
        double tp = t - 1.0 / cfg.query_rate;
        mm.calc_frame(sk, tp, pp); mm.calc_frame_vel(tp, pv);
        build_amp_obs(pp, pv, p, v, std::isnan(ground_h) ? (double)kin.origin.y : ground_h, out);
    }
};

}  // namespace orc
