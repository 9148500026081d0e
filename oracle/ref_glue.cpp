// TEST INFRASTRUCTURE ONLY -- C entry points over the reference's OWN, UNMODIFIED sources.
//
// oracle/_ref/libdm_ref.so = this file + /root/reference/DeepMimicCore/{util/MathUtil, util/Rand, util/JsonUtil,
// util/FileUtil, util/Timer, util/Annealer, util/DynamicTimeWarper, util/json/*, sim/SpAlg, sim/RBDUtil, sim/RBDModel, sim/CtCtrlUtil,
// anim/KinTree, anim/Shape, anim/Motion, anim/Character, anim/KinCharacter, anim/KinController, anim/MotionController,
// anim/ClipsController}.cpp compiled where they lie (recipe: oracle/Makefile, target `ref`) against oracle/eigen_shim
// (Eigen is not installed) and oracle/gl_stub (type names only).  Nothing is copied into this repository.
//
// What the functions below do: marshal plain double arrays into the reference's Eigen types, call the reference
// function named in the comment, marshal the result back.  Six functions (ref_spd_tau, ref_reward_terms, ref_record_state,
// ref_amp_obs, ref_time_warp, ref_action_to_target) additionally COMPOSE reference functions in the order of a reference routine that cannot
// itself be compiled here because its translation unit includes Bullet headers; each cites the lines it follows.
//
// Only tests/ (and tests/golden/make_ref_golden.py) load this library.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

#include "anim/ClipsController.h"
#include "anim/KinCharacter.h"
#include "anim/KinTree.h"
#include "anim/Motion.h"
#include "anim/MotionController.h"
#include "render/DrawMesh.h"
#include "render/DrawUtil.h"
#include "render/MeshUtil.h"
#include "sim/CtCtrlUtil.h"
#include "sim/RBDModel.h"
#include "sim/RBDUtil.h"
#include "sim/SpAlg.h"
#include "util/DynamicTimeWarper.h"
#include "util/MathUtil.h"
#include "util/Timer.h"
#include "util/Annealer.h"

// ---- link-time stand-ins for the renderer (anim/Character.cpp references them; draw is disabled) -----------------
bool cDrawUtil::EnableDraw() { return false; }
cDrawMesh::cDrawMesh() {}
cDrawMesh::~cDrawMesh() {}
bool cMeshUtil::LoadObj(const std::string&, cDrawMesh&) { return false; }

namespace {

typedef Eigen::VectorXd VecX;
typedef Eigen::MatrixXd MatX;

VecX vin(const double* p, int n) { VecX v(n); for (int i = 0; i < n; ++i) v[i] = p[i]; return v; }
void vout(const VecX& v, double* p) { for (int i = 0; i < (int)v.size(); ++i) p[i] = v[i]; }
tQuaternion qin(const double* p) { return tQuaternion(p[0], p[1], p[2], p[3]); }   // (w, x, y, z)
void qout(const tQuaternion& q, double* p) { p[0] = q.w(); p[1] = q.x(); p[2] = q.y(); p[3] = q.z(); }
tVector v3in(const double* p) { return tVector(p[0], p[1], p[2], 0); }
void v3out(const tVector& v, double* p) { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; }
void m4out(const tMatrix& m, double* p /*rot 9 row-major + pos 3*/) {
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) p[a * 3 + b] = m(a, b);
    p[9] = m(0, 3); p[10] = m(1, 3); p[11] = m(2, 3);
}

struct Skel {
    MatX jm, bd;
    int J, P;
    cRBDModel rbd;
    tVector gravity;
};

struct KinChar {
    std::shared_ptr<cKinCharacter> kc;
};

}  // namespace

extern "C" {

// ---- cMathUtil / cKinTree scalar+quaternion functions.  op codes shared with orc_math_op (oracle/dm_oracle.cpp) ----
int ref_math_op(int op, const double* in, double* out) {
    switch (op) {
        case 0: qout(cMathUtil::ExpMapToQuaternion(v3in(in)), out); return 4;                       // MathUtil.cpp:573-599
        case 1: v3out(cMathUtil::QuaternionToExpMap(qin(in)), out); return 3;                        // :607-615
        case 2: out[0] = cMathUtil::QuatDiffTheta(qin(in), qin(in + 4)); return 1;                   // :527-549
        case 3: v3out(cMathUtil::CalcQuaternionVel(qin(in), qin(in + 4), in[8]), out); return 3;    // :493-500
        case 4: v3out(cMathUtil::CalcQuaternionVelRel(qin(in), qin(in + 4), in[8]), out); return 3; // :502-510
        case 5: { tVector n, t; cMathUtil::CalcNormalTangent(qin(in), n, t); v3out(n, out); v3out(t, out + 3); return 6; }  // :617-623
        case 6: out[0] = cMathUtil::NormalizeAngle(in[0]); return 1;
        case 7: qout(qin(in).slerp(in[8], qin(in + 4)), out); return 4;                               // Eigen (shim) slerp as KinTree.cpp:1529-1572 calls it
        case 8: out[0] = cMathUtil::CheckNextInterval(in[0], in[1], in[2]) ? 1 : 0; return 1;         // :850-857
        case 9: out[0] = cKinTree::CalcHeading(qin(in)); return 1;                                    // KinTree.cpp:1619-1627
        case 10: { tVector ax; double th; cMathUtil::QuaternionToAxisAngle(qin(in), ax, th); v3out(ax, out); out[3] = th; return 4; }
        case 11: qout(cMathUtil::AxisAngleToQuaternion(v3in(in), in[3]), out); return 4;
        case 12: v3out(cMathUtil::QuatRotVec(qin(in), v3in(in + 4)), out); return 3;
        case 13: { tMatrix m = cMathUtil::RotateMat(qin(in)); for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out[a * 3 + b] = m(a, b); return 9; }
        case 14: { tMatrix m = cMathUtil::RotateMat(v3in(in)); for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out[a * 3 + b] = m(a, b); return 9; }  // euler
        case 15: qout(cMathUtil::EulerToQuaternion(v3in(in)), out); return 4;
        case 16: qout(cMathUtil::StandardizeQuat(qin(in)), out); return 4;
        case 17: qout(cMathUtil::QuatDiff(qin(in), qin(in + 4)), out); return 4;
        case 18: { tMatrix m = cMathUtil::RotateMat(v3in(in), in[3]); for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out[a * 3 + b] = m(a, b); return 9; }  // axis-angle
        case 19: {  // RotMatToQuaternion of RotateMat(q): round trip through the reference's own matrix path
            qout(cMathUtil::RotMatToQuaternion(cMathUtil::RotateMat(qin(in))), out); return 4;
        }
        case 20: qout(cKinTree::CalcHeadingRot(qin(in)), out); return 4;
        case 21: out[0] = cMotion::CalcPhase(in[0], in[1], in[2], in[3] != 0); return 1;
    }
    return -1;
}

// ---- skeleton-level functions -------------------------------------------------------------------------------------
void* ref_skel_create(const double* jm, const double* bd, int J, const double* gravity) {
    Skel* s = new Skel();
    s->J = J;
    s->jm.resize(J, cKinTree::eJointDescMax);
    s->bd.resize(J, cKinTree::eBodyParamMax);
    for (int j = 0; j < J; ++j) {
        for (int c = 0; c < cKinTree::eJointDescMax; ++c) s->jm(j, c) = jm[j * cKinTree::eJointDescMax + c];
        for (int c = 0; c < cKinTree::eBodyParamMax; ++c) s->bd(j, c) = bd[j * cKinTree::eBodyParamMax + c];
    }
    s->P = cKinTree::GetNumDof(s->jm);
    s->gravity = tVector(gravity[0], gravity[1], gravity[2], 0);
    s->rbd.Init(s->jm, s->bd, s->gravity);   // RBDModel.cpp:12-34
    return s;
}
void ref_skel_destroy(void* h) { delete (Skel*)h; }
int ref_skel_num_dof(void* h) { return ((Skel*)h)->P; }
int ref_joint_desc_max() { return cKinTree::eJointDescMax; }
int ref_body_param_max() { return cKinTree::eBodyParamMax; }

// cKinTree::Load + LoadBodyDefs on a character file: out_jm J x 19, out_bd J x 17 (caller passes capacity)
int ref_load_char(const char* file, double* out_jm, double* out_bd, int cap_j) {
    std::ifstream f(file);
    if (!f.good()) return -1;
    Json::Reader reader; Json::Value root;
    if (!reader.parse(f, root)) return -2;
    if (root["Skeleton"].isNull()) return -3;
    MatX jm, bd;
    if (!cKinTree::Load(root["Skeleton"], jm)) return -4;          // KinTree.cpp:433-481 (as cCharacter::LoadSkeleton calls it)
    if (!cKinTree::LoadBodyDefs(file, bd)) return -5;               // KinTree.cpp:125-172
    int J = (int)jm.rows();
    if (J > cap_j) return -6;
    for (int j = 0; j < J; ++j) {
        for (int c = 0; c < cKinTree::eJointDescMax; ++c) out_jm[j * cKinTree::eJointDescMax + c] = jm(j, c);
        for (int c = 0; c < cKinTree::eBodyParamMax; ++c) out_bd[j * cKinTree::eBodyParamMax + c] = bd(j, c);
    }
    return J;
}

void ref_skel_lerp_poses(void* h, const double* p0, const double* p1, double t, double* out) {          // KinTree.cpp:1529-1572
    Skel* s = (Skel*)h; VecX o; cKinTree::LerpPoses(s->jm, vin(p0, s->P), vin(p1, s->P), t, o); vout(o, out);
}
void ref_skel_calc_vel(void* h, const double* p0, const double* p1, double dt, double* out) {           // KinTree.cpp:1470-1508
    Skel* s = (Skel*)h; VecX o; cKinTree::CalcVel(s->jm, vin(p0, s->P), vin(p1, s->P), dt, o); vout(o, out);
}
void ref_skel_vel_to_pose_diff(void* h, const double* pose, const double* vel, double* out) {           // KinTree.cpp:1579-1610
    Skel* s = (Skel*)h; VecX o; cKinTree::VelToPoseDiff(s->jm, vin(pose, s->P), vin(vel, s->P), o); vout(o, out);
}
void ref_skel_post_process_pose(void* h, double* pose) {                                               // KinTree.cpp:1510-1527
    Skel* s = (Skel*)h; VecX p = vin(pose, s->P); cKinTree::PostProcessPose(s->jm, p); vout(p, pose);
}
// out: [root_rot_err, root_ang_vel_err, pose_err[J], vel_err[J]]  (KinTree.cpp:1319-1426)
void ref_skel_pose_errs(void* h, const double* p0, const double* p1, const double* v0, const double* v1, double* out) {
    Skel* s = (Skel*)h; VecX a = vin(p0, s->P), b = vin(p1, s->P), c = vin(v0, s->P), d = vin(v1, s->P);
    out[0] = cKinTree::CalcRootRotErr(s->jm, a, b);
    out[1] = cKinTree::CalcRootAngVelErr(s->jm, c, d);
    out[2] = 0; out[2 + s->J] = 0;
    for (int j = 1; j < s->J; ++j) {
        out[2 + j] = cKinTree::CalcPoseErr(s->jm, j, a, b);
        out[2 + s->J + j] = cKinTree::CalcVelErr(s->jm, j, c, d);
    }
}
// per joint: JointWorldTrans (rot 9 + pos 3), then per link BodyWorldTrans (rot 9 + pos 3)   (KinTree.cpp:1078-1091, body: :986-1003)
void ref_skel_world_trans(void* h, const double* pose, double* out_joint, double* out_body) {
    Skel* s = (Skel*)h; VecX p = vin(pose, s->P);
    for (int j = 0; j < s->J; ++j) {
        m4out(cKinTree::JointWorldTrans(s->jm, p, j), out_joint + 12 * j);
        if (cKinTree::IsValidBody(s->bd, j)) m4out(cKinTree::BodyWorldTrans(s->jm, s->bd, p, j), out_body + 12 * j);
        else for (int k = 0; k < 12; ++k) out_body[12 * j + k] = 0;
    }
}
// link COM linear velocity and joint-frame angular velocity (cKinTree::CalcBodyPartVel, CalcJointWorldAngularVel)
void ref_skel_link_vel(void* h, const double* pose, const double* vel, double* out /*J x 6: v(3) w(3)*/) {
    Skel* s = (Skel*)h; VecX p = vin(pose, s->P), v = vin(vel, s->P);
    for (int j = 0; j < s->J; ++j) {
        tVector lv = tVector::Zero();
        if (cKinTree::IsValidBody(s->bd, j)) lv = cKinTree::CalcBodyPartVel(s->jm, s->bd, p, v, j);
        tVector w = cKinTree::CalcJointWorldAngularVel(s->jm, p, v, j);
        v3out(lv, out + 6 * j); v3out(w, out + 6 * j + 3);
    }
}
// cRBDModel::Update -> mass matrix H [P x P, row-major] and bias force C [P]   (RBDModel.cpp:36-46; RBDUtil.cpp:4-195)
void ref_skel_mass_bias(void* h, const double* pose, const double* vel, double* H, double* C) {
    Skel* s = (Skel*)h;
    s->rbd.Update(vin(pose, s->P), vin(vel, s->P));
    const MatX& M = s->rbd.GetMassMat(); const VecX& c = s->rbd.GetBiasForce();
    for (int i = 0; i < s->P; ++i) { for (int j = 0; j < s->P; ++j) H[i * s->P + j] = M(i, j); C[i] = c[i]; }
}
void ref_skel_inv_dyna(void* h, const double* pose, const double* vel, const double* acc, double* tau) {  // RBDUtil.cpp:4-97
    Skel* s = (Skel*)h;
    s->rbd.Update(vin(pose, s->P), vin(vel, s->P));
    VecX t; cRBDUtil::SolveInvDyna(s->rbd, vin(acc, s->P), t); vout(t, tau);
}
void ref_skel_com(void* h, const double* pose, const double* vel, double* com, double* com_vel) {        // RBDUtil.cpp:572-613
    Skel* s = (Skel*)h; tVector c, v;
    cRBDUtil::CalcCoM(s->jm, s->bd, vin(pose, s->P), vin(vel, s->P), c, v);
    v3out(c, com); v3out(v, com_vel);
}
void ref_skel_origin_trans(void* h, const double* pose, double* out12) {                                 // KinTree.cpp:1653-1664
    Skel* s = (Skel*)h; m4out(cKinTree::BuildOriginTrans(vin(pose, s->P)), out12);
}
double ref_skel_total_mass(void* h) { return cKinTree::CalcTotalMass(((Skel*)h)->bd); }
// spatial inertia of link j about its parent joint frame, 6x6 row-major (RBDUtil.cpp:615-749)
void ref_skel_inertia(void* h, int j, double* out36) {
    Skel* s = (Skel*)h;
    cSpAlg::tSpMat I = cRBDUtil::BuildMomentInertia(s->bd, j);
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out36[a * 6 + b] = I(a, b);
}

// cCtCtrlUtil action bounds / offset / scale per joint (CtCtrlUtil.cpp:10-35 and the Build* functions it dispatches to)
int ref_skel_action_meta(void* h, double* lo, double* hi, double* offset, double* scale) {
    Skel* s = (Skel*)h; int n = 0;
    for (int j = 1; j < s->J; ++j) {
        VecX l, u, o, sc;
        cCtCtrlUtil::BuildBoundsPD(s->jm, j, l, u);
        cCtCtrlUtil::BuildOffsetScalePD(s->jm, j, o, sc);
        for (int k = 0; k < (int)l.size(); ++k) { lo[n] = l[k]; hi[n] = u[k]; offset[n] = o[k]; scale[n] = sc[k]; ++n; }
    }
    return n;
}

// Composition following cImpPDController::CalcControlForces (ImpPDController.cpp:136-188) and BuildTargetPose/Vel
// (:190-243): every arithmetic step is a call into the compiled reference (cRBDModel, cKinTree, the shim's LDLT).
// kp/kd: pose-layout gain vectors (zeros on the root, as cImpPDController::Init leaves them for an invalid PD controller).
void ref_spd_tau(void* h, const double* pose_, const double* vel_, const double* tar_pose_, const double* kp_, const double* kd_,
                 double dt, double* out_tau) {
    Skel* s = (Skel*)h; const int P = s->P; double t = dt;
    VecX pose = vin(pose_, P), vel = vin(vel_, P), tar_pose = vin(tar_pose_, P), Kp = vin(kp_, P), Kd = vin(kd_, P);
    VecX tar_vel = VecX::Zero(P);
    s->rbd.Update(pose, vel);
    MatX M = s->rbd.GetMassMat();
    const VecX& C = s->rbd.GetBiasForce();
    for (int i = 0; i < P; ++i) M(i, i) += t * Kd[i];
    VecX pose_inc;
    cKinTree::VelToPoseDiff(s->jm, pose, vel, pose_inc);
    pose_inc = pose + t * pose_inc;
    cKinTree::PostProcessPose(s->jm, pose_inc);
    VecX pose_err;
    cKinTree::CalcVel(s->jm, pose_inc, tar_pose, 1, pose_err);
    VecX vel_err = tar_vel - vel;
    VecX acc = Kp.cwiseProduct(pose_err) + Kd.cwiseProduct(vel_err) - C;
    acc = M.ldlt().solve(acc);
    VecX tau = Kp.cwiseProduct(pose_err) + Kd.cwiseProduct(vel_err - t * acc);
    vout(tau, out_tau);
}

// Composition following cSceneImitate::CalcRewardImitate (scenes/SceneImitate.cpp:7-127).  The sim character's
// CalcCOM/CalcCOMVel/CalcJointPos/BuildOriginTrans read Bullet link transforms; for a pose vector they coincide with the
// kinematic-tree quantities used here.  out: [pose_err, vel_err, end_eff_err, root_err, com_err, reward].
void ref_reward_terms(void* h, const double* pose0_, const double* vel0_, const double* pose1_, const double* vel1_,
                      const double* joint_w, double ground_h0, double kin_origin_y, double* out) {
    Skel* s = (Skel*)h; const int P = s->P;
    VecX pose0 = vin(pose0_, P), vel0 = vin(vel0_, P), pose1 = vin(pose1_, P), vel1 = vin(vel1_, P);
    double pose_w = 0.5, vel_w = 0.05, end_eff_w = 0.15, root_w = 0.2, com_w = 0.1;
    double total_w = pose_w + vel_w + end_eff_w + root_w + com_w;
    pose_w /= total_w; vel_w /= total_w; end_eff_w /= total_w; root_w /= total_w; com_w /= total_w;
    int num_joints = s->J;
    const double pose_scale = 2.0 / 15 * num_joints, vel_scale = 0.1 / 15 * num_joints;
    const double end_eff_scale = 10, root_scale = 5, com_scale = 10, err_scale = 1;
    tMatrix origin_trans = cKinTree::BuildOriginTrans(pose0);
    tMatrix kin_origin_trans = cKinTree::BuildOriginTrans(pose1);
    tVector com0, com_vel0, com1, com_vel1;
    cRBDUtil::CalcCoM(s->jm, s->bd, pose0, vel0, com0, com_vel0);
    cRBDUtil::CalcCoM(s->jm, s->bd, pose1, vel1, com1, com_vel1);
    tVector root_pos0 = cKinTree::GetRootPos(pose0), root_pos1 = cKinTree::GetRootPos(pose1);
    tQuaternion root_rot0 = cKinTree::GetRootRot(pose0), root_rot1 = cKinTree::GetRootRot(pose1);
    tVector root_vel0 = cKinTree::GetRootVel(vel0), root_vel1 = cKinTree::GetRootVel(vel1);
    tVector root_ang_vel0 = cKinTree::GetRootAngVel(vel0), root_ang_vel1 = cKinTree::GetRootAngVel(vel1);
    double pose_err = 0, vel_err = 0, end_eff_err = 0, root_err = 0, com_err = 0;
    pose_err += joint_w[0] * cKinTree::CalcRootRotErr(s->jm, pose0, pose1);
    vel_err += joint_w[0] * cKinTree::CalcRootAngVelErr(s->jm, vel0, vel1);
    for (int j = 1; j < num_joints; ++j) {
        pose_err += joint_w[j] * cKinTree::CalcPoseErr(s->jm, j, pose0, pose1);
        vel_err += joint_w[j] * cKinTree::CalcVelErr(s->jm, j, vel0, vel1);
        if (cKinTree::IsEndEffector(s->jm, j)) {
            tVector pos0 = cKinTree::CalcJointWorldPos(s->jm, pose0, j);
            tVector pos1 = cKinTree::CalcJointWorldPos(s->jm, pose1, j);
            tVector pos_rel0 = pos0 - root_pos0, pos_rel1 = pos1 - root_pos1;
            pos_rel0[1] = pos0[1] - ground_h0;
            pos_rel1[1] = pos1[1] - kin_origin_y;
            pos_rel0 = origin_trans * pos_rel0;
            pos_rel1 = kin_origin_trans * pos_rel1;
            end_eff_err += (pos_rel1 - pos_rel0).squaredNorm();
        }
    }
    root_pos0[1] -= ground_h0;
    root_pos1[1] -= kin_origin_y;
    double root_pos_err = (root_pos0 - root_pos1).squaredNorm();
    double root_rot_err = cMathUtil::QuatDiffTheta(root_rot0, root_rot1);
    root_rot_err *= root_rot_err;
    double root_vel_err = (root_vel1 - root_vel0).squaredNorm();
    double root_ang_vel_err = (root_ang_vel1 - root_ang_vel0).squaredNorm();
    root_err = root_pos_err + 0.1 * root_rot_err + 0.01 * root_vel_err + 0.001 * root_ang_vel_err;
    com_err = 0.1 * (com_vel1 - com_vel0).squaredNorm();
    out[0] = pose_err; out[1] = vel_err; out[2] = end_eff_err; out[3] = root_err; out[4] = com_err;
    out[5] = pose_w * exp(-err_scale * pose_scale * pose_err) + vel_w * exp(-err_scale * vel_scale * vel_err) +
             end_eff_w * exp(-err_scale * end_eff_scale * end_eff_err) + root_w * exp(-err_scale * root_scale * root_err) +
             com_w * exp(-err_scale * com_scale * com_err);
}

// Composition following cCtController::RecordState / BuildStatePose / BuildStateVel / BuildStatePhase
// (sim/CtController.cpp:281-293,373-478).  The sim character's body-part world position / rotation / velocities are read
// from Bullet there; for a generalized state they coincide with cKinTree::BodyWorldTrans / CalcBodyPartVel /
// CalcJointWorldAngularVel used here.  flags: bit0 phase input, bit1 record_world_root_pos, bit2 record_world_root_rot.
int ref_record_state(void* h, const double* pose_, const double* vel_, double phase, double ground_h, int flags, double* out) {
    Skel* s = (Skel*)h; const int P = s->P;
    VecX pose = vin(pose_, P), vel = vin(vel_, P);
    const bool phase_in = flags & 1, world_pos = flags & 2, world_rot = flags & 4;
    int idx = 0;
    if (phase_in) out[idx++] = phase;
    tMatrix origin_trans = cKinTree::BuildOriginTrans(pose);
    tQuaternion origin_quat = cMathUtil::RotMatToQuaternion(origin_trans);
    tVector root_pos = cKinTree::GetRootPos(pose);
    tVector root_pos_rel = root_pos;
    root_pos_rel[1] -= ground_h;
    root_pos_rel[3] = 1;
    root_pos_rel = origin_trans * root_pos_rel;
    root_pos_rel[3] = 0;
    out[idx++] = root_pos_rel[1];
    const int root_id = 0;
    for (int i = 0; i < s->J; ++i) {
        tMatrix bw = cKinTree::BodyWorldTrans(s->jm, s->bd, pose, i);
        tVector curr_pos = tVector(bw(0, 3), bw(1, 3), bw(2, 3), 0);
        curr_pos[1] -= ground_h;
        if (!world_pos || i != root_id) {
            curr_pos[3] = 1;
            curr_pos = origin_trans * curr_pos;
            curr_pos -= root_pos_rel;
            curr_pos[3] = 0;
        }
        out[idx] = curr_pos[0]; out[idx + 1] = curr_pos[1]; out[idx + 2] = curr_pos[2];
        tQuaternion curr_quat = cMathUtil::RotMatToQuaternion(bw);
        if (!world_rot || i != root_id) curr_quat = origin_quat * curr_quat;
        tVector n, t;
        cMathUtil::CalcNormalTangent(curr_quat, n, t);
        out[idx + 3] = n[0]; out[idx + 4] = n[1]; out[idx + 5] = n[2];
        out[idx + 6] = t[0]; out[idx + 7] = t[1]; out[idx + 8] = t[2];
        idx += 9;
    }
    for (int i = 0; i < s->J; ++i) {
        tVector lv = cKinTree::CalcBodyPartVel(s->jm, s->bd, pose, vel, i);
        tVector av = cKinTree::CalcJointWorldAngularVel(s->jm, pose, vel, i);
        if (!world_rot || i != root_id) { lv = origin_trans * lv; av = origin_trans * av; }
        out[idx] = lv[0]; out[idx + 1] = lv[1]; out[idx + 2] = lv[2];
        out[idx + 3] = av[0]; out[idx + 4] = av[1]; out[idx + 5] = av[2];
        idx += 6;
    }
    return idx;
}

// Composition following cSceneImitateAMP::BuildAMPObs / RecordAMPObsPose / RecordAMPObsVel (scenes/SceneImitateAMP.cpp:279-396;
// the translation unit includes Bullet headers): [pose block of `pose`, pose block of `prev_pose`, vel block of `vel`, vel block of
// `prev_vel`], every geometric step a call into the compiled reference (cKinTree::CalcHeadingRot, CalcBodyPartPos,
// cMathUtil::CalcNormalTangent, VecToQuat, QuatRotVec).  local_root = --enable_amp_obs_local_root.  Returns the size written.
static int ref_amp_pose_(const Skel* s, const VecX& pose, double ground_h, const tQuaternion& ref_origin_rot, bool local_root, double* out) {
    const int pos_dim = cKinTree::gPosDim;
    int o = 0;
    const tVector root_pos = cKinTree::GetRootPos(pose);
    tQuaternion root_rot = cKinTree::GetRootRot(pose);
    out[o++] = root_pos[1] - ground_h;
    if (local_root) root_rot = ref_origin_rot * root_rot;
    tVector n, t;
    cMathUtil::CalcNormalTangent(root_rot, n, t);
    for (int k = 0; k < pos_dim; ++k) { out[o + k] = n[k]; out[o + pos_dim + k] = t[k]; }
    o += 2 * pos_dim;
    for (int j = 1; j < s->J; ++j) {
        cKinTree::eJointType jt = cKinTree::GetJointType(s->jm, j);
        int off = cKinTree::GetParamOffset(s->jm, j), sz = cKinTree::GetParamSize(s->jm, j);
        if (jt == cKinTree::eJointTypeSpherical) {
            tQuaternion q = cMathUtil::VecToQuat(tVector(pose[off], pose[off + 1], pose[off + 2], pose[off + 3]));
            cMathUtil::CalcNormalTangent(q, n, t);
            for (int k = 0; k < pos_dim; ++k) { out[o + k] = n[k]; out[o + pos_dim + k] = t[k]; }
            o += 2 * pos_dim;
        } else { for (int k = 0; k < sz; ++k) out[o + k] = pose[off + k]; o += sz; }
    }
    for (int j = 0; j < s->J; ++j) {
        if (!cKinTree::IsEndEffector(s->jm, j)) continue;          // cCharacter::mEndEffectors: joints flagged IsEndEffector, in joint order
        tVector p = cKinTree::CalcBodyPartPos(s->jm, s->bd, pose, j);
        p -= root_pos;
        p = cMathUtil::QuatRotVec(ref_origin_rot, p);
        for (int k = 0; k < pos_dim; ++k) out[o + k] = p[k];
        o += pos_dim;
    }
    return o;
}
static int ref_amp_vel_(const Skel* s, const VecX& vel, const tQuaternion& ref_origin_rot, bool local_root, double* out) {
    tVector rv = cKinTree::GetRootVel(vel), rw = cKinTree::GetRootAngVel(vel);
    if (local_root) { rv = cMathUtil::QuatRotVec(ref_origin_rot, rv); rw = cMathUtil::QuatRotVec(ref_origin_rot, rw); }
    int o = 0;
    for (int k = 0; k < 3; ++k) out[o++] = rv[k];
    for (int k = 0; k < 3; ++k) out[o++] = rw[k];
    const int root_size = cKinTree::GetParamSize(s->jm, 0);
    for (int i = root_size; i < s->P; ++i) out[o++] = vel[i];
    return o;
}
int ref_amp_obs(void* h, const double* prev_pose, const double* prev_vel, const double* pose_, const double* vel_, double ground_h, int local_root, double* out) {
    Skel* s = (Skel*)h;
    VecX pp = vin(prev_pose, s->P), pv = vin(prev_vel, s->P), p = vin(pose_, s->P), v = vin(vel_, s->P);
    tQuaternion ref_origin_rot = cKinTree::CalcHeadingRot(p);
    int o = 0;
    o += ref_amp_pose_(s, p, ground_h, ref_origin_rot, local_root != 0, out + o);
    o += ref_amp_pose_(s, pp, ground_h, ref_origin_rot, local_root != 0, out + o);
    o += ref_amp_vel_(s, v, ref_origin_rot, local_root != 0, out + o);
    o += ref_amp_vel_(s, pv, ref_origin_rot, local_root != 0, out + o);
    return o;
}

// Composition following cCtPDController::ConvertActionToTargetPose (sim/CtPDController.cpp:133-166) for one spherical
// joint: exp-map action -> clamp ||a|| <= 2 pi ... the steps in the reference are cMathUtil::ExpMapToAxisAngle,
// AxisAngleToQuaternion; the oracle-side counterpart is orc_math_op(0) plus the controller's set_action.
void ref_action_to_target(const double* a3, double* q4) {
    tVector exp_map(a3[0], a3[1], a3[2], 0);
    tVector axis; double theta;
    cMathUtil::ExpMapToAxisAngle(exp_map, axis, theta);
    qout(cMathUtil::AxisAngleToQuaternion(axis, theta), q4);
}

// ---- kinematic character (anim/KinCharacter.cpp + MotionController.cpp + Motion.cpp, loaded from the data files) ---
void* ref_kinchar_create(const char* char_file, const char* motion_file) {
    KinChar* k = new KinChar();
    k->kc = std::shared_ptr<cKinCharacter>(new cKinCharacter());
    cKinCharacter::tParams p;
    p.mCharFile = char_file; p.mLoadDrawShapes = false;
    if (!k->kc->Init(p)) { delete k; return nullptr; }
    std::shared_ptr<cMotionController> ctrl(new cMotionController());     // as cSceneKinChar/SceneImitate::BuildKinCharController
    ctrl->Init(k->kc.get(), motion_file);
    if (!ctrl->GetMotion().IsValid()) { delete k; return nullptr; }
    k->kc->SetController(ctrl);
    return k;
}
void ref_kinchar_destroy(void* h) { delete (KinChar*)h; }
int ref_kinchar_num_frames(void* h) { return ((KinChar*)h)->kc->GetNumMotionFrames(); }
int ref_kinchar_num_dof(void* h) { return ((KinChar*)h)->kc->GetNumDof(); }
double ref_kinchar_duration(void* h) { return ((KinChar*)h)->kc->GetMotionDuration(); }
int ref_kinchar_loop(void* h) { return ((KinChar*)h)->kc->EnableMotionLoop() ? 1 : 0; }
void ref_kinchar_frame(void* h, int f, double* frame, double* frame_vel, double* time) {
    KinChar* k = (KinChar*)h;
    vout(k->kc->GetMotionFrame(f), frame); vout(k->kc->GetMotionFrameVel(f), frame_vel);
    *time = k->kc->GetMotion()->GetFrameTime(f);
}
void ref_kinchar_set_origin(void* h, const double* pos3, const double* rot4) {
    KinChar* k = (KinChar*)h; k->kc->SetOriginPos(v3in(pos3)); k->kc->SetOriginRot(qin(rot4));
}
void ref_kinchar_get_origin(void* h, double* pos3, double* rot4) {
    KinChar* k = (KinChar*)h; v3out(k->kc->GetOriginPos(), pos3); qout(k->kc->GetOriginRot(), rot4);
}
// cKinCharacter::CalcPose / CalcVel at an arbitrary time (KinCharacter.cpp:363-406 -> MotionController.cpp:25-47,102-111)
void ref_kinchar_eval(void* h, double t, double* pose, double* vel) {
    KinChar* k = (KinChar*)h; VecX p, v; k->kc->CalcPose(t, p); k->kc->CalcVel(t, v); vout(p, pose); vout(v, vel);
}
// SetTime + Pose(), Update(dt): the stateful path of cSceneImitate::UpdateKinChar (SceneImitate.cpp:306-318)
void ref_kinchar_set_time(void* h, double t) { KinChar* k = (KinChar*)h; k->kc->SetTime(t); k->kc->Pose(); }
void ref_kinchar_update(void* h, double dt) { ((KinChar*)h)->kc->Update(dt); }
double ref_kinchar_time(void* h) { return ((KinChar*)h)->kc->GetTime(); }
double ref_kinchar_phase(void* h) { return ((KinChar*)h)->kc->GetPhase(); }
int ref_kinchar_cycle(void* h) { return ((KinChar*)h)->kc->GetCycle(); }
int ref_kinchar_motion_over(void* h) { return ((KinChar*)h)->kc->IsMotionOver() ? 1 : 0; }
void ref_kinchar_state(void* h, double* pose, double* vel) { KinChar* k = (KinChar*)h; vout(k->kc->GetPose(), pose); vout(k->kc->GetVel(), vel); }
// cKinCharacter::SetRootPos / SetRootRotation move the origin (KinCharacter.cpp:270-300): used by SyncKinCharNewCycle
void ref_kinchar_set_root_pos(void* h, const double* p3) { ((KinChar*)h)->kc->SetRootPos(v3in(p3)); }
void ref_kinchar_rotate_root(void* h, const double* q4) { ((KinChar*)h)->kc->RotateRoot(qin(q4)); }
void ref_kinchar_cycle_root_delta(void* h, double* out3) { v3out(((KinChar*)h)->kc->GetCycleRootDelta(), out3); }
// raw cMotion::CalcFrame / CalcFrameVel (no origin, no cycle offset) as cSceneImitateAMP::RecordAMPObsExpert samples them
void ref_kinchar_motion_eval(void* h, double t, double* frame, double* vel) {
    KinChar* k = (KinChar*)h; cMotion::tFrame f, v;
    k->kc->GetMotion()->CalcFrame(t, f); k->kc->GetMotion()->CalcFrameVel(t, v); vout(f, frame); vout(v, vel);
}

// cDynamicTimeWarper (util/DynamicTimeWarper.cpp, compiled) driven as cSceneImitateAMP::BuildTimeWarper / UpdateTimeWarper /
// CalcRewardTimeWarp drive it (scenes/SceneImitateAMP.cpp:173-205,417-460), with the cost function of :5-25 restated here
// (cSceneImitateAMP.cpp itself includes Bullet headers).  data0 [n x dim], data1 [m x dim] row-major; returns CalcAlignment().
static double time_warp_cost_(const Eigen::VectorXd* d0, const Eigen::VectorXd* d1) {
    double cost = 0.0; int num_points = (int)d0->size() / 3;
    for (int p = 0; p < num_points; ++p) {
        tVector a((*d0)(3 * p), (*d0)(3 * p + 1), (*d0)(3 * p + 2), 0), b((*d1)(3 * p), (*d1)(3 * p + 1), (*d1)(3 * p + 2), 0);
        cost += (b - a).norm();
    }
    return cost / num_points;
}
double ref_time_warp(const double* data0, int n, const double* data1, int m, int dim, int buffer_size) {
    cDynamicTimeWarper w;
    w.Init(dim, dim, buffer_size, time_warp_cost_);
    for (int i = 0; i < n; ++i) w.AddSample0(vin(data0 + (size_t)i * dim, dim));
    for (int j = 0; j < m; ++j) w.AddSample1(vin(data1 + (size_t)j * dim, dim));
    return w.CalcAlignment();
}

// cTimer (util/Timer.cpp:55-83): run `n` updates of dt from a reset with max_time, return the index of the first update after
// which IsEnd() is true (or -1)
int ref_timer_first_end(double max_time, double dt, int n) {
    cTimer::tParams p; p.mTimeMin = max_time; p.mTimeMax = max_time;
    cTimer t; t.Init(p); t.Reset();
    for (int i = 0; i < n; ++i) { t.Update(dt); if (t.IsEnd()) return i; }
    return -1;
}

// cTimer::Reset (util/Timer.cpp:55-73) with the process-global generator (cMathUtil::SeedRand): n max times of a timer of the given type
// (0 uniform, 1 exp) and parameters
void ref_timer_draws(int type, double tmin, double tmax, double texp, unsigned long seed, int n, double* out) {
    cMathUtil::SeedRand(seed);
    cTimer::tParams p; p.mType = type ? cTimer::eTypeExp : cTimer::eTypeUniform; p.mTimeMin = tmin; p.mTimeMax = tmax; p.mTimeExp = texp;
    cTimer t; t.Init(p);
    for (int i = 0; i < n; ++i) { t.Reset(); out[i] = t.GetMaxTime(); }
}
// cTimer::tParams::Blend (util/Timer.cpp:12-20) at the lerp cAnnealer::Eval (util/Annealer.cpp; cRLSceneSimChar::SetupTimerAnnealer: pow 4)
// gives for t = sample_count / anneal_samples: out = {lerp, min, max, exp}
void ref_timer_anneal(const double* p0 /*min max exp*/, const double* p1, double t, double* out) {
    cAnnealer::tParams ap; ap.mType = cAnnealer::eTypePow; ap.mPow = 4.0;
    cAnnealer an; an.Init(ap);
    const double lerp = an.Eval(t);
    cTimer::tParams a, b; a.mTimeMin = p0[0]; a.mTimeMax = p0[1]; a.mTimeExp = p0[2]; b.mTimeMin = p1[0]; b.mTimeMax = p1[1]; b.mTimeExp = p1[2];
    cTimer::tParams c = a.Blend(b, lerp);
    out[0] = lerp; out[1] = c.mTimeMin; out[2] = c.mTimeMax; out[3] = c.mTimeExp;
}

// cMathUtil's generator call by call (util/MathUtil.cpp:61-134 -> cRand, util/Rand.cpp): op 0 RandDouble(a, b), 1 RandDoubleExp(a), 2 RandDoubleNorm(a, b),
// 3 RandInt(), 4 RandInt(a, b), 5 RandUint().  ref_math_seed = cMathUtil::SeedRand (util/MathUtil.cpp:114-118: cRand::Seed, then one RandInt for srand).
void ref_math_seed(unsigned long seed) { cMathUtil::SeedRand(seed); }
double ref_math_rand(int op, double a, double b) {
    switch (op) {
    case 0: return cMathUtil::RandDouble(a, b);
    case 1: return cMathUtil::RandDoubleExp(a);
    case 2: return cMathUtil::RandDoubleNorm(a, b);
    case 3: return (double)cMathUtil::RandInt();
    case 4: return (double)cMathUtil::RandInt((int)a, (int)b);
    default: return (double)cMathUtil::RandUint();
    }
}
// The process-global generator cMathUtil::gRand in the order a `--scene imitate` cDeepMimicCore consumes it, driven through the reference's own
// compiled cMathUtil / cRand / cTimer (the scene and ground classes themselves need Bullet; which call comes when is read off their sources):
//   cDeepMimicCore::SeedRand (DeepMimicCore.cpp:20-23)            cMathUtil::SeedRand(seed)
//   cScene::cScene (scenes/Scene.cpp:5)                            mRand.Seed(cMathUtil::RandUint())
//   cRLSceneSimChar::Init (RLSceneSimChar.cpp:27-31)               cScene::Init twice: InitTimers (cTimer::Init -> Reset) + ResetParams (2 x ResetTimers)
//   cGround::cGround (sim/Ground.cpp:68)                           cMathUtil::RandUint()
//   every cScene::Reset (RLSceneSimChar.cpp:234-244, 277-284)      4 x cTimer::Reset, test mode then pins the limit; cSceneImitate::ResetKinChar ->
//                                                                  CalcRandKinResetTime = cMathUtil::RandDouble(0, duration) (SceneImitate.cpp:494-500)
// tparams: (n_resets + 1) x 3 = {min, max, exp} of the timer at Init (row 0) and at each reset (annealing); out: n_resets x 2 = {kin time, time limit};
// init_out: {time limit after Init, first expert-sample time of the scene generator (mRand.RandDouble(0, dur), SceneImitateAMP.cpp:119)}
void ref_rng_session(int seed, int timer_type, const double* tparams, double dur, int n_resets, int test_mode, double test_max, double* out, double* init_out) {
    cMathUtil::SeedRand(seed);
    cRand scene_rand; scene_rand.Seed(cMathUtil::RandUint());
    cTimer timer;
    auto params = [&](int i) { cTimer::tParams p; p.mType = timer_type ? cTimer::eTypeExp : cTimer::eTypeUniform; p.mTimeMin = tparams[3 * i]; p.mTimeMax = tparams[3 * i + 1]; p.mTimeExp = tparams[3 * i + 2]; return p; };
    for (int k = 0; k < 2; ++k) { timer.Init(params(0)); timer.Reset(); timer.Reset(); }
    cMathUtil::RandUint();
    init_out[0] = timer.GetMaxTime();
    for (int i = 0; i < n_resets; ++i) {
        timer.SetParams(params(i + 1));
        for (int k = 0; k < 4; ++k) timer.Reset();
        if (test_mode) timer.SetMaxTime(test_max);
        out[2 * i] = cMathUtil::RandDouble(0, dur);
        out[2 * i + 1] = timer.GetMaxTime();
    }
    init_out[1] = scene_rand.RandDouble(0, dur);
}

}  // extern "C"
