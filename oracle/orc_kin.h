// TEST INFRASTRUCTURE ONLY -- CPU oracle (see orc_math.h header).
//
// orc_kin.h: skeleton tables, forward kinematics, pose/vel arithmetic, motion clip and
// kinematic (reference) character.  Restates
//   cKinTree          /root/reference/DeepMimicCore/anim/KinTree.cpp
//   cMotion           /root/reference/DeepMimicCore/anim/Motion.cpp
//   cKinController / cMotionController   anim/KinController.cpp, anim/MotionController.cpp
//   cKinCharacter     anim/KinCharacter.cpp
#pragma once
#include "orc_math.h"
#include <cassert>
#include <cstdio>
#include <limits>

namespace orc {

enum JointType { JT_REVOLUTE = 0, JT_PLANAR, JT_PRISMATIC, JT_FIXED, JT_SPHERICAL, JT_NONE };
enum { JD_TYPE = 0, JD_PARENT, JD_AX, JD_AY, JD_AZ, JD_ATX, JD_ATY, JD_ATZ, JD_LL0, JD_LL1, JD_LL2,
       JD_LH0, JD_LH1, JD_LH2, JD_TORQUE_LIM, JD_FORCE_LIM, JD_IS_EE, JD_DIFF_W, JD_PARAM_OFFSET, JD_MAX };
enum { BD_SHAPE = 0, BD_MASS, BD_COLGROUP, BD_FALL, BD_AX, BD_AY, BD_AZ, BD_ATX, BD_ATY, BD_ATZ,
       BD_P0, BD_P1, BD_P2, BD_CR, BD_CG, BD_CB, BD_CA, BD_MAX };
enum Shape { SH_NULL = 0, SH_BOX, SH_CAPSULE, SH_SPHERE, SH_CYLINDER, SH_PLANE };

typedef std::vector<real> Vec;

struct Skeleton {
    int J = 0, P = 0;
    std::vector<double> jm;   // J x 19 (cKinTree::tJointDesc rows)
    std::vector<double> bd;   // J x 17 (cKinTree::tBodyDef rows)

    double jd(int j, int c) const { return jm[j * JD_MAX + c]; }
    double bdv(int j, int c) const { return bd[j * BD_MAX + c]; }
    int type(int j) const { return (int)jd(j, JD_TYPE); }
    int parent(int j) const { return (int)jd(j, JD_PARENT); }
    bool is_root(int j) const { return j == 0; }
    int offset(int j) const { return (int)jd(j, JD_PARAM_OFFSET); }
    // cKinTree::GetParamSize (KinTree.cpp:768-802)
    int size(int j) const {
        if (is_root(j)) return 7;
        switch (type(j)) {
            case JT_REVOLUTE: return 1; case JT_PRISMATIC: return 1; case JT_PLANAR: return 3;
            case JT_FIXED: return 0; case JT_SPHERICAL: return 4; default: assert(false); return 0;
        }
    }
    V3 attach_pt(int j) const { return V3(jd(j, JD_AX), jd(j, JD_AY), jd(j, JD_AZ)); }
    V3 attach_theta(int j) const { return V3(jd(j, JD_ATX), jd(j, JD_ATY), jd(j, JD_ATZ)); }
    V3 body_attach_pt(int j) const { return V3(bdv(j, BD_AX), bdv(j, BD_AY), bdv(j, BD_AZ)); }
    V3 body_attach_theta(int j) const { return V3(bdv(j, BD_ATX), bdv(j, BD_ATY), bdv(j, BD_ATZ)); }
    double mass(int j) const { return bdv(j, BD_MASS); }
    int shape(int j) const { return (int)bdv(j, BD_SHAPE); }
    bool valid_body(int j) const { return shape(j) != SH_NULL; }
    bool is_end_eff(int j) const { return jd(j, JD_IS_EE) != 0; }

    void init(const double* joint_mat, const double* body_defs, int nj) {
        J = nj;
        jm.assign(joint_mat, joint_mat + nj * JD_MAX);
        bd.assign(body_defs, body_defs + nj * BD_MAX);
        // cKinTree::PostProcessJointMat (KinTree.cpp:1005-1020)
        int off = 0;
        for (int j = 0; j < J; ++j) { jm[j * JD_MAX + JD_PARAM_OFFSET] = off; off += size(j); }
        jm[JD_AX] = jm[JD_AY] = jm[JD_AZ] = 0;
        P = off;
    }
};

static inline V3 root_pos(const Vec& p) { return V3(p[0], p[1], p[2]); }
static inline void set_root_pos(Vec& p, const V3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline Q4 root_rot(const Vec& p) { return Q4(p[3], p[4], p[5], p[6]); }
static inline void set_root_rot(Vec& p, const Q4& q) { p[3] = q.w; p[4] = q.x; p[5] = q.y; p[6] = q.z; }
static inline V3 root_vel(const Vec& v) { return V3(v[0], v[1], v[2]); }
static inline V3 root_ang_vel(const Vec& v) { return V3(v[3], v[4], v[5]); }   // slot 6 is the dead 4th entry
static inline Q4 joint_quat(const Vec& p, int off) { return Q4(p[off], p[off + 1], p[off + 2], p[off + 3]); }
static inline void set_joint_quat(Vec& p, int off, const Q4& q) { p[off] = q.w; p[off + 1] = q.x; p[off + 2] = q.y; p[off + 3] = q.z; }

// cKinTree::BuildAttachTrans (KinTree.cpp:1022-1032)
static inline Xf attach_trans(const Skeleton& s, int j) { return Xf(rot_euler(s.attach_theta(j)), s.attach_pt(j)); }

// cKinTree::ChildParentTrans{Root,Revolute,Fixed,Spherical} (KinTree.cpp:1034-1069,1758-1830)
static inline Xf child_parent_trans(const Skeleton& s, const Vec& pose, int j) {
    Xf A = attach_trans(s, j);
    if (s.is_root(j)) {
        Xf R(rot_quat(root_rot(pose)), V3());
        Xf T(M3::identity(), root_pos(pose));
        return A * T * R;
    }
    switch (s.type(j)) {
        case JT_REVOLUTE: return A * Xf(rot_axis(V3(0, 0, 1), pose[s.offset(j)]), V3());
        case JT_FIXED: return A;
        case JT_SPHERICAL: return A * Xf(rot_quat(joint_quat(pose, s.offset(j))), V3());
        default: assert(false && "unsupported joint type on the imitate path"); return A;
    }
}
// cKinTree::JointWorldTrans (KinTree.cpp:1078-1091)
static inline Xf joint_world_trans(const Skeleton& s, const Vec& pose, int j) {
    Xf m;
    for (int c = j; c != -1; c = s.parent(c)) m = child_parent_trans(s, pose, c) * m;
    return m;
}
// cKinTree::BodyJointTrans (KinTree.cpp:1108-1118); local COM is zero for every supported shape
static inline Xf body_joint_trans(const Skeleton& s, int j) {
    return Xf(M3::identity(), s.body_attach_pt(j)) * Xf(rot_euler(s.body_attach_theta(j)), V3());
}
// cKinTree::CalcJointWorldPos (KinTree.cpp:538-553)
static inline V3 joint_world_pos(const Skeleton& s, const Vec& pose, int j) { return joint_world_trans(s, pose, j).t; }

// cKinTree::CalcHeading (KinTree.cpp:1619-1627)
static inline real calc_heading(const Q4& rot) {
    V3 d = qrot(rot, V3(1, 0, 0));
    return std::atan2(-d.z, d.x);
}
// cKinTree::BuildHeadingTrans / BuildOriginTrans (KinTree.cpp:1645-1664): world -> origin frame
static inline Xf origin_trans(const Vec& pose) {
    real heading = calc_heading(root_rot(pose));
    M3 R = rot_axis(V3(0, 1, 0), -heading);
    V3 o = root_pos(pose); o.y = 0;
    return Xf(R, V3()) * Xf(M3::identity(), -o);
}

// cKinTree::PostProcessPose (KinTree.cpp:1510-1527)
static inline void post_process_pose(const Skeleton& s, Vec& pose) {
    set_root_rot(pose, qnormalized(root_rot(pose)));
    for (int j = 1; j < s.J; ++j)
        if (s.type(j) == JT_SPHERICAL) set_joint_quat(pose, s.offset(j), qnormalized(joint_quat(pose, s.offset(j))));
}
// cKinTree::CalcVel (KinTree.cpp:1470-1508)
static inline void calc_vel(const Skeleton& s, const Vec& p0, const Vec& p1, real dt, Vec& out) {
    out.assign(s.P, 0);
    V3 rv = (root_pos(p1) - root_pos(p0)) / dt;
    V3 ra = quat_vel(root_rot(p0), root_rot(p1), dt);
    out[0] = rv.x; out[1] = rv.y; out[2] = rv.z; out[3] = ra.x; out[4] = ra.y; out[5] = ra.z; out[6] = 0;
    for (int j = 1; j < s.J; ++j) {
        int off = s.offset(j), sz = s.size(j);
        if (s.type(j) == JT_SPHERICAL) {
            V3 w = quat_vel_rel(joint_quat(p0, off), joint_quat(p1, off), dt);
            out[off] = w.x; out[off + 1] = w.y; out[off + 2] = w.z; out[off + 3] = 0;
        } else {
            for (int k = 0; k < sz; ++k) out[off + k] = (p1[off + k] - p0[off + k]) / dt;
        }
    }
}
// cKinTree::LerpPoses (KinTree.cpp:1529-1572)
static inline void lerp_poses(const Skeleton& s, const Vec& p0, const Vec& p1, real lerp, Vec& out) {
    out.assign(s.P, 0);
    V3 rp = (1 - lerp) * root_pos(p0) + lerp * root_pos(p1);
    Q4 rr = qnormalized(slerp(root_rot(p0), lerp, root_rot(p1)));
    set_root_pos(out, rp); set_root_rot(out, rr);
    for (int j = 1; j < s.J; ++j) {
        int off = s.offset(j), sz = s.size(j);
        if (s.type(j) == JT_SPHERICAL) set_joint_quat(out, off, slerp(joint_quat(p0, off), lerp, joint_quat(p1, off)));
        else for (int k = 0; k < sz; ++k) out[off + k] = (1 - lerp) * p0[off + k] + lerp * p1[off + k];
    }
}
// cKinTree::VelToPoseDiff (KinTree.cpp:1579-1610)
static inline void vel_to_pose_diff(const Skeleton& s, const Vec& pose, const Vec& vel, Vec& out) {
    out = vel;
    set_root_rot(out, quat_diff_mul(root_rot(pose), root_ang_vel(vel)));
    for (int j = 1; j < s.J; ++j) {
        if (s.type(j) == JT_SPHERICAL) {
            int off = s.offset(j);
            Q4 d = quat_diff_mul(joint_quat(pose, off), V3(vel[off], vel[off + 1], vel[off + 2]));
            set_joint_quat(out, off, d);
        }
    }
}

// ---------------------------------------------------------------- motion clip
struct Motion {
    int F = 0, P = 0;
    bool loop = false;
    std::vector<double> times;       // frame start times (cumulative)
    std::vector<Vec> frames;         // F x P
    std::vector<Vec> frame_vel;      // F x P
    V3 cycle_root_delta;             // cKinController::CalcCycleRootDelta

    double duration() const { return times[F - 1]; }   // Motion.cpp:432-437

    // cMotion::Load + cKinController::LoadMotion (Motion.cpp:104-141,302-430; KinController.cpp:104-147)
    void load(const Skeleton& s, const double* raw, int nf, int np, bool wrap) {
        F = nf; P = np; loop = wrap;
        assert(np == s.P);
        times.assign(F, 0); frames.assign(F, Vec(P, 0));
        for (int f = 0; f < F; ++f) {
            times[f] = raw[f * (P + 1)];
            for (int k = 0; k < P; ++k) frames[f][k] = (real)raw[f * (P + 1) + 1 + k];
        }
        // PostProcessFrames (Motion.cpp:403-430): durations -> start times; centre on frame 0 (x,z);
        // the kin character's post-process func normalises quaternions (KinCharacter.cpp:443-447).
        V3 off0 = root_pos(frames[0]); off0.y = 0;
        double t = 0;
        for (int f = 0; f < F; ++f) {
            double dur = times[f];
            times[f] = t; t += dur;
            set_root_pos(frames[f], root_pos(frames[f]) - off0);
            post_process_pose(s, frames[f]);
        }
        // UpdateVel -> BuildFrameVel (Motion.cpp:170-191) with cKinTree::CalcVel as the vel func
        frame_vel.assign(F, Vec(P, 0));
        for (int f = 0; f < F - 1; ++f) {
            real dt = (real)(times[f + 1] - times[f]);
            calc_vel(s, frames[f], frames[f + 1], dt, frame_vel[f]);
        }
        if (F > 1) frame_vel[F - 1] = frame_vel[F - 2];
        // cKinController::PostProcessMotion (KinController.cpp:131-147)
        V3 beg = root_pos(frames[0]);
        for (int f = 0; f < F; ++f) { V3 rp = root_pos(frames[f]); rp.x -= beg.x; rp.z -= beg.z; set_root_pos(frames[f], rp); }
        // cKinController::CalcCycleRootDelta (KinController.cpp:149-161)
        cycle_root_delta = root_pos(frames[F - 1]) - root_pos(frames[0]);
        cycle_root_delta.y = 0;
    }
    // cMotion::CalcCycleCount (Motion.cpp:476-484)
    int cycle_count(double time) const {
        int c = (int)std::floor(time / duration());
        return loop ? c : std::max(0, std::min(c, 1));
    }
    // cMotion::CalcPhase (Motion.cpp:32-47,469-474)
    double phase(double time) const {
        double ph = time / duration();
        if (loop) ph -= std::floor(ph); else ph = std::max(0.0, std::min(ph, 1.0));
        return ph;
    }
    // cMotion::CalcIndexBlend (Motion.cpp:486-515)
    void index_blend(double time, int& idx, double& blend) const {
        double max_time = duration();
        if (!loop) {
            if (time <= 0) { idx = 0; blend = 0; return; }
            else if (time >= max_time) { idx = F - 2; blend = 1; return; }
        }
        int cc = cycle_count(time);
        time -= cc * duration();
        const double* it = std::upper_bound(times.data(), times.data() + F, time);
        idx = (int)(it - times.data() - 1);
        double t0 = times[idx], t1 = times[idx + 1];
        blend = (time - t0) / (t1 - t0);
    }
    // cMotion::CalcFrame -> BlendFrames with cKinTree::LerpPoses (Motion.cpp:249-274)
    void calc_frame(const Skeleton& s, double time, Vec& out) const {
        int idx; double blend; index_blend(time, idx, blend);
        blend = std::max(0.0, std::min(blend, 1.0));   // cMathUtil::Saturate
        lerp_poses(s, frames[idx], frames[idx + 1], (real)blend, out);
    }
    // cMotion::CalcFrameVel (Motion.cpp:276-293)
    void calc_frame_vel(double time, Vec& out) const {
        if (!loop && time >= duration()) { out.assign(P, 0); return; }
        int idx; double blend; index_blend(time, idx, blend);
        out.assign(P, 0);
        for (int k = 0; k < P; ++k) out[k] = (real)((1.0 - blend) * frame_vel[idx][k] + blend * frame_vel[idx + 1][k]);
    }
    bool is_over(double time) const { return !loop && time >= duration(); }
};

// ---------------------------------------------------------------- kinematic character
struct KinChar {
    const Skeleton* sk = nullptr;
    const Motion* mo = nullptr;
    double time = 0;          // cKinController::mTime
    V3 origin;                // cKinCharacter::mOrigin
    Q4 origin_rot;            // cKinCharacter::mOriginRot
    Vec pose, vel;

    // cMotionController::CalcPose + cKinCharacter::CalcPose (MotionController.cpp:25-41,102-111; KinCharacter.cpp:363-386)
    void calc_pose(double t, Vec& out) const {
        mo->calc_frame(*sk, t, out);
        if (mo->loop) {
            int cc = mo->cycle_count(t);
            set_root_pos(out, root_pos(out) + (real)cc * mo->cycle_root_delta);
        }
        V3 rp = root_pos(out);
        Q4 rr = standardize(origin_rot * root_rot(out));
        rp = qrot(origin_rot, rp) + origin;
        set_root_pos(out, rp); set_root_rot(out, rr);
    }
    // cKinCharacter::CalcVel (KinCharacter.cpp:388-406)
    void calc_vel(double t, Vec& out) const {
        mo->calc_frame_vel(t, out);
        V3 v = qrot(origin_rot, root_vel(out)), w = qrot(origin_rot, root_ang_vel(out));
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = w.x; out[4] = w.y; out[5] = w.z;
    }
    // cKinCharacter::Pose (KinCharacter.cpp:199-211)
    void do_pose() { calc_pose(time, pose); calc_vel(time, vel); }
    // cKinCharacter::Update (KinCharacter.cpp:72-82)
    void update(double dt) { time += dt; do_pose(); }
    double phase() const { return mo->phase(time); }
    // cKinCharacter::MoveOrigin / SetRootPos (KinCharacter.cpp:236-273)
    void set_root_pos_(const V3& p) {
        V3 delta = p - root_pos(pose);
        origin += delta;
        set_root_pos(pose, root_pos(pose) + delta);
    }
    // cKinCharacter::RotateOrigin (KinCharacter.cpp:287-322) -- used by sync_char_root_rot / rand rot reset
    void rotate_origin(const Q4& rot) {
        origin_rot = qnormalized(rot * origin_rot);
        V3 rp = root_pos(pose);
        origin = rp + qrot(rot, origin - rp);
        set_root_rot(pose, qnormalized(rot * root_rot(pose)));
        V3 v = qrot(rot, root_vel(vel)), w = qrot(rot, root_ang_vel(vel));
        vel[0] = v.x; vel[1] = v.y; vel[2] = v.z; vel[3] = w.x; vel[4] = w.y; vel[5] = w.z;
    }
    // cKinCharacter::RotateRoot == cCharacter::RotateRoot -> SetRootRotation -> RotateOrigin(dq)
    void rotate_root(const Q4& drot) { rotate_origin(drot); }
};

}  // namespace orc
