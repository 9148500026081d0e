// TEST INFRASTRUCTURE ONLY -- CPU oracle (see orc_math.h header).
//
// orc_rbd.h: rigid-body-dynamics model in joint space.  Restates
//   cRBDModel   /root/reference/DeepMimicCore/sim/RBDModel.cpp:36-46,183-238
//   cRBDUtil    /root/reference/DeepMimicCore/sim/RBDUtil.cpp
//       SolveInvDyna (RNEA) 4-97, BuildMassMat (CRBA) 123-195, BuildEndEffectorJacobian 225-249,
//       CalcCoM 572-613, BuildMomentInertia* 615-749, BuildInertiaSpatialMat 742-749,
//       BuildJointSubspace* 798-893, BuildCj* 895-995, BuildBiasForce 997-1001
// Spatial quantities live in each joint's own frame exactly as in the reference.
//
// Two knobs exist because the simulator (Bullet in the reference) and the SPD controller
// disagree on two details:
//   inertia_model : 0 = DeepMimic analytic shapes (RBDUtil.cpp:644-740, used by SPD)
//                   1 = Bullet 2.88 calculateLocalInertia [EXT-BULLET]: identical for box and
//                       sphere, capsule = box formula on (2(r+m), 2(r+h/2+m), 2(r+m)), m = 0.04/world_scale
//   exact_root_cj : false = reference BuildCjRoot (treats the world-frame root angular
//                   velocity as body-frame when differentiating E^T; used by SPD),
//                   true  = exact S-dot*qdot for the floating base (used by the simulator)
#pragma once
#include "orc_kin.h"

namespace orc {

struct RBDOpts { int inertia_model = 0; bool exact_root_cj = false; double world_scale = 4.0; };

// 6 x P joint subspace, stored per joint as up to 7 columns
struct Subspace { int n; SV col[7]; };

struct RBDModel {
    const Skeleton* sk = nullptr;
    RBDOpts opt;
    V3 gravity = V3(0, (real)-9.8, 0);
    Vec pose, vel;
    std::vector<Xf> child_parent;      // cRBDModel::mChildParentMatArr
    std::vector<ST> world_joint;       // cRBDModel::mSpWorldJointTransArr
    std::vector<Subspace> S;
    std::vector<real> H;               // P x P
    Vec C;                             // P

    void init(const Skeleton* s, const RBDOpts& o, const V3& g) { sk = s; opt = o; gravity = g; }

    // RBDUtil.cpp:644-740 (and the Bullet variant for capsules)
    SM moment_inertia(int j) const {
        real mass = (real)sk->mass(j);
        real x = 0, y = 0, z = 0;
        real p0 = (real)sk->bdv(j, BD_P0), p1 = (real)sk->bdv(j, BD_P1), p2 = (real)sk->bdv(j, BD_P2);
        switch (sk->shape(j)) {
            case SH_BOX:
                x = mass / 12 * (p1 * p1 + p2 * p2); y = mass / 12 * (p0 * p0 + p2 * p2); z = mass / 12 * (p0 * p0 + p1 * p1);
                break;
            case SH_CAPSULE: {
                real r = (real)0.5 * p0, h = p1;
                if (opt.inertia_model == 1) {
                    // [EXT-BULLET] btCapsuleShape::calculateLocalInertia: box of half extents (r, r+h/2, r) + margin
                    real mg = (real)(0.04 / opt.world_scale);
                    real lx = 2 * (r + mg), ly = 2 * (r + (real)0.5 * h + mg), lz = 2 * (r + mg);
                    x = mass / 12 * (ly * ly + lz * lz); y = mass / 12 * (lx * lx + lz * lz); z = mass / 12 * (lx * lx + ly * ly);
                } else {
                    real c_vol = kPi * r * r * h, hs_vol = kPi * (real)2 / 3 * r * r * r;
                    real density = mass / (c_vol + 2 * hs_vol);
                    real cm = c_vol * density, hsm = hs_vol * density;
                    x = cm * ((real)0.25 * r * r + ((real)1 / 12) * h * h) + 2 * hsm * ((real)0.4 * r * r + ((real)3 / 8) * r * h + (real)0.25 * h * h);
                    y = ((real)0.5 * cm + (real)0.8 * hsm) * r * r;
                    z = x;
                }
                break;
            }
            case SH_SPHERE: { real r = (real)0.5 * p0; x = y = z = (real)0.4 * mass * r * r; break; }
            case SH_CYLINDER: { real r = (real)0.5 * p0, h = p1; x = mass / 12 * (3 * r * r + h * h); y = mass * r * r / 2; z = x; break; }
            default: assert(false);
        }
        SM I; I.m[0][0] = x; I.m[1][1] = y; I.m[2][2] = z; I.m[3][3] = I.m[4][4] = I.m[5][5] = mass;
        return I;
    }
    // RBDUtil.cpp:742-749
    SM inertia_spatial(int j) const {
        SM Ic = moment_inertia(j);
        ST X = mat_to_trans(body_joint_trans(*sk, j));
        return spatial_mat_f(X) * Ic * spatial_mat_m(inv_trans(X));
    }
    ST sp_child_parent(int j) const { return mat_to_trans(child_parent[j]); }
    ST sp_parent_child(int j) const { return mat_to_trans(inv_rigid(child_parent[j])); }

    // RBDUtil.cpp:798-893
    Subspace joint_subspace(int j) const {
        Subspace s; s.n = sk->size(j);
        for (int i = 0; i < 7; ++i) s.col[i] = SV();
        if (sk->is_root(j)) {
            M3 E = rot_quat(root_rot(pose));
            M3 Et = transpose(E);
            for (int c = 0; c < 3; ++c) {
                s.col[c].v = V3(Et.m[0][c], Et.m[1][c], Et.m[2][c]);       // S.block(3,0) = E^T
                s.col[3 + c].o = V3(Et.m[0][c], Et.m[1][c], Et.m[2][c]);   // S.block(0,3) = E^T (col 6 stays 0)
            }
            return s;
        }
        switch (sk->type(j)) {
            case JT_REVOLUTE: s.col[0].o = V3(0, 0, 1); break;
            case JT_FIXED: break;
            case JT_SPHERICAL: s.col[0].o = V3(1, 0, 0); s.col[1].o = V3(0, 1, 0); s.col[2].o = V3(0, 0, 1); break;
            default: assert(false);
        }
        return s;
    }
    // RBDUtil.cpp:895-995
    SV build_cj(int j) const {
        if (!sk->is_root(j)) return SV();
        Q4 quat = root_rot(pose);
        V3 vel_lin = root_vel(vel), vel_ang = root_ang_vel(vel);
        if (opt.exact_root_cj) {
            // d/dt(E^T) v = -(E^T w) x (E^T v) for a world-frame angular velocity w
            M3 Et = transpose(rot_quat(quat));
            return SV(V3(), -cross(Et * vel_ang, Et * vel_lin));
        }
        Q4 dq = quat_diff_mul(quat, vel_ang);
        M3 mat;
        mat.m[0][0] = 4 * (quat.w * dq.w + quat.x * dq.x);
        mat.m[1][1] = 4 * (quat.w * dq.w + quat.y * dq.y);
        mat.m[2][2] = 4 * (quat.w * dq.w + quat.z * dq.z);
        mat.m[1][0] = 2 * (dq.x * quat.y + quat.x * dq.y - dq.w * quat.z - quat.w * dq.z);
        mat.m[0][1] = 2 * (dq.x * quat.y + quat.x * dq.y + dq.w * quat.z + quat.w * dq.z);
        mat.m[2][0] = 2 * (dq.x * quat.z + quat.x * dq.z + dq.w * quat.y + quat.w * dq.y);
        mat.m[0][2] = 2 * (dq.x * quat.z + quat.x * dq.z - dq.w * quat.y - quat.w * dq.y);
        mat.m[2][1] = 2 * (dq.y * quat.z + quat.y * dq.z - dq.w * quat.x - quat.w * dq.x);
        mat.m[1][2] = 2 * (dq.y * quat.z + quat.y * dq.z + dq.w * quat.x + quat.w * dq.x);
        return SV(V3(), mat * vel_lin);
    }

    // cRBDModel::Update (RBDModel.cpp:36-46)
    void update(const Vec& p, const Vec& v) {
        const int J = sk->J, P = sk->P;
        pose = p; vel = v;
        S.resize(J); child_parent.resize(J); world_joint.resize(J);
        for (int j = 0; j < J; ++j) S[j] = joint_subspace(j);
        for (int j = 0; j < J; ++j) child_parent[j] = child_parent_trans(*sk, pose, j);
        // CalcWorldJointTransforms (RBDUtil.cpp:751-775)
        for (int j = 0; j < J; ++j) {
            ST wp; int par = sk->parent(j);
            if (par != -1) wp = world_joint[par];
            world_joint[j] = comp_trans(sp_parent_child(j), wp);
        }
        build_mass_mat();
        Vec acc(P, 0);
        solve_inv_dyna(acc, C);
    }

    // RBDUtil.cpp:123-195
    void build_mass_mat() {
        const int J = sk->J, P = sk->P;
        H.assign((size_t)P * P, 0);
        std::vector<SM> Is(J), cpF(J), pcM(J);
        for (int j = 0; j < J; ++j) {
            if (sk->valid_body(j)) Is[j] = inertia_spatial(j);
            ST cp = sp_child_parent(j);
            cpF[j] = spatial_mat_f(cp);
            pcM[j] = spatial_mat_m(inv_trans(cp));
        }
        for (int j = J - 1; j >= 0; --j) {
            if (!sk->valid_body(j)) continue;
            int par = sk->parent(j);
            if (par != -1) Is[par] = Is[par] + cpF[j] * Is[j] * pcM[j];
            int dim = sk->size(j);
            if (dim > 0) {
                int off = sk->offset(j);
                std::vector<SV> F(dim);
                for (int c = 0; c < dim; ++c) F[c] = Is[j] * S[j].col[c];
                for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) H[(off + a) * P + off + b] = svdot(S[j].col[a], F[b]);
                int cur = j;
                while (sk->parent(cur) != -1) {
                    for (int c = 0; c < dim; ++c) F[c] = cpF[cur] * F[c];
                    cur = sk->parent(cur);
                    int coff = sk->offset(cur), cdim = sk->size(cur);
                    for (int a = 0; a < dim; ++a) for (int b = 0; b < cdim; ++b) {
                        real val = svdot(F[a], S[cur].col[b]);
                        H[(off + a) * P + coff + b] = val;
                        H[(coff + b) * P + off + a] = val;
                    }
                }
            }
        }
    }

    // RBDUtil.cpp:4-97
    void solve_inv_dyna(const Vec& acc, Vec& out_tau) const {
        const int J = sk->J, P = sk->P;
        SV vel0, acc0(V3(), -gravity);
        std::vector<SV> vels(J), accs(J), fs(J);
        for (int j = 0; j < J; ++j) {
            if (!sk->valid_body(j)) continue;
            ST pc = sp_parent_child(j);
            int off = sk->offset(j), n = sk->size(j);
            SV cj = build_cj(j);
            SV vj, Sddq;
            for (int c = 0; c < n; ++c) { vj = vj + vel[off + c] * S[j].col[c]; Sddq = Sddq + acc[off + c] * S[j].col[c]; }
            SM I = inertia_spatial(j);
            int par = sk->parent(j);
            SV vp = (par != -1) ? vels[par] : vel0;
            SV ap = (par != -1) ? accs[par] : acc0;
            SV cv = apply_trans_m(pc, vp) + vj;
            SV ca = apply_trans_m(pc, ap) + Sddq + cj + cross_m(cv, vj);
            SV cf = I * ca + cross_f(cv, I * cv);
            vels[j] = cv; accs[j] = ca; fs[j] = cf;
        }
        out_tau.assign(P, 0);
        for (int j = J - 1; j >= 0; --j) {
            if (!sk->valid_body(j)) continue;
            int off = sk->offset(j), n = sk->size(j);
            for (int c = 0; c < n; ++c) out_tau[off + c] = svdot(S[j].col[c], fs[j]);
            int par = sk->parent(j);
            if (par != -1) fs[par] = fs[par] + apply_trans_f(sp_child_parent(j), fs[j]);
        }
    }
};

// cRBDUtil::BuildEndEffectorJacobian * vel  (RBDUtil.cpp:225-249,490-496): world spatial velocity of joint j's frame
static inline SV calc_world_vel(const Skeleton& sk, const Vec& pose, const Vec& vel, int joint_id) {
    RBDModel m; m.sk = &sk; m.pose = pose; m.vel = vel;     // only joint_subspace() is used
    ST cur;                       // identity
    std::vector<std::pair<int, SV> > cols;                   // (pose index, column)
    for (int c = joint_id; c != -1; c = sk.parent(c)) {
        Subspace S = m.joint_subspace(c);
        int off = sk.offset(c);
        for (int k = 0; k < S.n; ++k) cols.push_back(std::make_pair(off + k, apply_trans_m(cur, S.col[k])));
        Xf pcm = inv_rigid(child_parent_trans(sk, pose, c));
        cur = comp_trans(cur, mat_to_trans(pcm));
    }
    SV sv;
    for (size_t i = 0; i < cols.size(); ++i) sv = sv + vel[cols[i].first] * apply_inv_trans_m(cur, cols[i].second);
    return sv;
}
// cRBDUtil::CalcCoM (RBDUtil.cpp:572-613)
static inline void calc_com(const Skeleton& sk, const Vec& pose, const Vec& vel, V3& out_com, V3& out_vel) {
    out_com = V3(); out_vel = V3();
    real total = 0;
    for (int j = 0; j < sk.J; ++j) {
        if (!sk.valid_body(j)) continue;
        Xf body_world = joint_world_trans(sk, pose, j) * body_joint_trans(sk, j);
        V3 world_com = body_world.t;
        ST com_trans(M3::identity(), world_com);
        SV sv = apply_trans_m(com_trans, calc_world_vel(sk, pose, vel, j));
        real m = (real)sk.mass(j);
        out_com += m * world_com; out_vel += m * sv.v; total += m;
    }
    out_com = out_com / total; out_vel = out_vel / total;
}

// Dense symmetric solve used for H.ldlt().solve(b) (Eigen LDLT, RBDUtil.cpp:111, ImpPDController.cpp:188).
// Restated as an LDL^T factorisation on the index set with non-zero diagonal; rows/columns whose
// diagonal is exactly zero (the dead 4th slot of the root quaternion) get a
// zero solution component, which is what Eigen's pivoting LDLT returns for those entries.
static inline void ldlt_solve(const std::vector<real>& A, int n, const Vec& b, Vec& x) {
    std::vector<int> idx;
    for (int i = 0; i < n; ++i) if (A[(size_t)i * n + i] != 0) idx.push_back(i);
    int m = (int)idx.size();
    std::vector<real> L((size_t)m * m, 0), D(m, 0);
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j <= i; ++j) {
            real s = A[(size_t)idx[i] * n + idx[j]];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * m + k] * L[(size_t)j * m + k] * D[k];
            if (i == j) { D[i] = s; L[(size_t)i * m + i] = 1; }
            else L[(size_t)i * m + j] = s / D[j];
        }
    }
    Vec y(m, 0);
    for (int i = 0; i < m; ++i) { real s = b[idx[i]]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * m + k] * y[k]; y[i] = s; }
    for (int i = 0; i < m; ++i) y[i] /= D[i];
    for (int i = m - 1; i >= 0; --i) { real s = y[i]; for (int k = i + 1; k < m; ++k) s -= L[(size_t)k * m + i] * y[k]; y[i] = s; }
    x.assign(n, 0);
    for (int i = 0; i < m; ++i) x[idx[i]] = y[i];
}

}  // namespace orc
