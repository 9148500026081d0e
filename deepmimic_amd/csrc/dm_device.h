// Device code of the imitate hot path: one wavefront (64 lanes) simulates one character.
//
//   lanes <-> links      for kinematics / Newton-Euler passes (15 or 23 of 64 lanes active)
//   lanes <-> dofs       for mass-matrix rows, Cholesky rows, triangular solves (34 / 64 lanes)
//   lanes <-> contact candidates, then constraint rows, for collision + PGS (<= 64 rows)
//
// All per-env working data lives in LDS for the whole call (20 scene updates = one control step); HBM is
// touched only to load the env record at entry and to store record + observation + reward at exit.
// The workgroup is exactly one wavefront, so __syncthreads() is a wave-local LDS fence.
//
// Reference functions realised here (DeepMimicCore/...):
//   kin_*        anim/Motion.cpp:249-293,476-515; anim/MotionController.cpp:25-47,102-111; anim/KinCharacter.cpp:363-406
//   dynamics()   sim/RBDUtil.cpp:4-97 (RNEA), 123-195 (CRBA) -- same H and C, evaluated as a Newton-Euler pass in
//                world-aligned axes about each joint's own origin (no 6x6 frame transforms; fp32-safe)
//   spd()        sim/ImpPDController.cpp:136-195, sim/SimBodyJoint.cpp:299-307,636-695
//   substep()    DM-physics v1 standing in for btMultiBodyDynamicsWorld::stepSimulation
//   emit()       scenes/SceneImitate.cpp:7-127,163-205; sim/CtController.cpp:281-478; sim/SimCharacter.cpp:542-586
//   reset_env()  scenes/SceneSimChar.cpp:487-583,628-644; scenes/SceneImitate.cpp:320-368,386-418
#pragma once
#include "dm_math.h"
#include "dm_types.h"

#ifdef DM_EMU
static inline long long dm_clock() { return 0; }
#define DM_DEV inline
static inline int dm_atomic_or(int* p, int v) { int o = *p; *p = o | v; return o; }
namespace dmk {
template <typename T> static inline T wave_bcast(T v, int src) {
    T* x = reinterpret_cast<T*>(emu::g_xchg);
    x[threadIdx.x] = v; __syncthreads(); T r = x[src]; __syncthreads(); return r;
}
template <typename T> static inline T lane_bcast(T v, int src) { return wave_bcast(v, src); }
template <typename Real> struct RowFile {       // per-lane array indexed by a wave-uniform runtime index
    Real v[kMaxRows];
    inline Real get(int r) const { return v[r]; }
    inline void set(int r, Real x) { v[r] = x; }
};
}
#else
#define DM_DEV __device__ __forceinline__
__device__ __forceinline__ long long dm_clock() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ int dm_atomic_or(int* p, int v) { return atomicOr(p, v); }
namespace dmk {
__device__ __forceinline__ float wave_bcast(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double wave_bcast(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int wave_bcast(int v, int src) { return __shfl(v, src, 64); }
// broadcast from a wave-uniform lane: v_readlane_b32 (no LDS traffic, SGPR result)
__device__ __forceinline__ float lane_bcast(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ double lane_bcast(double v, int src) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// per-lane array indexed by a wave-uniform runtime index.  For float it is two 32-wide register vectors that the
// backend addresses with v_movrels/v_movreld (M0-relative VGPR indexing), so the row of A never leaves the VGPRs.
template <typename Real> struct RowFile {
    Real v[kMaxRows];
    __device__ __forceinline__ Real get(int r) const { return v[r]; }
    __device__ __forceinline__ void set(int r, Real x) { v[r] = x; }
};
template <> struct RowFile<float> {
    typedef float v32 __attribute__((ext_vector_type(32)));
    v32 a, b;
    __device__ __forceinline__ float get(int r) const { return (r < 32) ? a[r] : b[r - 32]; }
    __device__ __forceinline__ void set(int r, float x) { if (r < 32) a[r] = x; else b[r - 32] = x; }
};
}
#endif

namespace dmk {

template <typename Real, int NJ, int ND, int NP, int NCAP>
struct Lds {
    static constexpr int kYStride = kMaxRows + 1;
    Real pose[NP], vel[NP], tar[NP];
    Real qd[ND], tau[ND], bias[ND], rhs[ND], vstar[ND], xs[ND];
    Real R[NJ][9], p[NJ][3], com[NJ][3], Rb[NJ][9], w[NJ][3], vj[NJ][3], al[NJ][3], aj[NJ][3];
    Real f[NJ][3], n[NJ][3], Iw[NJ][6];
    Real Fs[NJ][3], Ns[NJ][3], Ic[NJ][10];
    Real axis[ND][3];
    static constexpr int kHS = ((ND + 3) / 4) * 4 + (((((ND + 3) / 4)) % 2 == 0) ? 4 : 0);   // row stride: 16-B multiple, odd count of 16-B slots
    alignas(16) Real H[ND][kHS];
    Real dinv[ND];                         // 1 / L_kk of the current Cholesky factor
    Real scratch[ND * kYStride];          // Y = L^-1 J^T during the constraint solve; kin pose / vel at emit time
    Real row_b[kMaxRows], lam[kMaxRows];
    Real cx[NCAP][3], cdist[NCAP];     // ground-contact candidates (NCAP = 64 or 128)
    int csel[NCAP], cslot[kMaxRows];
    Real kin[8];                           // kin origin pos(3), origin rot(4)
    Real sc[24];                           // small float scratch
    double clk[6];                         // kin_time, ctrl_time, init_time_offset, timer_time, timer_max
    int flg[8];                            // need_new_action, contact_mask, episode_count, valid, nrows, ncontacts
};

enum { CLK_KIN = 0, CLK_CTRL, CLK_INIT_OFF, CLK_TIMER, CLK_TIMER_MAX };
enum { FLG_NEED_ACTION = 0, FLG_CONTACT, FLG_EPISODE, FLG_VALID, FLG_NROWS, FLG_NCONT };

template <typename Real, int NJ, int ND, int NP, int NCAP>
struct EnvSim {
    typedef Lds<Real, NJ, ND, NP, NCAP> L;
    typedef V3<Real> v3; typedef Q4<Real> q4; typedef M3<Real> m3;
    const ModelDev<Real>& m; L& s; const int l;
    long long* prof = nullptr; long long tprev = 0;     // phase-cycle accounting (profiling kernel only)
    DM_DEV EnvSim(const ModelDev<Real>& m_, L& s_, int l_) : m(m_), s(s_), l(l_) {}
    DM_DEV void mark(int phase) { if (prof) { long long t = dm_clock(); if (l == 0) prof[phase] += t - tprev; tprev = t; } }
    DM_DEV void sync() const { __syncthreads(); }
    DM_DEV Real* Y(int k) const { return s.scratch + k * L::kYStride; }

    // ------------------------------------------------------------------ HBM <-> LDS
    DM_DEV void load(const EnvState<Real>& st, int e) {
        for (int i = l; i < m.P; i += kWave) { s.pose[i] = st.pose[(size_t)e * m.P + i]; s.vel[i] = st.vel[(size_t)e * m.P + i]; s.tar[i] = st.tar[(size_t)e * m.P + i]; }
        if (l < m.D) s.tau[l] = st.tau[(size_t)e * m.D + l];
        if (l < 8) s.kin[l] = st.kin[(size_t)e * 8 + l];
        if (l < 6) s.clk[l] = st.clock[(size_t)e * 6 + l];
        if (l < 4) s.flg[l] = st.flag[(size_t)e * 4 + l];
        sync();
    }
    DM_DEV void store(const EnvState<Real>& st, int e) {
        sync();
        for (int i = l; i < m.P; i += kWave) { st.pose[(size_t)e * m.P + i] = s.pose[i]; st.vel[(size_t)e * m.P + i] = s.vel[i]; st.tar[(size_t)e * m.P + i] = s.tar[i]; }
        if (l < m.D) st.tau[(size_t)e * m.D + l] = s.tau[l];
        if (l < 8) st.kin[(size_t)e * 8 + l] = s.kin[l];
        if (l < 6) st.clock[(size_t)e * 6 + l] = s.clk[l];
        if (l < 4) st.flag[(size_t)e * 4 + l] = s.flg[l];
    }

    // ------------------------------------------------------------------ kinematics (level-synchronous over tree depth)
    // Fills R,p (joint frames), com, Rb (body frames), w (link angular velocity), vj (velocity of the joint origin)
    // and, for the zero-qddot Newton-Euler pass, al (angular acceleration) and aj (acceleration of the joint origin)
    // with base acceleration a0 (gravity enters as a0 = -g).
    DM_DEV void kinematics(const Real* pose, const Real* vel, v3 a0) {
        int par = -1, dep = -1, jt = 0, off = 0;
        if (l < m.J) { par = m.parent[l]; dep = m.depth[l]; jt = m.jtype[l]; off = m.pose_off[l]; }
        for (int d = 0; d <= m.max_depth; ++d) {
            if (l < m.J && dep == d) {
                m3 Rj; v3 pj, w, vj, al, aj;
                if (par < 0) {
                    Rj = quat_to_rot(ldq(pose + 3)); pj = ld3(pose);
                    w = ld3(vel + 3); vj = ld3(vel); al = mk3((Real)0, (Real)0, (Real)0); aj = a0;
                } else {
                    m3 Rp = ldm3(s.R[par]);
                    v3 r = Rp * ld3(m.attach + l * 3);
                    pj = ld3(s.p[par]) + r;
                    m3 Rpa = m.arot_ident[l] ? Rp : Rp * ldm3(m.attach_rot + l * 9);
                    v3 wl = mk3((Real)0, (Real)0, (Real)0);
                    if (jt == JT_SPHERICAL) { Rj = Rpa * quat_to_rot(ldq(pose + off)); wl = ld3(vel + off); }
                    else if (jt == JT_REVOLUTE) { Rj = Rpa * rot_z(pose[off]); wl.z = vel[off]; }
                    else Rj = Rpa;
                    v3 wp = ld3(s.w[par]), alp = ld3(s.al[par]);
                    v3 wrel = Rj * wl;
                    w = wp + wrel;
                    vj = ld3(s.vj[par]) + cross(wp, r);
                    al = alp + cross(wp, wrel);
                    aj = ld3(s.aj[par]) + cross(alp, r) + cross(wp, cross(wp, r));
                }
                stm3(s.R[l], Rj); st3(s.p[l], pj); st3(s.w[l], w); st3(s.vj[l], vj); st3(s.al[l], al); st3(s.aj[l], aj);
                st3(s.com[l], pj + Rj * ld3(m.battach + l * 3));
                stm3(s.Rb[l], m.brot_ident[l] ? Rj : Rj * ldm3(m.brot + l * 9));
            }
            sync();
        }
    }

    // ------------------------------------------------------------------ mass matrix H and bias force C
    // iset: 0 = SPD inertias, 1 = simulator inertias.  Requires kinematics() for the same state.
    DM_DEV void dynamics(int iset) {
        const int J = m.J, D = m.D;
        if (l < J) {                       // per-link force / moment about the COM and world inertia about the COM
            m3 Rb = ldm3(s.Rb[l]);
            const Real* Id = m.inertia + ((size_t)iset * J + l) * 3;
            Real I0 = Id[0], I1 = Id[1], I2 = Id[2];
            Real Iw[6];                    // xx xy xz yy yz zz
            Iw[0] = Rb.m[0] * Rb.m[0] * I0 + Rb.m[1] * Rb.m[1] * I1 + Rb.m[2] * Rb.m[2] * I2;
            Iw[1] = Rb.m[0] * Rb.m[3] * I0 + Rb.m[1] * Rb.m[4] * I1 + Rb.m[2] * Rb.m[5] * I2;
            Iw[2] = Rb.m[0] * Rb.m[6] * I0 + Rb.m[1] * Rb.m[7] * I1 + Rb.m[2] * Rb.m[8] * I2;
            Iw[3] = Rb.m[3] * Rb.m[3] * I0 + Rb.m[4] * Rb.m[4] * I1 + Rb.m[5] * Rb.m[5] * I2;
            Iw[4] = Rb.m[3] * Rb.m[6] * I0 + Rb.m[4] * Rb.m[7] * I1 + Rb.m[5] * Rb.m[8] * I2;
            Iw[5] = Rb.m[6] * Rb.m[6] * I0 + Rb.m[7] * Rb.m[7] * I1 + Rb.m[8] * Rb.m[8] * I2;
            for (int k = 0; k < 6; ++k) s.Iw[l][k] = Iw[k];
            v3 w = ld3(s.w[l]), al = ld3(s.al[l]);
            v3 rc = ld3(s.com[l]) - ld3(s.p[l]);
            v3 ac = ld3(s.aj[l]) + cross(al, rc) + cross(w, cross(w, rc));
            st3(s.f[l], m.mass[l] * ac);
            v3 Iwv = mk3(Iw[0] * w.x + Iw[1] * w.y + Iw[2] * w.z, Iw[1] * w.x + Iw[3] * w.y + Iw[4] * w.z, Iw[2] * w.x + Iw[4] * w.y + Iw[5] * w.z);
            v3 Ial = mk3(Iw[0] * al.x + Iw[1] * al.y + Iw[2] * al.z, Iw[1] * al.x + Iw[3] * al.y + Iw[4] * al.z, Iw[2] * al.x + Iw[4] * al.y + Iw[5] * al.z);
            st3(s.n[l], Ial + cross(w, Iwv));
        }
        if (l < D) {                       // world axis of every generalized velocity
            int j = m.dof_joint[l], kind = m.dof_kind[l], ax = m.dof_axis[l];
            v3 a;
            if (kind == DK_ROOT_LIN || kind == DK_ROOT_ANG) a = mk3((Real)(ax == 0), (Real)(ax == 1), (Real)(ax == 2));
            else a = col(ldm3(s.R[j]), (kind == DK_REV) ? 2 : ax);
            st3(s.axis[l], a);
        }
        sync();
        if (l < J) {                       // subtree sums about this joint's origin (descendants have larger ids)
            uint32_t mask = m.subtree_mask[l];
            v3 pj = ld3(s.p[l]);
            v3 Fs = mk3((Real)0, (Real)0, (Real)0), Ns = Fs, h = Fs;
            Real mc = 0, Ic[6] = { 0, 0, 0, 0, 0, 0 };
            for (int k = l; k < J; ++k) {
                if (!((mask >> k) & 1u)) continue;
                v3 d = ld3(s.com[k]) - pj, fk = ld3(s.f[k]);
                Fs = Fs + fk; Ns = Ns + ld3(s.n[k]) + cross(d, fk);
                Real mk = m.mass[k], dd = dot(d, d);
                mc += mk; h = h + mk * d;
                Ic[0] += s.Iw[k][0] + mk * (dd - d.x * d.x); Ic[1] += s.Iw[k][1] - mk * d.x * d.y; Ic[2] += s.Iw[k][2] - mk * d.x * d.z;
                Ic[3] += s.Iw[k][3] + mk * (dd - d.y * d.y); Ic[4] += s.Iw[k][4] - mk * d.y * d.z; Ic[5] += s.Iw[k][5] + mk * (dd - d.z * d.z);
            }
            st3(s.Fs[l], Fs); st3(s.Ns[l], Ns);
            s.Ic[l][0] = mc; s.Ic[l][1] = h.x; s.Ic[l][2] = h.y; s.Ic[l][3] = h.z;
            for (int k = 0; k < 6; ++k) s.Ic[l][4 + k] = Ic[k];
        }
        sync();
        if (l < D) {
            int j = m.dof_joint[l], kind = m.dof_kind[l], ax = m.dof_axis[l];
            v3 a = ld3(s.axis[l]);
            s.bias[l] = (kind == DK_ROOT_LIN) ? s.Fs[0][ax] : dot(a, ld3(s.Ns[j]));
            // momentum of the composite body of joint j under unit velocity of this dof, about p_j
            const Real* ic = s.Ic[j];
            v3 h = mk3(ic[1], ic[2], ic[3]), Pm, Lp;
            if (kind == DK_ROOT_LIN) { Pm = ic[0] * a; Lp = cross(h, a); }
            else {
                Pm = cross(a, h);
                Lp = mk3(ic[4] * a.x + ic[5] * a.y + ic[6] * a.z, ic[5] * a.x + ic[7] * a.y + ic[8] * a.z, ic[6] * a.x + ic[8] * a.y + ic[9] * a.z);
            }
            v3 pj = ld3(s.p[j]);
            uint64_t anc = m.dof_anc[l];
            for (int k = 0; k <= l; ++k) {
                Real val = 0;
                if ((anc >> k) & 1ull) {
                    int jk = m.dof_joint[k];
                    v3 ak = ld3(s.axis[k]);
                    if (m.dof_kind[k] == DK_ROOT_LIN) val = dot(ak, Pm);
                    else val = dot(ak, Lp + cross(pj - ld3(s.p[jk]), Pm));
                }
                s.H[l][k] = val; s.H[k][l] = val;
            }
        }
        sync();
    }

    // ------------------------------------------------------------------ dense SPD linear algebra, register resident
    // Lane i owns row i of H.  The factorisation and the triangular solves run entirely in VGPRs with
    // v_readlane broadcasts (no LDS round trips, no barriers); the factor L and 1/diag(L) are written back to LDS
    // for the per-row forward substitutions of the constraint solve.  All loops are fully unrolled (ND is the
    // compile-time dof count of the kernel class), so every register-array index is static.
    //
    // Factor s.H = L L^T; when do_solve, also x := H^-1 x for the LDS vector x (length D).
    DM_DEV void chol_solve(Real* xvec, bool do_solve) {
        const int D = m.D;
        Real h[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) h[k] = (l < D && k < D) ? s.H[l < ND ? l : 0][k] : ((l == k) ? (Real)1 : (Real)0);
        Real dinv = 1;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            Real piv = lane_bcast(h[k], k);
            Real inv = (Real)1 / dm_sqrt(piv);
            Real lik = h[k] * inv;
            h[k] = lik;
            if (l == k) dinv = inv;
#pragma unroll
            for (int j = k + 1; j < ND; ++j) h[j] -= lik * lane_bcast(lik, j);
        }
        if (l < ND) {
#pragma unroll
            for (int k = 0; k < ND; ++k) s.H[l][k] = h[k];
            s.dinv[l] = dinv;
        }
        Real x = 0;
        if (do_solve) {
            x = (l < D) ? xvec[l] : (Real)0;
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                Real t = x * dinv; Real xk = lane_bcast(t, k);
                if (l == k) x = t; else if (l > k) x -= h[k] * xk;
            }
        }
        sync();
        if (do_solve) { x = back_substitute(x, dinv); if (l < D) xvec[l] = x; sync(); }
    }
    // x_l := (L^-T x)_l with column l of L read back from LDS
    DM_DEV Real back_substitute(Real x, Real dinv) {
        Real c[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) c[k] = (l < ND && k >= l) ? s.H[k][l < ND ? l : 0] : (Real)0;
#pragma unroll
        for (int k = ND - 1; k >= 0; --k) {
            Real t = x * dinv; Real xk = lane_bcast(t, k);
            if (l == k) x = t; else if (l < k) x -= c[k] * xk;
        }
        return x;
    }
    // x := L^-T x for an LDS vector, using the factor stored by chol_solve
    DM_DEV void solve_upper(Real* xvec) {
        Real x = (l < m.D) ? xvec[l] : (Real)0;
        Real dinv = (l < ND) ? s.dinv[l] : (Real)1;
        x = back_substitute(x, dinv);
        if (l < m.D) xvec[l] = x;
        sync();
    }

    DM_DEV v3 gravity_a0() const { return mk3(-m.gravity[0], -m.gravity[1], -m.gravity[2]); }

    // ------------------------------------------------------------------ stable-PD torques
    DM_DEV void spd(Real dt) {
        const int D = m.D;
        // reference BuildCjRoot quirk == uniform extra base acceleration v0 x (E w - w)  
        v3 v0 = ld3(s.vel), w0 = ld3(s.vel + 3);
        m3 E = quat_to_rot(ldq(s.pose + 3));
        v3 a0 = gravity_a0() + cross(v0, E * w0 - w0);
        mark(0);
        kinematics(s.pose, s.vel, a0);
        mark(1);
        dynamics(0);
        mark(2);
        if (l < m.J && l > 0) {            // pose error per joint -> rhs = Kp e + Kd (0 - qd)
            int jt = m.jtype[l], off = m.pose_off[l], dof = m.dof_off[l];
            if (jt == JT_SPHERICAL) {
                q4 q = ldq(s.pose + off); v3 om = ld3(s.vel + off);
                q4 dq = quat_diff_mul(q, om);
                q4 qh = qnormalize(mkq(q.w + dt * dq.w, q.x + dt * dq.x, q.y + dt * dq.y, q.z + dt * dq.z));
                v3 e = quat_to_rotvec(qmul(qconj(qh), ldq(s.tar + off)), (Real)0.000001);
                for (int k = 0; k < 3; ++k) s.xs[dof + k] = m.kp[dof + k] * comp(e, k) + m.kd[dof + k] * (-s.vel[off + k]);
            } else if (jt == JT_REVOLUTE) {
                Real th = normalize_angle(s.pose[off]);
                Real e = s.tar[off] - (th + dt * s.vel[off]);
                s.xs[dof] = m.kp[dof] * e + m.kd[dof] * (-s.vel[off]);
            }
        }
        if (l < 6) s.xs[l] = 0;
        sync();
        if (l < D) { s.rhs[l] = s.xs[l] - s.bias[l]; s.H[l][l] += dt * m.kd[l]; }
        sync();
        mark(4);
        chol_solve(s.rhs, true);           // rhs = qddot
        mark(3);
        if (l < D) s.tau[l] = (l < 6) ? (Real)0 : s.xs[l] - m.kd[l] * dt * s.rhs[l];
        sync();
        if (l < m.J && l > 0) {            // clamp the torque norm per joint (SimBodyJoint.cpp:299-307)
            int jt = m.jtype[l], dof = m.dof_off[l]; Real lim = m.torque_lim[l];
            if (jt == JT_SPHERICAL) {
                Real mag = dm_sqrt(s.tau[dof] * s.tau[dof] + s.tau[dof + 1] * s.tau[dof + 1] + s.tau[dof + 2] * s.tau[dof + 2]);
                if (mag > lim) { Real k = lim / mag; s.tau[dof] *= k; s.tau[dof + 1] *= k; s.tau[dof + 2] *= k; }
            } else if (jt == JT_REVOLUTE) {
                Real mag = dm_abs(s.tau[dof]);
                if (mag > lim) s.tau[dof] *= lim / mag;
            }
        }
        sync();
    }

    // ------------------------------------------------------------------ rigid-body substep (DM-physics v1)
    DM_DEV Real clamp_vel(Real v, int dof) const {
        Real mx = (dof < 3) ? m.max_lin_vel : m.max_ang_vel;
        return dm_max(-mx, dm_min(mx, v));
    }
    // J row of direction d at world point x on `link`, written into column r of Y; returns J . vstar
    DM_DEV Real build_point_row(int r, int link, v3 x, v3 d) {
        const int D = m.D;
        for (int k = 0; k < D; ++k) Y(k)[r] = 0;
        Y(0)[r] = d.x; Y(1)[r] = d.y; Y(2)[r] = d.z;
        v3 mm = cross(x - ld3(s.p[0]), d);
        Y(3)[r] = mm.x; Y(4)[r] = mm.y; Y(5)[r] = mm.z;
        for (int j = link; j > 0; j = m.parent[j]) {
            int jt = m.jtype[j], dof = m.dof_off[j];
            if (jt == JT_FIXED) continue;
            v3 mj = cross(x - ld3(s.p[j]), d);
            m3 Rj = ldm3(s.R[j]);
            if (jt == JT_SPHERICAL) { v3 t = tmul(Rj, mj); Y(dof)[r] = t.x; Y(dof + 1)[r] = t.y; Y(dof + 2)[r] = t.z; }
            else if (jt == JT_REVOLUTE) Y(dof)[r] = dot(col(Rj, 2), mj);
        }
        Real c = 0;
        for (int k = 0; k < D; ++k) c += Y(k)[r] * s.vstar[k];
        return c;
    }

    DM_DEV void substep(Real h, DebugTaps<Real> dbg, int e) {
        const int D = m.D, J = m.J;
        mark(4);
        kinematics(s.pose, s.vel, gravity_a0());
        mark(5);
        dynamics(1);
        mark(6);
        if (dbg.H) { for (int i = l; i < D * D; i += kWave) dbg.H[(size_t)e * D * D + i] = s.H[i / D][i % D]; if (l < D) dbg.C[(size_t)e * D + l] = s.bias[l]; }
        if (l < D) s.rhs[l] = s.tau[l] - s.bias[l];
        sync();
        chol_solve(s.rhs, true);
        if (l < D) { s.qd[l] = s.vel[m.dof_vidx[l]]; s.vstar[l] = clamp_vel(s.qd[l] + h * s.rhs[l], l); }
        if (l == 0) s.flg[FLG_CONTACT] = 0;
        sync();
        if (dbg.vstar && l < D) dbg.vstar[(size_t)e * D + l] = s.vstar[l];

        mark(7);
        // ---- collision detection: lane = candidate point (CPL candidates per lane when NC > 64)
        constexpr int CPL = NCAP / kWave;
        bool active[CPL]; Real dist[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = l + kWave * q;
            active[q] = false; dist[q] = 0;
            if (c < m.NC) {
                int link = m.cand_link[c];
                v3 x = ld3(s.com[link]) + ldm3(s.Rb[link]) * ld3(m.cand_loc + c * 3);
                x.y -= m.cand_rad[c];
                dist[q] = x.y;
                st3(s.cx[c], x); s.cdist[c] = x.y;
                active[q] = x.y < m.thresh[link];
                if (x.y <= m.report_dist) dm_atomic_or(&s.flg[FLG_CONTACT], 1 << link);
            }
            s.csel[c] = active[q] ? 1 : 0;
        }
        sync();
        // manifold reduction: keep the max_contacts deepest, ties to the lower index; compact in index order
        int rank[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = l + kWave * q; rank[q] = 0;
            if (active[q]) for (int k = 0; k < m.NC; ++k) if (s.csel[k] && (s.cdist[k] < dist[q] || (s.cdist[k] == dist[q] && k < c))) ++rank[q];
        }
        sync();
#pragma unroll
        for (int q = 0; q < CPL; ++q) { active[q] = active[q] && rank[q] < m.max_contacts; s.csel[l + kWave * q] = active[q] ? 1 : 0; }
        sync();
        int nc = 0;
        {
            int slot[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) slot[q] = 0;
            for (int k = 0; k < m.NC; ++k) {
                int v = s.csel[k]; nc += v;
#pragma unroll
                for (int q = 0; q < CPL; ++q) if (k < l + kWave * q) slot[q] += v;
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) if (active[q]) s.cslot[slot[q]] = l + kWave * q;
        }
        const int NL = m.NL;
        const int R = NL + 3 * nc;
        if (l == 0) { s.flg[FLG_NROWS] = R; s.flg[FLG_NCONT] = nc; }
        sync();

        mark(8);
        // ---- constraint rows: lane = row.  limits | normals | frictions (2 per contact)
        Real b = 0, cvec = 0;
        if (l < R) {
            if (l < NL) {
                int j = m.lim_joint[l], off = m.pose_off[j], dof = m.dof_off[j];
                Real th = s.pose[off], pen_lo = th - m.lim_lo[j], pen_hi = m.lim_hi[j] - th;
                Real sgn, pen;
                if (pen_lo <= pen_hi) { sgn = 1; pen = pen_lo; } else { sgn = -1; pen = pen_hi; }
                for (int k = 0; k < D; ++k) Y(k)[l] = 0;
                Y(dof)[l] = sgn;
                cvec = sgn * s.vstar[dof];
                b = (pen > 0) ? -pen / h : -m.erp * pen / h;
            } else if (l < NL + nc) {
                int c = s.cslot[l - NL]; int lk = m.cand_link[c];
                cvec = build_point_row(l, lk, ld3(s.cx[c]), mk3((Real)0, (Real)1, (Real)0));
                Real dd = s.cdist[c];
                b = (dd > 0) ? -dd / h : -m.erp * dd / h;
            } else {
                int fi = l - NL - nc; int c = s.cslot[fi >> 1]; int lk = m.cand_link[c];
                v3 t = (fi & 1) ? mk3((Real)0, (Real)0, (Real)1) : mk3((Real)-1, (Real)0, (Real)0);   // btPlaneSpace1((0,1,0))
                cvec = build_point_row(l, lk, ld3(s.cx[c]), t);
                b = 0;
            }
        }
        sync();
        // y := L^-1 J_l^T in registers (static indices; L rows are wave-uniform LDS broadcasts)
        Real y[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) y[k] = (l < R && k < D) ? Y(k)[l] : (Real)0;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            Real acc = y[k];
#pragma unroll
            for (int q = 0; q < k; ++q) acc -= s.H[k][q] * y[q];
            y[k] = acc * s.dinv[k];
        }
        mark(9);
        // A = Y^T Y: lane l keeps row l of A in a register file indexed by the (wave-uniform) row id
        RowFile<Real> arow;
        Real adiag = 0;
#pragma unroll
        for (int k = 0; k < ND; ++k) adiag += y[k] * y[k];
        for (int r = 0; r < R; ++r) {
            Real acc = 0;
#pragma unroll
            for (int k = 0; k < ND; ++k) acc += y[k] * lane_bcast(y[k], r);
            arow.set(r, acc);
        }
        mark(10);
        // projected Gauss-Seidel in impulse space; u = J v* + A lambda is kept per lane
        Real lam = 0, u = cvec;
        const Real inv_adiag = (l < R) ? (Real)1 / adiag : (Real)0;
        for (int it = 0; it < m.solver_iters; ++it) {
            for (int r = 0; r < R; ++r) {
                Real lo = 0, hi = (Real)1e30;
                if (r >= NL + nc) {        // friction row: bounded by the current normal impulse of its contact
                    Real lam_n = lane_bcast(lam, NL + ((r - NL - nc) >> 1));
                    hi = m.friction * lam_n; lo = -hi;
                }
                Real nl = lam + (b - u) * inv_adiag;
                nl = dm_max(lo, dm_min(hi, nl));
                Real delta = lane_bcast(nl - lam, r);
                u += arow.get(r) * delta;
                if (l == r) lam = nl;
            }
        }
        // hand y back to LDS (k-major) for the lane-per-dof accumulation of Y lambda
        if (l < R) {
#pragma unroll
            for (int k = 0; k < ND; ++k) if (k < D) Y(k)[l] = y[k];
        }
        s.lam[l] = (l < R) ? lam : (Real)0;
        sync();
        mark(11);
        if (dbg.lambda) { dbg.lambda[(size_t)e * kMaxRows + l] = s.lam[l]; if (l < 2) dbg.rows[(size_t)e * 2 + l] = s.flg[FLG_NROWS + l]; }
        // delta v = L^-T (Y lambda)
        if (l < D) { Real z = 0; for (int r = 0; r < R; ++r) z += Y(l)[r] * s.lam[r]; s.xs[l] = z; }
        sync();
        solve_upper(s.xs);
        if (l < D) { s.qd[l] = clamp_vel(s.vstar[l] + s.xs[l], l); s.vel[m.dof_vidx[l]] = s.qd[l]; }
        sync();
        // ---- integrate positions (semi-implicit Euler, exponential map on rotations)
        if (l < J) {
            int jt = m.jtype[l], off = m.pose_off[l];
            if (l == 0) {
                for (int k = 0; k < 3; ++k) s.pose[k] += h * s.vel[k];
                q4 q = qnormalize(qmul(quat_exp(h * ld3(s.vel + 3)), ldq(s.pose + 3)));
                stq(s.pose + 3, q);
            } else if (jt == JT_SPHERICAL) {
                q4 q = qnormalize(qmul(ldq(s.pose + off), quat_exp(h * ld3(s.vel + off))));
                stq(s.pose + off, qstandardize(q));
            } else if (jt == JT_REVOLUTE) s.pose[off] += h * s.vel[off];
        }
        sync();
        mark(12);
    }

    // ------------------------------------------------------------------ reference motion
    // index / blend of cMotion::CalcIndexBlend, uniform across the wave
    DM_DEV void kin_index_blend(double time, int& idx, double& blend, int& cycle) const {
        double dur = m.duration;
        int cc = (int)floor(time / dur);
        cycle = m.loop ? cc : (cc < 0 ? 0 : (cc > 1 ? 1 : cc));
        if (!m.loop) {
            if (time <= 0) { idx = 0; blend = 0; return; }
            if (time >= dur) { idx = m.F - 2; blend = 1; return; }
        }
        time -= cycle * dur;
        int lo = 0, hi = m.F;              // upper_bound
        while (lo < hi) { int mid = (lo + hi) >> 1; if (m.frame_time[mid] <= time) lo = mid + 1; else hi = mid; }
        idx = lo - 1;
        double t0 = m.frame_time[idx], t1 = m.frame_time[idx + 1];
        blend = (time - t0) / (t1 - t0);
    }
    DM_DEV double kin_phase(double time) const {
        double ph = time / m.duration;
        if (m.loop) ph -= floor(ph); else ph = ph < 0 ? 0 : (ph > 1 ? 1 : ph);
        return ph;
    }
    // root position of the kin character at `time` (before the phase-wrap sync), every lane
    DM_DEV v3 kin_root_pos(double time) const {
        int idx, cyc; double blend; kin_index_blend(time, idx, blend, cyc);
        Real b = (Real)(blend < 0 ? 0 : (blend > 1 ? 1 : blend));
        const Real* f0 = m.frames + (size_t)idx * m.P; const Real* f1 = f0 + m.P;
        v3 rp = ((Real)1 - b) * ld3(f0) + b * ld3(f1);
        if (m.loop) rp = rp + (Real)cyc * mk3(m.cycle_delta[0], m.cycle_delta[1], m.cycle_delta[2]);
        return qrot(ldq(s.kin + 3), rp) + ld3(s.kin);
    }
    // full kin pose / vel at `time` into LDS arrays kp, kv (cKinCharacter::CalcPose / CalcVel)
    DM_DEV void kin_sample(double time, Real* kp, Real* kv) {
        int idx, cyc; double blend; kin_index_blend(time, idx, blend, cyc);
        Real b = (Real)(blend < 0 ? 0 : (blend > 1 ? 1 : blend));
        const Real* f0 = m.frames + (size_t)idx * m.P; const Real* f1 = f0 + m.P;
        q4 orot = ldq(s.kin + 3);
        if (l < m.J) {
            int jt = m.jtype[l], off = m.pose_off[l];
            if (l == 0) {
                v3 rp = ((Real)1 - b) * ld3(f0) + b * ld3(f1);
                q4 rr = qnormalize(qslerp(ldq(f0 + 3), b, ldq(f1 + 3), m.slerp_one));
                if (m.loop) rp = rp + (Real)cyc * mk3(m.cycle_delta[0], m.cycle_delta[1], m.cycle_delta[2]);
                rr = qstandardize(qmul(orot, rr));
                rp = qrot(orot, rp) + ld3(s.kin);
                st3(kp, rp); stq(kp + 3, rr);
            } else if (jt == JT_SPHERICAL) stq(kp + off, qslerp(ldq(f0 + off), b, ldq(f1 + off), m.slerp_one));
            else if (jt == JT_REVOLUTE) kp[off] = ((Real)1 - b) * f0[off] + b * f1[off];
        }
        bool over = !m.loop && time >= m.duration;
        const Real* v0 = m.frame_vel + (size_t)idx * m.P; const Real* v1 = v0 + m.P;
        Real bv = (Real)blend;
        for (int i = l; i < m.P; i += kWave) kv[i] = over ? (Real)0 : ((Real)1 - bv) * v0[i] + bv * v1[i];
        sync();
        if (l == 0) {
            v3 v = qrot(orot, ld3(kv)), w = qrot(orot, ld3(kv + 3));
            st3(kv, v); st3(kv + 3, w);
        }
        sync();
    }
    // cSceneImitate::UpdateKinChar: advance the clip clock, snap the origin on phase wrap
    DM_DEV void kin_update(double dt) {
        double t0 = s.clk[CLK_KIN], t1 = t0 + dt;
        double ph0 = kin_phase(t0), ph1 = kin_phase(t1);
        sync();
        if (l == 0) s.clk[CLK_KIN] = t1;
        if (ph1 < ph0 && m.sync_root_pos) {
            v3 kr = kin_root_pos(t1);
            if (l == 0) {
                v3 sp = ld3(s.pose);
                Real dh = kr.y - s.kin[1];
                v3 target = mk3(sp.x, (Real)0 + dh, sp.z);
                v3 delta = target - kr;
                s.kin[0] += delta.x; s.kin[1] += delta.y; s.kin[2] += delta.z;
            }
        }
        sync();
    }

    // ------------------------------------------------------------------ action -> PD targets
    DM_DEV void set_action(const float* a) {
        if (l < m.J && l > 0) {
            int jt = m.jtype[l], off = m.pose_off[l], ao = m.act_off[l];
            if (jt == JT_SPHERICAL) {
                v3 ev = mk3((Real)a[ao], (Real)a[ao + 1], (Real)a[ao + 2]);
                Real len = norm(ev); const Real max_len = (Real)(2 * DM_PI);
                if (len > max_len) ev = (max_len / len) * ev;
                stq(s.tar + off, qnormalize(exp_map_to_quat(ev)));
            } else if (jt == JT_REVOLUTE) s.tar[off] = (Real)a[ao];
        }
        sync();
    }
    // open-loop tracking (stream A1): PD target := reference pose at the current clip time.
    // Uses the same exp-map round trip as the action path so that both paths latch identical targets.
    DM_DEV void set_action_from_clip() {
        Real* kp = s.scratch; Real* kv = s.scratch + NP;
        kin_sample(s.clk[CLK_KIN], kp, kv);
        if (l < m.J && l > 0) {
            int jt = m.jtype[l], off = m.pose_off[l];
            if (jt == JT_SPHERICAL) {
                v3 ev = quat_to_rotvec(ldq(kp + off), (Real)0.000001);
                stq(s.tar + off, qnormalize(exp_map_to_quat(ev)));
            } else if (jt == JT_REVOLUTE) s.tar[off] = kp[off];
        }
        sync();
    }

    // ------------------------------------------------------------------ one scene update (cSceneSimChar::Update)
    DM_DEV void update(double dt, DebugTaps<Real> dbg, int e) {
        if (l == 0) { s.clk[CLK_TIMER] += dt; s.clk[CLK_CTRL] += dt; s.flg[FLG_NEED_ACTION] = 0; }
        kin_update(dt);
        spd((Real)dt);
        Real h = (Real)(dt / m.num_sim_substeps);
        for (int k = 0; k < m.num_sim_substeps; ++k) substep(h, dbg, e);
        if (l == 0) {                      // cCtController::CheckNeedNewAction (CtController.cpp:221-227)
            double cur = s.clk[CLK_CTRL] + s.clk[CLK_INIT_OFF], pad = 0.001 * dt;
            int c1 = (int)floor((cur + pad) / m.query_period), c0 = (int)floor((cur + pad - dt) / m.query_period);
            s.flg[FLG_NEED_ACTION] = (c1 != c0) ? 1 : 0;
        }
        sync();
    }

    // ------------------------------------------------------------------ termination
    DM_DEV bool has_fallen(const Real* kp) const {
        bool f = false;
        if (m.enable_contact_fall) { int cm = s.flg[FLG_CONTACT]; for (int j = 0; j < m.J; ++j) if (m.fall[j] && ((cm >> j) & 1)) f = true; }
        if (m.enable_root_rot_fail && kp) f = f || (quat_theta(qmul(ldq(kp + 3), qconj(ldq(s.pose + 3)))) > (Real)(0.5 * DM_PI));
        return f;
    }

    // ------------------------------------------------------------------ reward + observation + flags
    DM_DEV void emit(const StepIO<Real>& io, DebugTaps<Real> dbg, int e) {
        const int J = m.J;
        Real* kp = s.scratch; Real* kv = s.scratch + NP;
        Real* ee_k = s.scratch + 2 * NP;                 // J x 3 kin joint positions
        Real* red = s.scratch + 2 * NP + 3 * NJ;         // J x 4 per-joint reduction terms
        const v3 zero = mk3((Real)0, (Real)0, (Real)0);
        kin_sample(s.clk[CLK_KIN], kp, kv);
        if (dbg.kin_pose) for (int i = l; i < m.P; i += kWave) { dbg.kin_pose[(size_t)e * m.P + i] = kp[i]; dbg.kin_vel[(size_t)e * m.P + i] = kv[i]; }
        // kin character: joint positions and COM velocity (cRBDUtil::CalcCoM)
        kinematics(kp, kv, zero);
        if (l < J) {
            st3(ee_k + l * 3, ld3(s.p[l]));
            v3 vc = ld3(s.vj[l]) + cross(ld3(s.w[l]), ld3(s.com[l]) - ld3(s.p[l]));
            st3(red + l * 4, m.mass[l] * vc);
        }
        sync();
        if (l == 0) {
            v3 acc = zero; Real tm = 0;
            for (int j = 0; j < J; ++j) { acc = acc + ld3(red + j * 4); tm += m.mass[j]; }
            st3(s.sc + 0, ((Real)1 / tm) * acc);
        }
        sync();
        // sim character
        kinematics(s.pose, s.vel, zero);
        v3 vcom = zero;
        if (l < J) {
            vcom = ld3(s.vj[l]) + cross(ld3(s.w[l]), ld3(s.com[l]) - ld3(s.p[l]));
            st3(s.f[l], vcom);                              // reuse f[] as link COM velocity
            st3(red + l * 4, m.mass[l] * vcom);
        }
        sync();
        if (l == 0) {
            v3 acc = zero; Real tm = 0;
            for (int j = 0; j < J; ++j) { acc = acc + ld3(red + j * 4); tm += m.mass[j]; }
            st3(s.sc + 3, ((Real)1 / tm) * acc);
        }
        sync();
        if (dbg.links && l < J) {
            Real* o = dbg.links + ((size_t)e * J + l) * 21;
            st3(o, ld3(s.com[l])); for (int k = 0; k < 9; ++k) o[3 + k] = s.Rb[l][k];
            st3(o + 12, vcom); st3(o + 15, ld3(s.w[l])); st3(o + 18, ld3(s.p[l]));
        }
        // origin frames (cKinTree::BuildOriginTrans): rotation about y by -heading, translation by -root(x,z)
        Real head0 = calc_heading(ldq(s.pose + 3)), head1 = calc_heading(ldq(kp + 3));
        m3 O0 = rot_y(-head0), O1 = rot_y(-head1);
        // per-joint reward terms
        if (l < J) {
            int jt = m.jtype[l], off = m.pose_off[l];
            Real pe = 0, ve = 0, ee = 0;
            if (l == 0) {
                Real th = quat_theta(qmul(ldq(kp + 3), qconj(ldq(s.pose + 3))));
                pe = th * th;
                v3 dw = ld3(kv + 3) - ld3(s.vel + 3); ve = dot(dw, dw);
            } else if (jt == JT_SPHERICAL) {
                Real th = quat_theta(qmul(ldq(kp + off), qconj(ldq(s.pose + off))));
                pe = th * th;
                v3 dw = ld3(kv + off) - ld3(s.vel + off); ve = dot(dw, dw);
            } else if (jt == JT_REVOLUTE) {
                Real d = kp[off] - normalize_angle(s.pose[off]); pe = d * d;
                Real dv = kv[off] - s.vel[off]; ve = dv * dv;
            }
            if (m.is_ee[l]) {
                v3 p0 = ld3(s.p[l]), p1 = ld3(ee_k + l * 3);
                v3 rel0 = p0 - ld3(s.pose), rel1 = p1 - ld3(kp);
                rel0.y = p0.y - (Real)0; rel1.y = p1.y - s.kin[1];
                v3 dlt = O1 * rel1 - O0 * rel0;
                ee = dot(dlt, dlt);
            }
            red[l * 4 + 0] = m.diffw[l] * pe; red[l * 4 + 1] = m.diffw[l] * ve; red[l * 4 + 2] = ee;
        }
        sync();
        if (l == 0) {
            Real pose_err = 0, vel_err = 0, ee_err = 0;
            for (int j = 0; j < J; ++j) { pose_err += red[j * 4]; vel_err += red[j * 4 + 1]; ee_err += red[j * 4 + 2]; }
            v3 rp0 = ld3(s.pose), rp1 = ld3(kp); rp1.y -= s.kin[1];
            v3 dp = rp0 - rp1;
            Real th = quat_theta(qmul(ldq(kp + 3), qconj(ldq(s.pose + 3))));
            v3 dv = ld3(kv) - ld3(s.vel), dw = ld3(kv + 3) - ld3(s.vel + 3);
            Real root_err = dot(dp, dp) + (Real)0.1 * th * th + (Real)0.01 * dot(dv, dv) + (Real)0.001 * dot(dw, dw);
            v3 dc = ld3(s.sc + 0) - ld3(s.sc + 3);
            Real com_err = (Real)0.1 * dot(dc, dc);
            const Real pose_scale = (Real)2.0 / 15 * J, vel_scale = (Real)0.1 / 15 * J;
            Real r = (Real)0.5 * dm_exp(-pose_scale * pose_err) + (Real)0.05 * dm_exp(-vel_scale * vel_err)
                   + (Real)0.15 * dm_exp(-(Real)10 * ee_err) + (Real)0.2 * dm_exp(-(Real)5 * root_err) + (Real)0.1 * dm_exp(-(Real)10 * com_err);
            bool fallen = has_fallen(kp);
            if (fallen) r = 0;
            bool fail = (m.enable_fall_end && fallen) || (!m.loop && s.clk[CLK_KIN] >= m.duration);
            bool end = fail || (s.clk[CLK_TIMER] >= s.clk[CLK_TIMER_MAX]);
            if (io.rewards) io.rewards[e] = (float)r;
            if (io.terminate) io.terminate[e] = fail ? TERM_FAIL : TERM_NULL;
            if (io.episode_end) io.episode_end[e] = end ? 1 : 0;
            s.sc[6] = end ? (Real)1 : (Real)0;
            if (dbg.reward_terms) { Real* o = dbg.reward_terms + (size_t)e * 5; o[0] = pose_err; o[1] = vel_err; o[2] = ee_err; o[3] = root_err; o[4] = com_err; }
        }
        // CheckValidEpisode: any link velocity component beyond 100 (SimCharacter.cpp:571-586)
        if (l == 0) s.flg[FLG_VALID] = 1;
        sync();
        if (l < J) {
            v3 w = ld3(s.w[l]);
            Real mx = dm_max(dm_max(dm_abs(vcom.x), dm_abs(vcom.y)), dm_abs(vcom.z));
            mx = dm_max(mx, dm_max(dm_max(dm_abs(w.x), dm_abs(w.y)), dm_abs(w.z)));
            if (mx > (Real)100) s.flg[FLG_VALID] = 0;
        }
        sync();
        if (l == 0 && io.valid) io.valid[e] = s.flg[FLG_VALID];
        // observation
        if (io.states) {
            float* out = io.states + (size_t)e * m.S;
            int base = 0;
            if (m.enable_phase_input) {
                if (l == 0) { double ph = fmod(s.clk[CLK_CTRL] / m.duration, 1.0); if (ph < 0) ph += 1; out[0] = (float)ph; }
                base = 1;
            }
            v3 rpos = ld3(s.pose);
            if (l == 0) out[base] = (float)rpos.y;          // root height in the origin frame (ground at 0)
            if (l < J) {
                v3 pc = ld3(s.com[l]);
                if (!m.record_world_root_pos || l != 0) { v3 t = mk3(pc.x - rpos.x, pc.y, pc.z - rpos.z); pc = O0 * t; pc.y -= rpos.y; }
                m3 Rb = ldm3(s.Rb[l]);
                if (!m.record_world_root_rot || l != 0) Rb = O0 * Rb;
                v3 nrm = col(Rb, 1), tan = col(Rb, 0);
                float* o = out + base + 1 + 9 * l;
                o[0] = (float)pc.x; o[1] = (float)pc.y; o[2] = (float)pc.z;
                o[3] = (float)nrm.x; o[4] = (float)nrm.y; o[5] = (float)nrm.z; o[6] = (float)tan.x; o[7] = (float)tan.y; o[8] = (float)tan.z;
                v3 v = vcom, w = ld3(s.w[l]);
                if (!m.record_world_root_rot || l != 0) { v = O0 * v; w = O0 * w; }
                float* ov = out + base + 1 + 9 * J + 6 * l;
                ov[0] = (float)v.x; ov[1] = (float)v.y; ov[2] = (float)v.z; ov[3] = (float)w.x; ov[4] = (float)w.y; ov[5] = (float)w.z;
            }
        }
        sync();
    }

    // ------------------------------------------------------------------ reset
    DM_DEV void reset_env(double kin_time, double max_time) {
        Real* kp = s.scratch; Real* kv = s.scratch + NP;
        if (l == 0) {
            s.clk[CLK_TIMER] = 0; s.clk[CLK_TIMER_MAX] = max_time;
            s.clk[CLK_KIN] = kin_time; s.clk[CLK_CTRL] = kin_time; s.clk[CLK_INIT_OFF] = -kin_time;
            s.flg[FLG_NEED_ACTION] = 1; s.flg[FLG_CONTACT] = 0; s.flg[FLG_VALID] = 1; s.flg[FLG_EPISODE] += 1;
            s.kin[0] = s.kin[1] = s.kin[2] = 0; s.kin[3] = 1; s.kin[4] = s.kin[5] = s.kin[6] = 0;
        }
        if (l < m.D) s.tau[l] = 0;
        sync();
        kin_sample(kin_time, kp, kv);
        // sim := kin (cSimCharacter::SetPose/SetVel then BuildPose/BuildVel): unit quaternions, spherical w >= 0
        for (int i = l; i < m.P; i += kWave) { s.pose[i] = kp[i]; s.vel[i] = kv[i]; }
        sync();
        if (l < m.J) {
            int jt = m.jtype[l], off = m.pose_off[l];
            if (l == 0) { stq(s.pose + 3, qnormalize(ldq(s.pose + 3))); s.vel[6] = 0; if (m.enable_rand_placement) { s.pose[0] = 0; s.pose[2] = 0; } }
            else if (jt == JT_SPHERICAL) { stq(s.pose + off, qstandardize(qnormalize(ldq(s.pose + off)))); s.vel[off + 3] = 0; }
        }
        sync();
        // ResolveCharGroundIntersect: lift the root so that every link AABB clears the ground by 1 mm
        kinematics(s.pose, s.vel, mk3((Real)0, (Real)0, (Real)0));
        Real viol = 0;
        if (l < m.J) {
            const Real* he = m.aabb_he + l * 4; m3 Rb = ldm3(s.Rb[l]);
            Real ext = (he[3] != 0) ? he[0] : dm_abs(Rb.m[3]) * he[0] + dm_abs(Rb.m[4]) * he[1] + dm_abs(Rb.m[5]) * he[2];
            viol = dm_min((Real)0, s.com[l][1] - ext - (Real)0.001);
        }
        s.row_b[l] = viol;
        sync();
        if (l == 0) {
            Real mv = 0; for (int j = 0; j < m.J; ++j) mv = dm_min(mv, s.row_b[j]);
            if (mv < 0) s.pose[1] += -mv;
            // SyncKinCharRoot: kin root := sim root (moves the origin)
            for (int k = 0; k < 3; ++k) s.kin[k] += s.pose[k] - kp[k];
        }
        sync();
    }
};

// Counter-based uniform in [0,1): splitmix64 of (seed, env, episode, stream)
DM_HD double dm_rand01(uint64_t seed, uint64_t env, uint64_t episode, uint64_t stream) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (env * 0x100000001B3ull + episode * 0xD6E8FEB86659FD93ull + stream + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// ============================================================================ kernels
// grid = number of envs, block = one wavefront.
template <typename Real, int NJ, int ND, int NP, int NCAP>
__global__ void __launch_bounds__(64) k_env_step(ModelDev<Real> m, EnvState<Real> st, StepIO<Real> io, DebugTaps<Real> dbg) {
    __shared__ Lds<Real, NJ, ND, NP, NCAP> lds;
    const int e = blockIdx.x, l = threadIdx.x;
    EnvSim<Real, NJ, ND, NP, NCAP> sim(m, lds, l);
    if (dbg.prof) { sim.prof = dbg.prof + (size_t)e * 16; sim.tprev = dm_clock(); }
    sim.load(st, e);
    if (io.open_loop) sim.set_action_from_clip();
    else if (io.actions) sim.set_action(io.actions + (size_t)e * m.A);
    sim.mark(15);
    for (int u = 0; u < io.n_updates; ++u) sim.update(io.dt, dbg, e);
    if (io.emit) {
        sim.emit(io, dbg, e);
        const bool ended = lds.sc[6] != (Real)0;
        if (io.auto_reset && ended) {
            // mirrors DeepMimic.py:70-79: the terminal reward / flags were just written; start the next episode and
            // hand back the observation the agent needs for its first action (RecordState after Reset)
            uint64_t ep = (uint64_t)lds.flg[FLG_EPISODE];
            double kt = m.duration * dm_rand01(m.seed, (uint64_t)(e + m.env_off), ep, 0);
            double mt = (m.time_lim_max > m.time_lim_min) ? m.time_lim_min + (m.time_lim_max - m.time_lim_min) * dm_rand01(m.seed, (uint64_t)(e + m.env_off), ep, 1) : m.time_lim_max;
            sim.reset_env(kt, mt);
            StepIO<Real> io2 = io; io2.rewards = nullptr; io2.terminate = nullptr; io2.valid = nullptr; io2.episode_end = nullptr;
            DebugTaps<Real> nodbg = DebugTaps<Real>();
            sim.emit(io2, nodbg, e);
        }
        sim.mark(13);
    }
    sim.store(st, e);
    sim.mark(14);
}

// reset the envs listed in env_ids (or all when env_ids == null); kin_times / max_times optional per listed env
template <typename Real, int NJ, int ND, int NP, int NCAP>
__global__ void __launch_bounds__(64) k_env_reset(ModelDev<Real> m, EnvState<Real> st, const int* env_ids, const double* kin_times, const double* max_times) {
    __shared__ Lds<Real, NJ, ND, NP, NCAP> lds;
    const int b = blockIdx.x, l = threadIdx.x;
    const int e = env_ids ? env_ids[b] : b;
    EnvSim<Real, NJ, ND, NP, NCAP> sim(m, lds, l);
    sim.load(st, e);
    uint64_t ep = (uint64_t)lds.flg[FLG_EPISODE];
    double kt = kin_times ? kin_times[b] : m.duration * dm_rand01(m.seed, (uint64_t)(e + m.env_off), ep, 0);
    double mt = max_times ? max_times[b]
              : ((m.time_lim_max > m.time_lim_min) ? m.time_lim_min + (m.time_lim_max - m.time_lim_min) * dm_rand01(m.seed, (uint64_t)(e + m.env_off), ep, 1) : m.time_lim_max);
    sim.reset_env(kt, mt);
    sim.store(st, e);
}

// observation / reward / flags for the current state without stepping (RecordState, CalcReward, CheckTerminate)
template <typename Real, int NJ, int ND, int NP, int NCAP>
__global__ void __launch_bounds__(64) k_env_query(ModelDev<Real> m, EnvState<Real> st, StepIO<Real> io, DebugTaps<Real> dbg) {
    __shared__ Lds<Real, NJ, ND, NP, NCAP> lds;
    const int e = blockIdx.x, l = threadIdx.x;
    EnvSim<Real, NJ, ND, NP, NCAP> sim(m, lds, l);
    sim.load(st, e);
    sim.emit(io, dbg, e);
}

// component taps for parity tests: SPD torque for the stored state / one substep with the stored torque
template <typename Real, int NJ, int ND, int NP, int NCAP>
__global__ void __launch_bounds__(64) k_env_probe(ModelDev<Real> m, EnvState<Real> st, DebugTaps<Real> dbg, int what, double dt) {
    __shared__ Lds<Real, NJ, ND, NP, NCAP> lds;
    const int e = blockIdx.x, l = threadIdx.x;
    EnvSim<Real, NJ, ND, NP, NCAP> sim(m, lds, l);
    sim.load(st, e);
    if (what == 0) {                       // SPD
        sim.spd((Real)dt);
    } else if (what == 1) {                // substep
        sim.substep((Real)dt, dbg, e);
    } else if (what == 2) {                // dynamics only (SPD model) -> H, C taps
        typename EnvSim<Real, NJ, ND, NP, NCAP>::v3 v0 = ld3(lds.vel), w0 = ld3(lds.vel + 3);
        M3<Real> E = quat_to_rot(ldq(lds.pose + 3));
        sim.kinematics(lds.pose, lds.vel, sim.gravity_a0() + cross(v0, E * w0 - w0));
        sim.dynamics(0);
        if (dbg.H) { for (int i = l; i < m.D * m.D; i += kWave) dbg.H[(size_t)e * m.D * m.D + i] = lds.H[i / m.D][i % m.D]; if (l < m.D) dbg.C[(size_t)e * m.D + l] = lds.bias[l]; }
    }
    sim.store(st, e);
}

}  // namespace dmk
