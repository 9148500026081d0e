// Small fixed-size vector / quaternion helpers for the device code (also compiled by the host
// table builder).  Conventions follow the reference: quaternions are stored (w,x,y,z)
// (DeepMimicCore/util/MathUtil.cpp:512-520); rotation from quaternion as cMathUtil::RotateMat(q)
// (:207-237); angles normalised to [-pi,pi] as cMathUtil::NormalizeAngle (:33-46).
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef DM_HD
#if defined(__HIPCC__)
#define DM_HD __host__ __device__ __forceinline__
#else
#define DM_HD inline
#endif
#endif

namespace dmk {

template <typename T> struct V3 { T x, y, z; };
template <typename T> struct Q4 { T w, x, y, z; };
template <typename T> struct M3 { T m[9]; };   // row-major

#define DM_PI 3.14159265358979323846

template <typename T> DM_HD V3<T> mk3(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename T> DM_HD V3<T> ld3(const T* p) { return mk3(p[0], p[1], p[2]); }
template <typename T> DM_HD void st3(T* p, const V3<T>& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
template <typename T> DM_HD V3<T> operator+(const V3<T>& a, const V3<T>& b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> DM_HD V3<T> operator-(const V3<T>& a, const V3<T>& b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> DM_HD V3<T> operator-(const V3<T>& a) { return mk3(-a.x, -a.y, -a.z); }
template <typename T> DM_HD V3<T> operator*(T s, const V3<T>& a) { return mk3(s * a.x, s * a.y, s * a.z); }
template <typename T> DM_HD T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> DM_HD V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// c + a x b with the additions folded into the products (two FMAs per component instead of mul, fma, add)
template <typename T> DM_HD V3<T> cross_add(const V3<T>& c, const V3<T>& a, const V3<T>& b) {
    return mk3((c.x + a.y * b.z) - a.z * b.y, (c.y + a.z * b.x) - a.x * b.z, (c.z + a.x * b.y) - a.y * b.x);
}
template <typename T> DM_HD T comp(const V3<T>& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

DM_HD float dm_sqrt(float x) { return sqrtf(x); }
DM_HD double dm_sqrt(double x) { return sqrt(x); }
DM_HD float dm_sin(float x) { return sinf(x); }
DM_HD double dm_sin(double x) { return sin(x); }
DM_HD float dm_cos(float x) { return cosf(x); }
DM_HD double dm_cos(double x) { return cos(x); }
DM_HD float dm_acos(float x) { return acosf(x); }
DM_HD double dm_acos(double x) { return acos(x); }
DM_HD float dm_atan2(float y, float x) { return atan2f(y, x); }
DM_HD double dm_atan2(double y, double x) { return atan2(y, x); }
DM_HD float dm_fmod(float x, float y) { return fmodf(x, y); }
DM_HD double dm_fmod(double x, double y) { return fmod(x, y); }
DM_HD float dm_exp(float x) { return expf(x); }
DM_HD double dm_exp(double x) { return exp(x); }
DM_HD float dm_abs(float x) { return fabsf(x); }
DM_HD double dm_abs(double x) { return fabs(x); }
DM_HD float dm_min(float a, float b) { return fminf(a, b); }
DM_HD double dm_min(double a, double b) { return fmin(a, b); }
DM_HD float dm_max(float a, float b) { return fmaxf(a, b); }
DM_HD double dm_max(double a, double b) { return fmax(a, b); }
DM_HD int dm_min(int a, int b) { return a < b ? a : b; }
DM_HD int dm_max(int a, int b) { return a > b ? a : b; }

// sin and cos together.  fp32: Cody-Waite reduction by pi/2 + fdlibm kernel polynomials (|err| < 1 ulp for |x| < 1e4),
// ~25 instructions instead of two libm calls; fp64: libm.
DM_HD void dm_sincos(double x, double& sn, double& cs) { sn = sin(x); cs = cos(x); }
DM_HD void dm_sincos(float x, float& sn, float& cs) {
    const float kf = rintf(x * 0.636619772367581343f);
    const int k = (int)kf;
    float r = fmaf(kf, -1.5707963705e+00f, x); r = fmaf(kf, 4.3711388287e-08f, r); r = fmaf(kf, 1.7763568394e-15f, r);
    const float r2 = r * r;
    const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, fmaf(r2, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f), fmaf(r2, -0.5f, 1.0f));
    const float s0 = (k & 1) ? cp : sp, c0 = (k & 1) ? sp : cp;
    sn = (k & 2) ? -s0 : s0;
    cs = ((k + 1) & 2) ? -c0 : c0;
}

template <typename T> DM_HD T norm(const V3<T>& a) { return dm_sqrt(dot(a, a)); }

template <typename T> DM_HD M3<T> m3_identity() { M3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = (i % 4 == 0) ? (T)1 : (T)0; return r; }
template <typename T> DM_HD M3<T> ldm3(const T* p) { M3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
template <typename T> DM_HD void stm3(T* p, const M3<T>& a) { for (int i = 0; i < 9; ++i) p[i] = a.m[i]; }
template <typename T> DM_HD V3<T> operator*(const M3<T>& a, const V3<T>& v) {
    return mk3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
template <typename T> DM_HD V3<T> tmul(const M3<T>& a, const V3<T>& v) {   // a^T v
    return mk3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z, a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
template <typename T> DM_HD M3<T> operator*(const M3<T>& a, const M3<T>& b) {
    M3<T> r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
    return r;
}
template <typename T> DM_HD V3<T> col(const M3<T>& a, int c) { return mk3(a.m[c], a.m[3 + c], a.m[6 + c]); }

template <typename T> DM_HD T normalize_angle(T theta) {
    const T two_pi = (T)(2 * DM_PI), pi = (T)DM_PI;
    T n = dm_fmod(theta, two_pi);
    if (n > pi) n = -two_pi + n; else if (n < -pi) n = two_pi + n;
    return n;
}
template <typename T> DM_HD Q4<T> mkq(T w, T x, T y, T z) { Q4<T> q; q.w = w; q.x = x; q.y = y; q.z = z; return q; }
template <typename T> DM_HD Q4<T> ldq(const T* p) { return mkq(p[0], p[1], p[2], p[3]); }
template <typename T> DM_HD void stq(T* p, const Q4<T>& q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
template <typename T> DM_HD Q4<T> qmul(const Q4<T>& a, const Q4<T>& b) {
    return mkq(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
template <typename T> DM_HD Q4<T> qconj(const Q4<T>& q) { return mkq(q.w, -q.x, -q.y, -q.z); }
template <typename T> DM_HD T qdot(const Q4<T>& a, const Q4<T>& b) { return a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> DM_HD Q4<T> qnormalize(const Q4<T>& q) {
    T inv = (T)1 / dm_sqrt(qdot(q, q));
    return mkq(q.w * inv, q.x * inv, q.y * inv, q.z * inv);
}
template <typename T> DM_HD Q4<T> qstandardize(const Q4<T>& q) { return (q.w < 0) ? mkq(-q.w, -q.x, -q.y, -q.z) : q; }
// rotate a vector (Eigen q*v)
template <typename T> DM_HD V3<T> qrot(const Q4<T>& q, const V3<T>& v) {
    V3<T> u = mk3(q.x, q.y, q.z);
    V3<T> uv = cross(u, v); uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}
// cMathUtil::RotateMat(quat)
template <typename T> DM_HD M3<T> quat_to_rot(const Q4<T>& q) {
    T sqw = q.w * q.w, sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z;
    T invs = (T)1 / (sqx + sqy + sqz + sqw);
    M3<T> r;
    r.m[0] = (sqx - sqy - sqz + sqw) * invs; r.m[4] = (-sqx + sqy - sqz + sqw) * invs; r.m[8] = (-sqx - sqy + sqz + sqw) * invs;
    T t1 = q.x * q.y, t2 = q.z * q.w;
    r.m[3] = (T)2 * (t1 + t2) * invs; r.m[1] = (T)2 * (t1 - t2) * invs;
    t1 = q.x * q.z; t2 = q.y * q.w;
    r.m[6] = (T)2 * (t1 - t2) * invs; r.m[2] = (T)2 * (t1 + t2) * invs;
    t1 = q.y * q.z; t2 = q.x * q.w;
    r.m[7] = (T)2 * (t1 + t2) * invs; r.m[5] = (T)2 * (t1 - t2) * invs;
    return r;
}
template <typename T> DM_HD M3<T> rot_z(T th) {
    T c, s; dm_sincos(th, s, c);
    M3<T> r = m3_identity<T>(); r.m[0] = c; r.m[1] = -s; r.m[3] = s; r.m[4] = c;
    return r;
}
template <typename T> DM_HD M3<T> rot_y(T th) {
    T c, s; dm_sincos(th, s, c);
    M3<T> r = m3_identity<T>(); r.m[0] = c; r.m[2] = s; r.m[6] = -s; r.m[8] = c;
    return r;
}
// rotation vector (axis * angle) of a unit quaternion, angle normalised to [-pi,pi]; zero below `eps` on
// |sin(theta/2)|.  Same function as cMathUtil::QuaternionToAxisAngle (theta*axis) but evaluated with atan2 so that
// it stays accurate in fp32 for small angles (2*acos(w) loses all precision there).
template <typename T> DM_HD V3<T> quat_to_rotvec(const Q4<T>& q, T eps) {
    T s = dm_sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (!(s > eps)) return mk3((T)0, (T)0, (T)0);
    T th = normalize_angle((T)2 * dm_atan2(s, q.w));
    T k = th / s;
    return mk3(k * q.x, k * q.y, k * q.z);
}
// |rotation angle| semantics of cMathUtil::QuatTheta (threshold 1e-4 on sin(theta/2))
template <typename T> DM_HD T quat_theta(const Q4<T>& q) {
    T s = dm_sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (!(s > (T)0.0001)) return (T)0;
    return normalize_angle((T)2 * dm_atan2(s, q.w));
}
// exact exponential map of a rotation vector
template <typename T> DM_HD Q4<T> quat_exp(const V3<T>& rv) {
    T th = norm(rv), half = (T)0.5 * th;
    T sh, ch; dm_sincos(half, sh, ch);
    T s = (th < (T)1e-6) ? ((T)0.5 - th * th / (T)48) : sh / th;
    return mkq(ch, s * rv.x, s * rv.y, s * rv.z);
}
// cMathUtil::ExpMapToQuaternion (gThetaMin = 1e-6)
template <typename T> DM_HD Q4<T> exp_map_to_quat(const V3<T>& e) {
    T th = norm(e);
    if (!(th > (T)0.000001)) return mkq((T)1, (T)0, (T)0, (T)0);
    T ang = normalize_angle(th), c = dm_cos((T)0.5 * ang), s = dm_sin((T)0.5 * ang) / th;
    return mkq(c, s * e.x, s * e.y, s * e.z);
}
// 0.5 * q (x) (0,omega): cMathUtil::BuildQuaternionDiffMat(q) * omega
template <typename T> DM_HD Q4<T> quat_diff_mul(const Q4<T>& q, const V3<T>& o) {
    const T h = (T)0.5;
    return mkq(-h * q.x * o.x - h * q.y * o.y - h * q.z * o.z, h * q.w * o.x - h * q.z * o.y + h * q.y * o.z,
               h * q.z * o.x + h * q.w * o.y - h * q.x * o.z, -h * q.y * o.x + h * q.x * o.y + h * q.w * o.z);
}
// Eigen Quaternion::slerp
template <typename T> DM_HD Q4<T> qslerp(const Q4<T>& a, T t, const Q4<T>& b, T one_minus_eps) {
    T d = qdot(a, b), ad = dm_abs(d), s0, s1;
    if (ad >= one_minus_eps) { s0 = (T)1 - t; s1 = t; }
    else { T th = dm_acos(ad), st = dm_sin(th); s0 = dm_sin(((T)1 - t) * th) / st; s1 = dm_sin(t * th) / st; }
    if (d < 0) s1 = -s1;
    return mkq(s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z);
}
// heading of a root rotation (cKinTree::CalcHeading)
template <typename T> DM_HD T calc_heading(const Q4<T>& q) {
    V3<T> d = qrot(q, mk3((T)1, (T)0, (T)0));
    return dm_atan2(-d.z, d.x);
}

// Counter-based uniform in [0,1): splitmix64 of (seed, env, episode, stream)  (host code draws expert sample times with it too)
DM_HD double dm_rand01(uint64_t seed, uint64_t env, uint64_t episode, uint64_t stream) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (env * 0x100000001B3ull + episode * 0xD6E8FEB86659FD93ull + stream + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

}  // namespace dmk
