// Device code of the imitate hot path: one wavefront (64 lanes) simulates one character.
//
//   lanes <-> links      for kinematics / Newton-Euler passes (15 or 23 of 64 lanes active)
//   lanes <-> dofs       for mass-matrix rows, Cholesky rows, triangular solves (34 / 64 lanes)
//   lanes <-> contact candidates, then constraint rows, for collision + PGS (<= 64 rows)
//
// All per-env working data lives in LDS / VGPRs for the whole call (20 scene updates = one control step); HBM is
// touched only to load the env record and the model block at entry and to store record + observation + reward at
// exit.  The workgroup is exactly one wavefront, so __syncthreads() is a wave-local LDS fence (no s_barrier).
//
// Register-resident linear algebra (lane i owns row i): the Cholesky factor, the triangular solves, the
// constraint-space vectors y_r = L^-1 J_r^T (lane r owns constraint row r, statically indexed VGPR array), the rows
// of A = Y^T Y and the projected Gauss-Seidel sweep all run in VGPRs with v_readlane / DPP cross-lane traffic.
//
// Reference functions realised here (DeepMimicCore/...):
//   kin_*        anim/Motion.cpp:249-293,476-515; anim/MotionController.cpp:25-47,102-111; anim/KinCharacter.cpp:363-406
//   dynamics()   sim/RBDUtil.cpp:4-97 (RNEA), 123-195 (CRBA) -- same H and C, evaluated as a Newton-Euler pass in
//                world-aligned axes about each joint's own origin (no 6x6 frame transforms; fp32-safe)
//   spd_*()      sim/ImpPDController.cpp:136-195, sim/SimBodyJoint.cpp:299-307,636-695
//   substep_*()  DM-physics v1 standing in for btMultiBodyDynamicsWorld::stepSimulation
//   emit()       scenes/SceneImitate.cpp:7-127,163-205; sim/CtController.cpp:281-478; sim/SimCharacter.cpp:542-586
//   reset_env()  scenes/SceneSimChar.cpp:487-583,628-644; scenes/SceneImitate.cpp:320-368,386-418
#pragma once
#include <type_traits>
#include "dm_math.h"
#include "dm_types.h"

// ============================================================================ wave-level primitives
#ifdef DM_EMU
static inline long long dm_clock() { return 0; }
template <int P> static inline void dm_setprio() {}
template <typename T> static inline T* dm_uniform_ptr(T* p) { return p; }
#define DM_DEV inline
#define DM_OPAQUE_S(x) ((void)0)
#define DM_OPAQUE_V(x) ((void)0)
#define DM_OPAQUE_D(x) ((void)0)
#define DM_SCHED_FENCE() ((void)0)
static inline int dm_atomic_or(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline int dm_atomic_min(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int dm_atomic_add(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline int dm_popc64(uint64_t v) { return __builtin_popcountll(v); }
static inline int dm_ctz32(uint32_t v) { return __builtin_ctz(v); }
namespace dmk {
// every lane of the wave must call these from uniform control flow (the emulator exchanges through a buffer)
template <typename T> static inline T wave_shfl(T v, int src) {
    T* x = reinterpret_cast<T*>(emu::g_xchg);
    x[threadIdx.x] = v; __syncthreads(); T r = x[src & 63]; __syncthreads(); return r;
}
template <typename T> static inline T lane_bcast(T v, int src) { return wave_shfl(v, src); }
static inline uint64_t wave_ballot(bool p) {
    uint64_t* x = emu::g_xchg;
    x[threadIdx.x] = p ? 1 : 0; __syncthreads();
    uint64_t m = 0; for (int i = 0; i < 64; ++i) m |= (x[i] & 1ull) << i;
    __syncthreads(); return m;
}
template <typename T> static inline T wave_sum(T v) {
    T* x = reinterpret_cast<T*>(emu::g_xchg);
    x[threadIdx.x] = v; __syncthreads();
    T s = 0; for (int i = 0; i < 64; ++i) s += x[i];
    __syncthreads(); return s;
}
template <int MASK, typename T> static inline T wave_shfl_xor_c(T v) { return wave_shfl(v, (int)(threadIdx.x ^ MASK)); }
static inline float dm_rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline double dm_rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float dm_rcp(float x) { return 1.0f / x; }
static inline double dm_rcp(double x) { return 1.0 / x; }
template <typename T> static inline T dm_med3(T lo, T x, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
// (bit l of MASK) ? a : b for a compile-time lane mask
template <uint64_t MASK, typename T> static inline T lane_sel(T a, T b, int l, uint64_t) { return ((MASK >> (l & 63)) & 1ull) ? a : b; }
template <int R, typename T> static inline T row_sel_c(T oldv, T newv, int l, uint32_t) { return l == R ? newv : oldv; }
template <typename Real, int N> struct RowFile {       // per-lane array indexed by a wave-uniform runtime index
    Real v[N];
    inline Real get(int r) const { return v[r]; }
    inline void set(int r, Real x) { v[r] = x; }
};
// G[i] = sum_k y_l[k] y_i[k], i < 32, for every lane l (Gram rows of the first 32 constraint rows)
template <int NP2, typename R2, typename Real> static inline void wave_gram32(const R2* y2, Real (&out)[32]) {
    Real* x = reinterpret_cast<Real*>(emu::g_xchg);
    for (int p = 0; p < NP2; ++p) { x[threadIdx.x * 2 * NP2 + 2 * p] = y2[p][0]; x[threadIdx.x * 2 * NP2 + 2 * p + 1] = y2[p][1]; }
    __syncthreads();
    for (int i = 0; i < 32; ++i) { Real a = 0; for (int k = 0; k < 2 * NP2; ++k) a += x[threadIdx.x * 2 * NP2 + k] * x[i * 2 * NP2 + k]; out[i] = a; }
    __syncthreads();
}
// two / four packed reals (GCC vector extension on the host emulator)
// 64-row Gram: generic statement (the emulator build and fp64)
template <int NP2, typename R2, typename Real> static inline void wave_gram64(const R2* y2, Real (&out)[64]) {
    for (int i = 0; i < 64; ++i) { Real a = 0; for (int p = 0; p < NP2; ++p) a += y2[p][0] * lane_bcast(y2[p][0], i) + y2[p][1] * lane_bcast(y2[p][1], i); out[i] = a; }
}
template <int NP2, int H, typename R2, typename Real> static inline void wave_gram64_half(const R2* y2, Real (&out)[32]) {
    for (int i = 0; i < 32; ++i) { Real a = 0; for (int p = 0; p < NP2; ++p) a += y2[p][0] * lane_bcast(y2[p][0], 32 * H + i) + y2[p][1] * lane_bcast(y2[p][1], 32 * H + i); out[i] = a; }
}
template <typename Real> struct VecT;
template <> struct VecT<float> { typedef float v2 __attribute__((vector_size(8))); typedef float v4 __attribute__((vector_size(16))); };
template <> struct VecT<double> { typedef double v2 __attribute__((vector_size(16))); typedef double v4 __attribute__((vector_size(32))); };
}
#else
#define DM_DEV __device__ __forceinline__
// make a wave-uniform (SGPR) / per-lane (VGPR) value opaque to the optimizer: stops loop-invariant hoisting of the
// hundreds of compare masks an unrolled sweep would otherwise keep live (and spill)
#define DM_OPAQUE_S(x) asm volatile("" : "+s"(x))
#define DM_OPAQUE_V(x) asm volatile("" : "+v"(x))
// a double (or pointer) pinned where it stands: keeps the optimizer from hoisting the rare-path f64 arithmetic on scene parameters
// (range widths of the random draws, ...) out of the 20-update loop into VGPR pairs that then live -- and spill -- across the whole kernel
#define DM_OPAQUE_D(x) asm volatile("" : "+v"(x))
// nothing is scheduled across this point (keeps the rank-1 updates of one Cholesky column next to the loads that feed them)
#define DM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ long long dm_clock() { return (long long)__builtin_readcyclecounter(); }
template <int P> __device__ __forceinline__ void dm_setprio() { __builtin_amdgcn_s_setprio(P); }
// a pointer the caller knows to be wave-uniform, handed to the compiler as a scalar pair (global accesses then take the SGPR base + VGPR offset form)
template <typename T> __device__ __forceinline__ T* dm_uniform_ptr(T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T*)(((uint64_t)hi << 32) | (uint64_t)lo);
}
__device__ __forceinline__ int dm_atomic_or(int* p, int v) { return atomicOr(p, v); }
__device__ __forceinline__ int dm_atomic_min(int* p, int v) { return atomicMin(p, v); }
__device__ __forceinline__ int dm_atomic_add(int* p, int v) { return atomicAdd(p, v); }
__device__ __forceinline__ int dm_popc64(uint64_t v) { return __popcll(v); }
__device__ __forceinline__ int dm_ctz32(uint32_t v) { return __builtin_ctz(v); }
namespace dmk {
__device__ __forceinline__ float wave_shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double wave_shfl(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int wave_shfl(int v, int src) { return __shfl(v, src, 64); }
// broadcast from a wave-uniform lane: v_readlane_b32 (no LDS traffic, SGPR result)
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_bcast(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ double lane_bcast(double v, int src) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __ballot(p); }
// sum over the 64 lanes, result in every lane.  fp32: DPP butterfly inside each row of 16, row broadcasts across rows.
#define DM_DPP_F(v, ctrl, rmask, bctl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, bctl))
__device__ __forceinline__ float wave_sum(float v) {
    v += DM_DPP_F(v, 0xB1, 0xf, true);     // quad_perm [1,0,3,2]
    v += DM_DPP_F(v, 0x4E, 0xf, true);     // quad_perm [2,3,0,1]
    v += DM_DPP_F(v, 0x141, 0xf, true);    // row_half_mirror
    v += DM_DPP_F(v, 0x140, 0xf, true);    // row_mirror: every lane holds the sum of its row
    v += DM_DPP_F(v, 0x142, 0xa, false);   // row_bcast15 into rows 1 and 3
    v += DM_DPP_F(v, 0x143, 0xc, false);   // row_bcast31 into rows 2 and 3: row 3 holds the total
    return lane_bcast(v, 63);
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// exchange with lane (l ^ mask): DPP quad permutes for mask 1 and 2, LDS crossbar (ds_bpermute) beyond
template <int MASK> __device__ __forceinline__ float wave_shfl_xor_c(float v) {
    if (MASK == 1) return DM_DPP_F(v, 0xB1, 0xf, true);
    if (MASK == 2) return DM_DPP_F(v, 0x4E, 0xf, true);
    return __shfl_xor(v, MASK, 64);
}
template <int MASK> __device__ __forceinline__ double wave_shfl_xor_c(double v) { return __shfl_xor(v, MASK, 64); }
__device__ __forceinline__ float dm_med3(float lo, float x, float hi) { return __builtin_amdgcn_fmed3f(lo, x, hi); }
__device__ __forceinline__ double dm_med3(double lo, double x, double hi) { return fmax(lo, fmin(hi, x)); }
// 1/sqrt and 1/x: hardware approximation + one Newton step in fp32 (rel. error ~1e-7), exact in fp64
__device__ __forceinline__ float dm_rsqrt(float x) { float r = __builtin_amdgcn_rsqf(x); return r * (1.5f - 0.5f * x * r * r); }
__device__ __forceinline__ double dm_rsqrt(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ float dm_rcp(float x) { float r = __builtin_amdgcn_rcpf(x); return r * (2.0f - x * r); }
__device__ __forceinline__ double dm_rcp(double x) { return 1.0 / x; }
// (bit l of MASK) ? a : b for a compile-time lane mask: the mask is an SGPR pair built on the scalar unit, the select one VALU (no v_cmp
// per lane test).  `z` is a wave-uniform zero the caller re-makes opaque per phase, so that the masks are rebuilt where they are used
// (one s_xor_b64) instead of being hoisted out of the 20-update loop and spilled.
template <uint64_t MASK> __device__ __forceinline__ float lane_sel(float a, float b, int, uint64_t z) {
    const uint64_t mk = MASK ^ z;
    float out;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(b), "v"(a), "s"(mk));
    return out;
}
template <uint64_t MASK> __device__ __forceinline__ double lane_sel(double a, double b, int l, uint64_t) { return ((MASK >> (l & 63)) & 1ull) ? a : b; }
// lane R takes `newv`: the select mask is an SGPR pair made on the scalar unit from `one` (s_lshl + s_mov) instead of a v_cmp; `one` is
// re-made opaque per sweep, or the 64 masks are hoisted out of the iteration loop and spilled (then every row reloads its pair by v_readlane)
template <int R> __device__ __forceinline__ float row_sel_c(float oldv, float newv, int, uint32_t one) {
    const uint32_t m32 = one << (R & 31);
    const uint64_t mk = (R < 32) ? (uint64_t)m32 : ((uint64_t)m32 << 32);
    float out;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(oldv), "v"(newv), "s"(mk));
    return out;
}
template <int R> __device__ __forceinline__ double row_sel_c(double oldv, double newv, int l, uint32_t) { return l == R ? newv : oldv; }
// per-lane array indexed by a wave-uniform runtime index.  For float it is two 32-wide register vectors that the
// backend addresses with M0-relative VGPR indexing, so a row of A never leaves the VGPRs.
template <typename Real, int N> struct RowFile {
    Real v[N];
    __device__ __forceinline__ Real get(int r) const { return v[r]; }
    __device__ __forceinline__ void set(int r, Real x) { v[r] = x; }
};
template <> struct RowFile<float, 32> {
    typedef float v32 __attribute__((ext_vector_type(32)));
    v32 a;
    __device__ __forceinline__ float get(int r) const { return a[r]; }
    __device__ __forceinline__ void set(int r, float x) { a[r] = x; }
};
template <> struct RowFile<float, 64> {
    typedef float v32 __attribute__((ext_vector_type(32)));
    v32 a, b;
    __device__ __forceinline__ float get(int r) const { return (r < 32) ? a[r] : b[r - 32]; }
    __device__ __forceinline__ void set(int r, float x) { if (r < 32) a[r] = x; else b[r - 32] = x; }
};
// 48 rows: the compiled dog3d class at 2 waves / SIMD has room for 16 more row registers; 25.6 % of its substeps have more than 32 rows,
// 1.8 % more than 48 (profiles/r03_rows_hist.txt), so the HBM / L2 overflow block leaves the common path
template <> struct RowFile<float, 48> {
    typedef float v32 __attribute__((ext_vector_type(32)));
    typedef float v16 __attribute__((ext_vector_type(16)));
    v32 a; v16 b;
    __device__ __forceinline__ float get(int r) const { return (r < 32) ? a[r] : b[r - 32]; }
    __device__ __forceinline__ void set(int r, float x) { if (r < 32) a[r] = x; else b[r - 32] = x; }
};
// 40 rows: the borrowed-lane path of the two-per-wave kernel (dm_device_duo.h: a heavy character of up to 40 rows = 12 contacts)
template <> struct RowFile<float, 40> {
    typedef float v32 __attribute__((ext_vector_type(32)));
    typedef float v8 __attribute__((ext_vector_type(8)));
    v32 a; v8 b;
    __device__ __forceinline__ float get(int r) const { return (r < 32) ? a[r] : b[r - 32]; }
    __device__ __forceinline__ void set(int r, float x) { if (r < 32) a[r] = x; else b[r - 32] = x; }
};
// two / four packed reals: v_pk_fma_f32 / ds_read_b128 operands
template <typename Real> struct VecT;
template <> struct VecT<float> { typedef float v2 __attribute__((ext_vector_type(2))); typedef float v4 __attribute__((ext_vector_type(4))); };
template <> struct VecT<double> { typedef double v2 __attribute__((ext_vector_type(2))); typedef double v4 __attribute__((ext_vector_type(4))); };
// G[i] = sum_k y_l[k] y_i[k], i < 32: Gram rows of the first 32 constraint rows.
// fp32: the one dense contraction of the path goes to the matrix core -- NP2 x v_mfma_f32_32x32x2_f32 (K = 2 per issue, full
// fp32 multiply-add).  Operand layout: lanes 0..31 carry Y[k0][lane], lanes 32..63 carry Y[k0+1][lane-32]; A and B operands
// are the same register (G = Y^T Y).  The 32x32 result comes back as 16 accumulators per lane (lane = column, rows split
// between the wave halves); symmetry turns lane j's column into row j, the other half arrives by one v_permlane32_swap each.
template <int NP2> __device__ __forceinline__ void wave_gram32(const VecT<float>::v2* y2, float (&out)[32]) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
        // gfx950 v_permlane32_swap: (a0, a1) -> [a0.lo | a1.lo]: the upper lanes receive the lower lanes' odd column
        const float opnd = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(y2[p][0]), __float_as_uint(y2[p][1]), false, false)[0]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(opnd, opnd, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        // lanes < 32 (the only ones with a row) keep their accumulator and take the rows held by their upper partner
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[v]), false, false);
        out[8 * (v / 4) + (v % 4)] = __uint_as_float(sw[0]);
        out[8 * (v / 4) + 4 + (v % 4)] = __uint_as_float(sw[1]);
    }
}
// G[i] = sum_k y_l[k] y_i[k], i < 64: all 64 Gram rows on the matrix core.  With op0 = [Y[k0][0..31] | Y[k1][0..31]] and
// op1 = [Y[k0][32..63] | Y[k1][32..63]] (one v_permlane32_swap of the lane's column pair) the four 32 x 32 blocks are the MFMA
// chains (op0,op0), (op0,op1), (op1,op0), (op1,op1); lane j of either half holds column j of each block, rows split between the
// halves.  Row r < 32 of G is [block00 column r | block10 column r], row 32 + r is [block01 column r | block11 column r] (G is
// symmetric), so two swaps per accumulator pair hand every lane its full row.
template <int NP2> __device__ __forceinline__ void wave_gram64(const VecT<float>::v2* y2, float (&out)[64]) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    const f16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f16v a00 = z, a01 = z, a10 = z, a11 = z;
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(y2[p][0]), __float_as_uint(y2[p][1]), false, false);
        const float op0 = __uint_as_float(sw[0]), op1 = __uint_as_float(sw[1]);
        a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(op0, op0, a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(op0, op1, a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(op1, op0, a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(op1, op1, a11, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        // lower lanes keep block00 / block10 and receive the upper partner's halves of them; upper lanes the same for block01 / block11
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a00[v]), __float_as_uint(a01[v]), false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a10[v]), __float_as_uint(a11[v]), false, false);
        out[8 * (v / 4) + (v % 4)] = __uint_as_float(s0[0]);
        out[8 * (v / 4) + 4 + (v % 4)] = __uint_as_float(s0[1]);
        out[32 + 8 * (v / 4) + (v % 4)] = __uint_as_float(s1[0]);
        out[32 + 8 * (v / 4) + 4 + (v % 4)] = __uint_as_float(s1[1]);
    }
}
// Half of the above: H = 0 gives every lane entries 0..31 of its row, H = 1 entries 32..63 (two MFMA chains and 32 accumulators live
// at a time instead of four and 64: the fallback of the two-per-wave kernel runs inside its register budget)
template <int NP2, int H> __device__ __forceinline__ void wave_gram64_half(const VecT<float>::v2* y2, float (&out)[32]) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    const f16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f16v b0 = z, b1 = z;
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(y2[p][0]), __float_as_uint(y2[p][1]), false, false);
        const float op0 = __uint_as_float(sw[0]), op1 = __uint_as_float(sw[1]);
        b0 = __builtin_amdgcn_mfma_f32_32x32x2f32(H ? op1 : op0, op0, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_32x32x2f32(H ? op1 : op0, op1, b1, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(b0[v]), __float_as_uint(b1[v]), false, false);
        out[8 * (v / 4) + (v % 4)] = __uint_as_float(s0[0]);
        out[8 * (v / 4) + 4 + (v % 4)] = __uint_as_float(s0[1]);
    }
}
template <int NP2, int H> __device__ __forceinline__ void wave_gram64_half(const VecT<double>::v2* y2, double (&out)[32]) {
#pragma unroll 1
    for (int i = 0; i < 32; ++i) {
        double a = 0;
        for (int p = 0; p < NP2; ++p) a += y2[p][0] * lane_bcast(y2[p][0], 32 * H + i) + y2[p][1] * lane_bcast(y2[p][1], 32 * H + i);
        out[i] = a;
    }
}
template <int NP2> __device__ __forceinline__ void wave_gram64(const VecT<double>::v2* y2, double (&out)[64]) {
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
        double a = 0;
#pragma unroll
        for (int p = 0; p < NP2; ++p) a += y2[p][0] * lane_bcast(y2[p][0], i) + y2[p][1] * lane_bcast(y2[p][1], i);
        out[i] = a;
    }
}
template <int NP2> __device__ __forceinline__ void wave_gram32(const VecT<double>::v2* y2, double (&out)[32]) {
#pragma unroll 1
    for (int i = 0; i < 32; ++i) {
        double a = 0;
#pragma unroll
        for (int p = 0; p < NP2; ++p) a += y2[p][0] * lane_bcast(y2[p][0], i) + y2[p][1] * lane_bcast(y2[p][1], i);
        out[i] = a;
    }
}
}
#endif

#ifndef DM_PRIO
#define DM_PRIO 1      // s_setprio by phase and load (dm_device_duo.h has the rationale and the measurements); 0: the round-3 kernels
#endif
// one character per wavefront: the same rule -- dependent-chain phases above the throughput phases of the SIMD's other wave
// DM_YPREF: the tree classes' y = L^-T J^T loop requests dof k - 1's records before dof k's arithmetic (EnvSim::substep_post)
#ifndef DM_YPREF
#define DM_YPREF 1
#endif
// DM_STPREF: dyn_subtree of the one-per-wave tree classes statically unrolled, one member ahead.  OFF: -0.5 % kernel time on the dog, but the compiler contracts the
// unrolled sums into different FMAs than the rolled loop's -- the dog's 6 committed replay bundles no longer reproduce bit for bit on the GPU (they do on the emulator), and the
// free-running DM-physics v2 dog leaves its oracle trajectory (tests/test_physics_v2.py).  Every other look-ahead of round 6 keeps the bits.
#ifndef DM_STPREF
#define DM_STPREF 0
#endif
// DM_PAIRPREF: the self-collision passes read their operands unpredicated, the two-per-wave kernel one pass ahead
#ifndef DM_PAIRPREF
#define DM_PAIRPREF 1
#endif
// DM_DRPREF: dyn_row (the dense classes' mass-matrix rows) requests pair p + 1's dof records before pair p's dot products; 2: one pair per scheduling region
#ifndef DM_DRPREF
#define DM_DRPREF 2
#endif
// DM_ELPREF: tree_elim requests pivot q + 1's published entries before pivot q's rank-1 update
#ifndef DM_ELPREF
#define DM_ELPREF 1
#endif
// DM_TFPREF: tree_fwd requests all its row reads before the first lane select
#ifndef DM_TFPREF
#define DM_TFPREF 1
#endif
#ifndef DM_YPREF_DENSE_FENCE
#define DM_YPREF_DENSE_FENCE 1
#endif
// DM_LCPREF: tree_load_col reads the momentum records in groups of this many, one group ahead (0: the fenced groups of 8 of rounds 3-5)
#ifndef DM_LCPREF
#define DM_LCPREF 4
#endif
#ifndef DM_PRIO_ONE_CHOL
#define DM_PRIO_ONE_CHOL (DM_PRIO ? 1 : 0)
#define DM_PRIO_ONE_Y (DM_PRIO ? 1 : 0)
#define DM_PRIO_ONE_BACK (DM_PRIO ? 1 : 0)
#define DM_PRIO_ONE_KIN (DM_PRIO ? 1 : 0)
#endif
namespace dmk {

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1 (where the index has to be a template argument)
// Workgroup -> unit of work (env / env pair), XCD-aware (round 6).  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs, each with its own L2; an env's
// rows (172 B of pose, 136 B of torque, ...) are not multiples of a cache line, so with unit = blockIdx the line shared by two neighbouring envs was fetched into two L2s
// and written back from two -- the "1.41 x" HBM traffic of rounds 2-5.  XCD x takes the contiguous units [x G / 8, (x + 1) G / 8): neighbours in memory meet in one L2.
// Which workgroup steps an env changes nothing in its arithmetic.
#ifndef DM_XCD_MAP
#define DM_XCD_MAP 1
#endif
#ifdef DM_EMU
static inline int dm_wg_unit() { return (int)blockIdx.x; }
#else
__device__ __forceinline__ int dm_wg_unit() {
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    return (DM_XCD_MAP && (G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
}
#endif
template <int B, int E, typename F> DM_DEV void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// BROAD classes: bounding radius of each link's self-collision capsule about its COM, plus its contact threshold (filled once per kernel)
template <typename Real, typename C, bool ON = C::BROAD> struct LdsBroad { Real brad[C::NJ]; };
template <typename Real, typename C> struct LdsBroad<Real, C, false> {};

// Per-wave LDS record.
template <typename Real, typename C>
struct Lds : LdsBroad<Real, C> {
    static constexpr int NJ = C::NJ, ND = C::ND, NP = C::NP, NCAP = C::NCAP;
    // Lower-triangular storage of H / its Cholesky factor: row k holds k+1 entries padded to a multiple of LPAD (4: 16-B
    // aligned rows for ds_read_b128 broadcasts; 2 for the large class, which trades them for 8-B reads to fit 8 waves per CU);
    // the diagonal slot holds 1/L_kk after the factorisation.
    static constexpr int LPAD = C::LPAD;
    // offset of row k = sum_{i<k} LPAD * ceil((i+1)/LPAD) = LPAD * (q (q+1) / 2 * LPAD + r (q+1)), q = k / LPAD, r = k % LPAD
    static constexpr int lrow(int k) { return LPAD * ((k / LPAD) * (k / LPAD + 1) / 2 * LPAD + (k % LPAD) * (k / LPAD + 1)); }
    // TREE classes store the factor of H = L^T L by COLUMNS, as a skyline: column j holds L_ij for i = 4 floor(j / 4) .. its last
    // descendant (quads aligned for 16-B broadcast reads), L_ij at Lt[lcb(j) + i], the diagonal slot holds 1 / L_jj (dog3d: 968 words
    // against 2 112 for the packed triangle)
    static constexpr int lcb(int j) { if constexpr (C::TREE) return C::Topo::T.colbase[j] - 4 * (j / 4); else return 0; }
    static constexpr int tree_words() { if constexpr (C::TREE) return C::Topo::T.lwords; else return 0; }
    static constexpr int kLWords = C::TREE ? tree_words() : lrow(ND);
    MdlLds<Real, C> mdl;
    Real pose[NP], vel[NP], tar[NP];
    Real tau[ND], rhs[ND];                 // (bias force: dofrec[k][7]; SPD force xs: aliases Ic, see EnvSim::xs)
    alignas(32) Real dofrec[ND][8];        // per dof: world axis a(3), g = (p_joint - p_root) x a (3), unconstrained velocity v*, bias force C
    Real R[NJ][9], p[NJ][3], com[NJ][3], w[NJ][3], vj[NJ][3], al[NJ][3], aj[NJ][3];
    Real Rb[C::ROT ? kMaxRotLinks : 1][C::ROT ? 9 : 1];   // body frames of the links whose body rotation is not the identity (== R when the class has no attach rotations)
    union alignas(32) {
        struct { Real f[NJ][3], n[NJ][3], Iw[NJ][6], Fs[NJ][3], Ns[NJ][3], Ic[NJ][10]; };   // Newton-Euler pass (dynamics)
        struct { int csel[NCAP]; Real cdistc[NCAP];                                          // manifold-reduction scratch (by candidate)
                 Real ct[kMaxContacts][8]; };                                                // contact slots: x(3), n(3), dist, links a | b << 8 (b = 255: ground)
    };
    // scratch() aliases Lt outside the update loop: emit() needs 2 NP + 7 NJ words (kin pose / vel, joint positions, reduction terms), the AMP expert
    // sample 4 NP -- a new topology or class must not silently overrun into the fields behind Lt
    static_assert(kLWords >= 2 * C::NP + 7 * C::NJ && kLWords >= 4 * C::NP, "Lds::Lt is too small for the scratch users (emit, amp_expert)");
    alignas(32) Real Lt[kLWords];
    Real kin[8];                           // kin origin pos(3), origin rot(4)
    Real sc[8];                            // small float scratch (kin / sim COM velocity, episode-end flag)
    double clk[6];                         // kin_time, ctrl_time, init_time_offset, timer_time, timer_max
    int flg[8];                            // need_new_action, contact_mask, episode_count, valid, nrows, ncontacts, parked, over
    int fall_mask;                         // links whose ground contact is a fall (bit j), from link_info at load
    int getup;                             // heading_amp_getup: the get-up timer of the env is running (mirror of the goal row, goal_sync_flags)
    Real obj[C::OBJ ? OB_WIDTH : 1];       // OBJ classes: the free rigid body (dribble_amp's ball): pos, rot wxyz, vel, ang vel
};

// What the end-of-call outputs need of a character whose episode ended mid-call (two characters per wavefront, early episode end)
// (OBJ classes: the free body's record too -- the ball of a parked character keeps being integrated beside its partner's updates, which the one-per-wave kernel, leaving its
// update loop at the episode's end, does not do.  An empty base otherwise: the biped class's snapshot, and with it the headline kernel's LDS layout, is unchanged)
template <typename Real, bool ON> struct ParkSnapObj { Real obj[OB_WIDTH]; };
template <typename Real> struct ParkSnapObj<Real, false> {};
template <typename Real, typename C>
struct ParkSnap : ParkSnapObj<Real, C::OBJ> { Real pose[C::NP], vel[C::NP], kin[8]; double clk[6]; int flg[4]; };

enum { CLK_KIN = 0, CLK_CTRL, CLK_INIT_OFF, CLK_TIMER, CLK_TIMER_MAX };
enum { FLG_NEED_ACTION = 0, FLG_CONTACT, FLG_EPISODE, FLG_VALID, FLG_NROWS, FLG_NCONT, FLG_PARKED, FLG_OVER };
// ---- the draw tape (dm_types.h TP_*; ModelDev::draw_tape): the reference's two generators as position-indexed tables, consumed in call order by lane 0 of an env.
// eng 0 = cMathUtil::gRand, 1 = the scene's mRand.  The arithmetic around a tabulated value is the reference's own expression and must round like the
// reference's build (gcc, no FMA): contraction is switched off for these few lines.
DM_HD double tape_u01(double* T, int eng) {                                  // uniform_real_distribution<double>(0, 1): two raw values
    const int p = (int)T[TP_POS_G + eng];
    if (p < 0 || p >= TP_K) { T[TP_ERR] = 1.0; return 0.5; }
    T[TP_POS_G + eng] = (double)(p + 2);
    return T[(eng ? TP_UM : TP_UG) + p];
}
DM_HD double tape_uniform(double* T, int eng, double lo, double hi) {        // cRand::RandDouble(min, max) (util/Rand.cpp:30-41): no draw when min == max
#pragma clang fp contract(off)
    if (lo == hi) return lo;
    double u = tape_u01(T, eng);
    u = lo + (u * (hi - lo));
    return u;
}
DM_HD double tape_normal(double* T, double mean, double stdev) {             // cRand::RandDoubleNorm (:50-55) on mRand: a polar pair serves two calls
#pragma clang fp contract(off)
    double v;
    if (T[TP_NAVAIL] != 0.0) { v = T[TP_NSAVED]; T[TP_NAVAIL] = 0.0; }
    else {
        const int p = (int)T[TP_POS_M];
        if (p < 0 || p >= TP_K) { T[TP_ERR] = 1.0; return mean; }
        const double* n = T + TP_NM + 3 * p;
        v = n[0]; T[TP_NSAVED] = n[1]; T[TP_NAVAIL] = 1.0; T[TP_POS_M] = (double)(p + (int)n[2]);
    }
    v = mean + stdev * v;
    return v;
}
DM_HD int tape_int_range(double* T, int lo, int hi) {                         // cRand::RandInt(min, max) (:62-75) on mRand
    if (lo == hi) return lo;
    const int p = (int)T[TP_POS_M];
    if (p < 0 || p >= TP_K) { T[TP_ERR] = 1.0; return lo; }
    const double* n = T + TP_IM + 2 * p;
    T[TP_POS_M] = (double)(p + (int)n[1]);
    return lo + (int)n[0] % (hi - lo);
}
// n x cTimer::Reset (util/Timer.cpp:55-73) with the parameters of the tape header (the last one stands), then the limit test mode pins (RLSceneSimChar.cpp:277-284)
DM_HD double tape_time_limit(double* T, int n) {
#pragma clang fp contract(off)
    double mt = T[TP_TMAX];
    for (int i = 0; i < n; ++i) {
        if (T[TP_TEXP] > 0.0) {
            const int p = (int)T[TP_POS_G];
            if (p < 0 || p >= TP_K) { T[TP_ERR] = 1.0; break; }
            T[TP_POS_G] = (double)(p + 2);
            const double lambda = 1 / T[TP_TEXP];
            mt = T[TP_TMIN] + T[TP_EG + p] / lambda;
            mt = mt < T[TP_TMAX] ? mt : T[TP_TMAX];                                              // std::min(max_time, mTimeMax)
        } else mt = tape_uniform(T, 0, T[TP_TMIN], T[TP_TMAX]);
    }
    if (T[TP_TPIN] >= 0.0) mt = T[TP_TPIN];
    return mt;
}


// TAPS = false compiles every debug tap / phase timer out of the instruction stream (production step kernel).
// LW = lanes per character: 64 (one character per wavefront) or 32 (two characters per wavefront, dm_device_duo.h).
template <typename Real, typename C, bool TAPS = true, int LW = kWave>
struct EnvSim {
    typedef Lds<Real, C> L;
    static constexpr int NJ = C::NJ, ND = C::ND, NP = C::NP, NCAP = C::NCAP, CPL = C::NCAP / kWave, RREG = C::RREG;
    static constexpr int NP2 = ND / 2;                 // register pairs per dof vector (ND is even for every class)
    static constexpr int NP2X = NP2 + (C::OBJ ? 3 : 0); // + the free body's 3 linear + 3 angular velocities (rows of Y = M^-1/2 J^T)
    typedef V3<Real> v3; typedef Q4<Real> q4; typedef M3<Real> m3;
    typedef typename VecT<Real>::v2 R2; typedef typename VecT<Real>::v4 R4;
    const ModelDev<Real>& m; L& s; int l;
    int li = 0;                                         // link_info word of this lane's link (0 for lanes >= J)
    int clip = 0;                                       // multi-clip dataset: the clip this env's kinematic character is on (goal row GS_CLIP; AMP / tap instantiations)
    int cand_link[CPL]; Real cand_loc[CPL][3], cand_rad[CPL];   // this lane's ground-contact candidates
    static constexpr int PPL = C::NPAIRCAP / kWave;             // self-collision pairs per lane
    int pair_code[PPL];
    // phase-cycle accounting (profiling kernel only): the deltas accumulate in registers and reach memory once, at the end of the launch -- a
    // read-modify-write of `prof` at every mark stalled the wave for a memory round trip per mark and inflated every phase (round 4)
    long long* prof = nullptr; long long tprev = 0; long long pacc[16];
    DM_DEV EnvSim(const ModelDev<Real>& m_, L& s_, int l_) : m(m_), s(s_), l(l_) {}
    // wave priority inside substep_post: never below the class's floor (C::PRIO_FLOOR; the fallback class of the two-per-wave kernel runs raised throughout)
    static constexpr int prio_of(int p) { return p > C::PRIO_FLOOR ? p : C::PRIO_FLOOR; }
    DM_DEV void mark(int phase) {
        if (TAPS && prof) {
            const long long t = dm_clock(), d = t - tprev; tprev = t;
#pragma unroll
            for (int i = 0; i < 16; ++i) if (i == phase) pacc[i] += d;       // (static register indices whatever `phase` is)
        }
    }
    DM_DEV void prof_begin(long long* p) {
        prof = p;
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[i] = 0;
        tprev = dm_clock();
    }
    DM_DEV void prof_flush() {
        if (TAPS && prof && l == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) prof[i] += pacc[i];
        }
    }
    DM_DEV void sync() const { __syncthreads(); }
    DM_DEV Real* scratch() const { return &s.Lt[0]; }   // L is dead outside the update loop: kin pose / vel / reductions live there
    DM_DEV Real& Lx(int r, int c) const { return s.Lt[L::lrow(r) + c]; }
    DM_DEV Real* xs() const { return &s.Ic[0][0]; }     // SPD joint forces Kp e + Kd e_v: alive from spd_rhs_pre to spd_post, while Ic is dead
    // world body frame of link j: its joint frame unless the link carries a body attach rotation (ClsLarge, compact table)
    DM_DEV const Real* Rbp(int j) const {
        if (C::ROT) { const int bo = s.mdl.rot_idx[C::ROT ? j : 0] & 15; if (bo) return s.Rb[C::ROT ? bo - 1 : 0]; }
        return s.R[j];
    }
    static DM_DEV v3 zero3() { return mk3((Real)0, (Real)0, (Real)0); }

    // ------------------------------------------------------------------ HBM <-> LDS
    DM_DEV void load_cands() {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = l + kWave * q;
            cand_link[q] = 0; cand_rad[q] = 0; cand_loc[q][0] = cand_loc[q][1] = cand_loc[q][2] = 0;
            if (c < m.NC) { cand_link[q] = m.cand_link[c]; cand_rad[q] = m.cand_rad[c]; for (int k = 0; k < 3; ++k) cand_loc[q][k] = m.cand_loc[c * 3 + k]; }
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) { const int c = l + kWave * q; pair_code[q] = (c < m.NPAIR) ? m.pair_code[c] : -1; }
    }
    DM_DEV void load_model() {
        uint32_t* dst = reinterpret_cast<uint32_t*>(&s.mdl);
        for (int i = l; i < m.mdl_words; i += LW) dst[i] = m.mdl_blob[i];
        if (LW == kWave) load_cands();
        sync();
        li = (l < m.J) ? s.mdl.link_info[l] : 0;
        if (l == 0) { int fm = 0; for (int j = 0; j < m.J; ++j) fm |= DM_LI_FALL(s.mdl.link_info[j]) << j; s.fall_mask = fm; s.getup = 0; }
        if constexpr (C::BROAD) { if (l < m.J) { const Real* c = s.mdl.cap[l]; s.brad[l] = dm_sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + c[3] + s.mdl.thresh[l]; } }
    }
    DM_DEV void load(const EnvState<Real>& st, int e) {
        load_model();
        for (int i = l; i < m.P; i += LW) { s.pose[i] = st.pose[(size_t)e * m.P + i]; s.vel[i] = st.vel[(size_t)e * m.P + i]; s.tar[i] = st.tar[(size_t)e * m.P + i]; }
        for (int i = l; i < m.D; i += LW) s.tau[i] = st.tau[(size_t)e * m.D + i];
        if (l < 8) s.kin[l] = st.kin[(size_t)e * 8 + l];
        if (l < 6) s.clk[l] = st.clock[(size_t)e * 6 + l];
        if (l < 4) s.flg[l] = st.flag[(size_t)e * 4 + l];
        if (l == 0) { s.flg[FLG_PARKED] = 0; s.flg[FLG_OVER] = 0; }
        if (C::OBJ && st.obj && l < OB_WIDTH) s.obj[C::OBJ ? l : 0] = st.obj[(size_t)e * OB_WIDTH + l];
        sync();
    }
    // `act`: lanes of a character that does not take part still pass the barriers (two characters per wavefront)
    DM_DEV void store(const EnvState<Real>& st, int e, bool act = true) {
        DM_OPAQUE_V(l);
        sync();
        if (!act) return;
        for (int i = l; i < m.P; i += LW) { st.pose[(size_t)e * m.P + i] = s.pose[i]; st.vel[(size_t)e * m.P + i] = s.vel[i]; st.tar[(size_t)e * m.P + i] = s.tar[i]; }
        for (int i = l; i < m.D; i += LW) st.tau[(size_t)e * m.D + i] = s.tau[i];
        if (l < 8) st.kin[(size_t)e * 8 + l] = s.kin[l];
        if (l < 6) st.clock[(size_t)e * 6 + l] = s.clk[l];
        if (l < 4) st.flag[(size_t)e * 4 + l] = s.flg[l];
        if (C::OBJ && st.obj && l < OB_WIDTH) st.obj[(size_t)e * OB_WIDTH + l] = s.obj[C::OBJ ? l : 0];
    }
    // Is the episode over after this update?  The reference's driver asks after EVERY update and ends the episode there, not at
    // the next action boundary (DeepMimic.py:62-80 update_world: world.update(timestep); is_episode_end -> end_episode, reset,
    // break; learning/rl_agent.py:_end_path records the state and reward of that moment).  Same tests as emit(): contact fall
    // (cSceneImitate::CheckTerminate, SceneImitate.cpp:193-205), a finished non-looping clip, the episode timer.  The root-rotation
    // failure test (enable_root_rot_fail) is root_rot_failed_now(), added by update() in the instantiations that carry it.
    // cSceneImitate::CheckRootRotFail (SceneImitate.cpp:466-492, `--enable_root_rot_fail`, off in every shipped arg file) on the state after this
    // update, for the per-update episode end of DM_END_EPISODE_EARLY: the driver asks IsEpisodeEnd after every update, so a character that has turned
    // more than 90 degrees away from the clip ends its episode at THAT update (round 4; until then the test ran at the action boundary only).  Compiled
    // into the AMP / tap instantiations, which the host selects for a scene with the flag on; lane 0.
    DM_DEV bool root_rot_failed_now() const {
        return quat_theta(qmul(kin_root_rot(s.clk[CLK_KIN]), qconj(ldq(s.pose + 3)))) > (Real)(0.5 * DM_PI);
    }
    DM_DEV bool episode_over_now() const {
        const bool fail = (m.enable_fall_end && has_fallen(nullptr)) || (!m.scene_amp && !m.loop && s.clk[CLK_KIN] >= m.duration);
        return fail || (s.clk[CLK_TIMER] >= s.clk[CLK_TIMER_MAX]);
    }
    // Two characters share one instruction stream: a character whose episode is over mid-call has what the outputs need of
    // its record copied aside (LDS) and is then moved out of the way -- 10 m up, at rest -- so that the updates its partner
    // still needs cost it no contact rows; unpark() brings the copy back before the outputs are written.  `act`: lanes of
    // the other character only pass the barriers.
    DM_DEV void park(ParkSnap<Real, C>& snap, bool act) {
        sync();
        if (act) {
            for (int i = l; i < m.P; i += LW) { snap.pose[i] = s.pose[i]; snap.vel[i] = s.vel[i]; }
            if (l < 8) snap.kin[l] = s.kin[l];
            if (l < 6) snap.clk[l] = s.clk[l];
            if (l < 4) snap.flg[l] = s.flg[l];
            if constexpr (C::OBJ) { if (l < OB_WIDTH) snap.obj[l] = s.obj[l]; }
        }
        sync();
        if (act) {
            for (int i = l; i < m.P; i += LW) s.vel[i] = (Real)0;
            if (l == 0) { s.pose[1] += (Real)10; s.flg[FLG_CONTACT] = 0; s.flg[FLG_PARKED] = 1; }
        }
        sync();
    }
    DM_DEV void unpark(const ParkSnap<Real, C>& snap) {
        sync();
        if (s.flg[FLG_PARKED]) {
            for (int i = l; i < m.P; i += LW) { s.pose[i] = snap.pose[i]; s.vel[i] = snap.vel[i]; }
            if (l < 8) s.kin[l] = snap.kin[l];
            if (l < 6) s.clk[l] = snap.clk[l];
            if (l < 4) s.flg[l] = snap.flg[l];
            if constexpr (C::OBJ) { if (l < OB_WIDTH) s.obj[l] = snap.obj[l]; }
        }
        sync();
        if (l == 0) s.flg[FLG_PARKED] = 0;
        sync();
    }

    // ------------------------------------------------------------------ kinematics (level-synchronous over tree depth)
    // Fills R,p (joint frames), com, Rb (body frames), w (link angular velocity), vj (velocity of the joint origin)
    // and, for the zero-qddot Newton-Euler pass, al (angular acceleration) and aj (acceleration of the joint origin)
    // with base acceleration a0 (gravity enters as a0 = -g).
    DM_DEV void kinematics(const Real* pose, const Real* vel, v3 a0) {
        const int par = DM_LI_PARENT(li), dep = (l < m.J) ? DM_LI_DEPTH(li) : -1, jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li);
        // joint-local rotation and angular velocity of this lane's joint (trig evaluated once, outside the level loop)
        m3 Rl = m3_identity<Real>(); v3 wl = zero3();
        if (l < m.J) {
            if (par < 0) Rl = quat_to_rot(ldq(pose + 3));
            else if (jt == JT_SPHERICAL) { Rl = quat_to_rot(ldq(pose + off)); wl = ld3(vel + off); }
            else if (jt == JT_REVOLUTE) { Rl = rot_z(pose[off]); wl.z = vel[off]; }
        }
        for (int d = 0; d <= m.max_depth; ++d) {
            if (dep == d) {
                m3 Rj; v3 pj, w, vj, al, aj;
                if (par < 0) {
                    Rj = Rl; pj = ld3(pose);
                    w = ld3(vel + 3); vj = ld3(vel); al = zero3(); aj = a0;
                } else {
                    m3 Rp = ldm3(s.R[par]);
                    v3 r = Rp * ld3(s.mdl.attach[l]);
                    pj = ld3(s.p[par]) + r;
                    if (C::ROT && !DM_LI_AROT_ID(li)) Rp = Rp * ldm3(s.mdl.attach_rot[C::ROT ? ((s.mdl.rot_idx[C::ROT ? l : 0] >> 4) & 15) - 1 : 0]);
                    Rj = Rp * Rl;
                    v3 wp = ld3(s.w[par]), alp = ld3(s.al[par]);
                    v3 wrel = Rj * wl;
                    w = wp + wrel;
                    const v3 wxr = cross(wp, r);
                    vj = ld3(s.vj[par]) + wxr;
                    al = cross_add(alp, wp, wrel);
                    aj = cross_add(cross_add(ld3(s.aj[par]), alp, r), wp, wxr);
                }
                stm3(s.R[l], Rj); st3(s.p[l], pj); st3(s.w[l], w); st3(s.vj[l], vj); st3(s.al[l], al); st3(s.aj[l], aj);
                st3(s.com[l], pj + Rj * ld3(s.mdl.battach[l]));
                if (C::ROT && !DM_LI_BROT_ID(li)) { const int bo = (s.mdl.rot_idx[C::ROT ? l : 0] & 15) - 1; stm3(s.Rb[C::ROT ? bo : 0], Rj * ldm3(s.mdl.brot[C::ROT ? bo : 0])); }
            }
            sync();
        }
    }

    // ------------------------------------------------------------------ mass matrix H (lower triangle, into s.L) and bias force C
    // iset: 0 = SPD inertias, 1 = simulator inertias.  diag_scale * kd is added to the diagonal (SPD: dt).
    // Requires kinematics() for the same state.  Also fills dofrec[k] = (axis, g) used by the constraint rows.
    DM_DEV void dynamics(int iset, Real diag_scale) {
        dyn_links(iset);
        if (l < m.D) dyn_dofrec(l);
        sync();
        dyn_subtree();
        sync();
        if constexpr (C::TREE) { if (l < m.D) dyn_mom(l, diag_scale); }
        else if (l < m.D) dyn_row(l, diag_scale);
        sync();
    }
    // per-link force / moment about the COM and world inertia about the COM (lane = link)
    DM_DEV void dyn_links(int iset) {
        if (l < m.J) {
            m3 Rb = ldm3(Rbp(l));
            const Real* Id = s.mdl.inertia[iset][l];
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
            v3 ac = cross_add(cross_add(ld3(s.aj[l]), al, rc), w, cross(w, rc));
            st3(s.f[l], s.mdl.mass[l] * ac);
            v3 Iwv = mk3(Iw[0] * w.x + Iw[1] * w.y + Iw[2] * w.z, Iw[1] * w.x + Iw[3] * w.y + Iw[4] * w.z, Iw[2] * w.x + Iw[4] * w.y + Iw[5] * w.z);
            v3 Ial = mk3(Iw[0] * al.x + Iw[1] * al.y + Iw[2] * al.z, Iw[1] * al.x + Iw[3] * al.y + Iw[4] * al.z, Iw[2] * al.x + Iw[4] * al.y + Iw[5] * al.z);
            st3(s.n[l], cross_add(Ial, w, Iwv));
        }
    }
    // world axis of generalized velocity k and its moment about the root origin -> dofrec[k] = (a, g)
    // (root translation: a = 0, g = unit axis, so that one formula a.X + g.Y serves every dof)
    DM_DEV void dyn_dofrec(int k) {
        const int di = s.mdl.dof_info[k], dj = DM_DI_JOINT(di), kind = DM_DI_KIND(di), ax = DM_DI_AXIS(di);
        v3 rec_a, rec_g;
        if (kind == DK_ROOT_LIN) { rec_a = zero3(); rec_g = mk3((Real)(ax == 0), (Real)(ax == 1), (Real)(ax == 2)); }
        else {
            v3 a;
            if (kind == DK_ROOT_ANG) a = mk3((Real)(ax == 0), (Real)(ax == 1), (Real)(ax == 2));
            else { const int c = (kind == DK_REV) ? 2 : ax; a = mk3(s.R[dj][c], s.R[dj][3 + c], s.R[dj][6 + c]); }   // column c of R (indexed in LDS, not in registers)
            rec_a = a; rec_g = cross(ld3(s.p[dj]) - ld3(s.p[0]), a);
        }
        st3(&s.dofrec[k][0], rec_a); st3(&s.dofrec[k][3], rec_g);
    }
    // subtree sums about each joint's origin (descendants have larger ids): G adjacent lanes share one link,
    // sub-lane g takes members lk+g, lk+g+G, ...; the G partial sums are folded with DPP quad permutes
    DM_DEV void dyn_subtree() {
        const int J = m.J;
        constexpr int G = (NJ * 4 <= LW) ? 4 : 2;
        const int lk = l / G, g = l % G;
        v3 Fs = zero3(), Ns = Fs, h = Fs;
        Real mc = 0, Ic[6] = { 0, 0, 0, 0, 0, 0 };
        if constexpr (DM_STPREF != 0 && C::TREE && LW == kWave) {
            // (round 6, second pass; the dog) statically unrolled over the member slots, the next member's records requested before the current one is accumulated --
            // every lane reads (a lane without a member at this slot its own link), only the accumulation is predicated: the rolled loop below waited for each member's
            // reads in turn, ten times per call for the root.  Same members in the same order.
            constexpr int NIT = (NJ + G - 1) / G;
            const bool lane_on = lk < J;
            const int lkc = lane_on ? lk : 0;
            const uint32_t mask = s.mdl.subtree_mask[lkc];
            const v3 pj = ld3(s.p[lkc]);
            v3 cm[2], fk2[2], nk2[2]; Real mk2[2], iw2[2][6]; bool on2[2];
            auto stload = [&](auto ic) {
                constexpr int it = decltype(ic)::value, sl = it & 1;
                const int k = lk + g + G * it;
                const bool on = lane_on && k < J && ((mask >> (k < 32 ? k : 0)) & 1u);
                const int kc = on ? k : lkc;
                on2[sl] = on; cm[sl] = ld3(s.com[kc]); fk2[sl] = ld3(s.f[kc]); nk2[sl] = ld3(s.n[kc]); mk2[sl] = s.mdl.mass[kc];
#pragma unroll
                for (int q = 0; q < 6; ++q) iw2[sl][q] = s.Iw[kc][q];
            };
            stload(std::integral_constant<int, 0>{});
            static_for<0, NIT>([&](auto ic) {
                constexpr int it = decltype(ic)::value, sl = it & 1;
                if (G * it < J) {                                      // (wave-uniform: no slot past the last link)
                    if constexpr (it + 1 < NIT) stload(std::integral_constant<int, (it + 1 < NIT ? it + 1 : 0)>{});
                    if (on2[sl]) {
                        const v3 d = cm[sl] - pj, fk = fk2[sl];
                        Fs = Fs + fk; Ns = Ns + nk2[sl] + cross(d, fk);
                        const Real mk = mk2[sl], dd = dot(d, d);
                        mc += mk; h = h + mk * d;
                        Ic[0] += iw2[sl][0] + mk * (dd - d.x * d.x); Ic[1] += iw2[sl][1] - mk * d.x * d.y; Ic[2] += iw2[sl][2] - mk * d.x * d.z;
                        Ic[3] += iw2[sl][3] + mk * (dd - d.y * d.y); Ic[4] += iw2[sl][4] - mk * d.y * d.z; Ic[5] += iw2[sl][5] + mk * (dd - d.z * d.z);
                    }
                }
            });
        } else
        if (lk < J) {
            const uint32_t mask = s.mdl.subtree_mask[lk];
            const v3 pj = ld3(s.p[lk]);
            for (int k = lk + g; k < J; k += G) {
                if (!((mask >> k) & 1u)) continue;
                v3 d = ld3(s.com[k]) - pj, fk = ld3(s.f[k]);
                Fs = Fs + fk; Ns = Ns + ld3(s.n[k]) + cross(d, fk);
                Real mk = s.mdl.mass[k], dd = dot(d, d);
                mc += mk; h = h + mk * d;
                Ic[0] += s.Iw[k][0] + mk * (dd - d.x * d.x); Ic[1] += s.Iw[k][1] - mk * d.x * d.y; Ic[2] += s.Iw[k][2] - mk * d.x * d.z;
                Ic[3] += s.Iw[k][3] + mk * (dd - d.y * d.y); Ic[4] += s.Iw[k][4] - mk * d.y * d.z; Ic[5] += s.Iw[k][5] + mk * (dd - d.z * d.z);
            }
        }
        Real acc[16] = { Fs.x, Fs.y, Fs.z, Ns.x, Ns.y, Ns.z, mc, h.x, h.y, h.z, Ic[0], Ic[1], Ic[2], Ic[3], Ic[4], Ic[5] };
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] += wave_shfl_xor_c<1>(acc[i]); if (G == 4) acc[i] += wave_shfl_xor_c<2>(acc[i]); }
        if (lk < J && g == 0) {
            for (int i = 0; i < 3; ++i) { s.Fs[lk][i] = acc[i]; s.Ns[lk][i] = acc[3 + i]; }
            for (int i = 0; i < 10; ++i) s.Ic[lk][i] = acc[6 + i];
        }
    }
    // bias force C_k and row k of H (lower triangle, into the packed store): zero, then the ancestor-or-self dofs j <= k:
    // H_kj = a_j . Lq + g_j . Pm with the composite-body momentum (Pm, Lq about the root origin) of dof k's unit velocity
    DM_DEV void dyn_row(int k, Real diag_scale) {
        const int di = s.mdl.dof_info[k], dj = DM_DI_JOINT(di), kind = DM_DI_KIND(di), ax = DM_DI_AXIS(di);
        const v3 a = (kind == DK_ROOT_LIN) ? ld3(&s.dofrec[k][3]) : ld3(&s.dofrec[k][0]);
        s.dofrec[k][7] = (kind == DK_ROOT_LIN) ? s.Fs[0][ax] : dot(a, ld3(s.Ns[dj]));
        const Real* ic = s.Ic[dj];
        v3 h = mk3(ic[1], ic[2], ic[3]), Pm, Lp;
        if (kind == DK_ROOT_LIN) { Pm = ic[0] * a; Lp = cross(h, a); }
        else {
            Pm = cross(a, h);
            Lp = mk3(ic[4] * a.x + ic[5] * a.y + ic[6] * a.z, ic[5] * a.x + ic[7] * a.y + ic[8] * a.z, ic[6] * a.x + ic[8] * a.y + ic[9] * a.z);
        }
        v3 Lq = cross_add(Lp, ld3(s.p[dj]) - ld3(s.p[0]), Pm);
        Real* row = &s.Lt[L::lrow(k)];
        // ancestor-or-self dofs j <= k: the chain of the dof's joint, cut at k.  The loop runs over ALL dofs with static j: every
        // lane reads the same record dofrec[j] (one conflict-free LDS broadcast, requested ahead by the unrolled schedule) and keeps
        // the value when j is on its chain -- against a per-lane walk over the set bits, whose record gathers collide on the LDS banks
        // and whose trip count is the longest chain of the wave.
        const uint32_t lo = s.mdl.chain_lo[dj] & ((k < 31) ? ((2u << k) - 1u) : ~0u), hi = (k < 32) ? 0u : (s.mdl.chain_hi[dj] & ((k < 63) ? ((2u << (k - 32)) - 1u) : ~0u));
        const Real dk = diag_scale * s.mdl.kd[dj];
        if constexpr (DM_DRPREF != 0 && LW == 32) {      // (two characters per wavefront: 256 registers; the one-per-wave biped kernels run at 128)
        // (round 6) the records of pair p + 1 are requested before the dot products of pair p, and the pair is computed by every lane (only the store is
        // predicated): with the arithmetic sunk under the lanes' `2 p <= k` test -- what the optimizer made of the loop below -- every pair waited for
        // its own three reads behind a branch.  Same values.
        R4 da[2][2]; R2 db[2][2];
        auto drload = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            static_for<0, 2>([&](auto cc) { constexpr int c = decltype(cc)::value, j = 2 * p + c;
                if constexpr (j < ND) { da[p & 1][c] = *reinterpret_cast<const R4*>(&s.dofrec[j][0]); db[p & 1][c] = *reinterpret_cast<const R2*>(&s.dofrec[j][4]); } });
        };
        drload(std::integral_constant<int, 0>{});
        static_for<0, NP2>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            if constexpr (p + 1 < NP2) drload(std::integral_constant<int, (p + 1 < NP2 ? p + 1 : 0)>{});
            R2 v2;
            static_for<0, 2>([&](auto cc) {
                constexpr int c = decltype(cc)::value, j = 2 * p + c;
                Real v = 0;
                if constexpr (j < ND) {
                    const R4 r0 = da[p & 1][c]; const R2 r1 = db[p & 1][c];
                    v = r0[0] * Lq.x + r0[1] * Lq.y + r0[2] * Lq.z + r0[3] * Pm.x + r1[0] * Pm.y + r1[1] * Pm.z;
                    if (j == k) v += dk;
                    const bool on = (((j < 32) ? lo : hi) >> (j & 31)) & 1u;
                    v = on ? v : (Real)0;
                    DM_OPAQUE_V(v);
                }
                v2[c] = v;
            });
            if (2 * p <= k) *reinterpret_cast<R2*>(&row[2 * p]) = v2;
            if (DM_DRPREF >= 2) DM_SCHED_FENCE();
        });
        } else {
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            R2 v2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = 2 * p + c;
                Real v = 0;
                if (j < ND) {
                    const R4 r0 = *reinterpret_cast<const R4*>(&s.dofrec[j][0]);
                    const R2 r1 = *reinterpret_cast<const R2*>(&s.dofrec[j][4]);
                    v = r0[0] * Lq.x + r0[1] * Lq.y + r0[2] * Lq.z + r0[3] * Pm.x + r1[0] * Pm.y + r1[1] * Pm.z;
                    if (j == k) v += dk;
                    const bool on = (((j < 32) ? lo : hi) >> (j & 31)) & 1u;
                    v = on ? v : (Real)0;
                }
                v2[c] = v;
            }
            if (2 * p <= k) *reinterpret_cast<R2*>(&row[2 * p]) = v2;
        }
        }
    }

    // TREE classes: bias force C_k and the composite-body momentum record (Lq, Pm, diagonal term) of dof k's unit velocity, published
    // in the (dead) factor storage; the lanes then build their COLUMN of H from the records straight into registers (tree_load_col)
    DM_DEV void dyn_mom(int k, Real diag_scale) {
        const int di = s.mdl.dof_info[k], dj = DM_DI_JOINT(di), kind = DM_DI_KIND(di), ax = DM_DI_AXIS(di);
        const v3 a = (kind == DK_ROOT_LIN) ? ld3(&s.dofrec[k][3]) : ld3(&s.dofrec[k][0]);
        s.dofrec[k][7] = (kind == DK_ROOT_LIN) ? s.Fs[0][ax] : dot(a, ld3(s.Ns[dj]));
        const Real* ic = s.Ic[dj];
        v3 h = mk3(ic[1], ic[2], ic[3]), Pm, Lp;
        if (kind == DK_ROOT_LIN) { Pm = ic[0] * a; Lp = cross(h, a); }
        else {
            Pm = cross(a, h);
            Lp = mk3(ic[4] * a.x + ic[5] * a.y + ic[6] * a.z, ic[5] * a.x + ic[7] * a.y + ic[8] * a.z, ic[6] * a.x + ic[8] * a.y + ic[9] * a.z);
        }
        v3 Lq = cross_add(Lp, ld3(s.p[dj]) - ld3(s.p[0]), Pm);
        Real* mr = &s.Lt[k * 8];
        st3(mr, Lq); st3(mr + 3, Pm); mr[6] = diag_scale * s.mdl.kd[dj];
    }
    // column l of H (lower triangle) from the momentum records: H_il = a_l . Lq_i + g_l . Pm_i for the descendants i of dof l
    // (wave-uniform broadcast reads of record i, static i), 0 elsewhere; the diagonal apart
    DM_DEV void tree_load_col(typename VecT<Real>::v2 (&c2)[ND / 2], Real& hd) {
        typedef typename C::Topo TP;
        const int lr = l < ND ? l : 0;
        uint64_t z = 0; DM_OPAQUE_S(z);
        const R4 q0 = *reinterpret_cast<const R4*>(&s.dofrec[lr][0]);
        const R2 q1 = *reinterpret_cast<const R2*>(&s.dofrec[lr][4]);
#if DM_LCPREF
        // (round 6) groups of DM_LCPREF records, the next group's reads requested before the current group's dot products: one exposed LDS round
        // trip for the whole column instead of one per fenced group
        constexpr int G = DM_LCPREF, NG = (ND + G - 1) / G;        // (the last group may be short: ND = 34)
        R4 ra[2][G]; R2 rb[2][G];
        auto grp_load = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            static_for<0, G>([&](auto jc) { constexpr int j = decltype(jc)::value, i = g * G + j;
                if constexpr (i < ND) { ra[g & 1][j] = *reinterpret_cast<const R4*>(&s.Lt[i * 8]); rb[g & 1][j] = *reinterpret_cast<const R2*>(&s.Lt[i * 8 + 4]); } });
        };
        grp_load(std::integral_constant<int, 0>{});
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < NG) { grp_load(std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{}); DM_SCHED_FENCE(); }
            static_for<0, G>([&](auto jc) { constexpr int j = decltype(jc)::value, i = g * G + j;
                if constexpr (i < ND) {
                    const R4 r0 = ra[g & 1][j]; const R2 r1 = rb[g & 1][j];
                    Real v = q0[0] * r0[0] + q0[1] * r0[1] + q0[2] * r0[2] + q0[3] * r0[3] + q1[0] * r1[0] + q1[1] * r1[1];
                    DM_OPAQUE_V(v);
                    c2[i >> 1][i & 1] = lane_sel<TP::T.anc[i < ND ? i : 0]>(v, (Real)0, l, z);
                }
            });
            DM_SCHED_FENCE();
        });
#else
        static_for<0, ND>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const R4 r0 = *reinterpret_cast<const R4*>(&s.Lt[i * 8]);
            const R2 r1 = *reinterpret_cast<const R2*>(&s.Lt[i * 8 + 4]);
            Real v = q0[0] * r0[0] + q0[1] * r0[1] + q0[2] * r0[2] + q0[3] * r0[3] + q1[0] * r1[0] + q1[1] * r1[1];
            DM_OPAQUE_V(v);            // evaluated where it stands: the optimizer otherwise sinks all 64 dot products below the loads (and spills the records)
            c2[i >> 1][i & 1] = lane_sel<TP::T.anc[i]>(v, (Real)0, l, z);       // dof i is a descendant of exactly the lanes anc(i)
            if ((i & 7) == 7) DM_SCHED_FENCE();        // keeps the 64 record reads from being issued (and kept live) all at once
        });
#endif
        const Real* mo = &s.Lt[lr * 8];
        hd = (l < ND) ? q0[0] * mo[0] + q0[1] * mo[1] + q0[2] * mo[2] + q0[3] * mo[3] + q1[0] * mo[4] + q1[1] * mo[5] + mo[6] : (Real)1;
    }
    // ------------------------------------------------------------------ dense SPD linear algebra, register resident
    // Lane i owns row i of H (lower triangle in s.L).  The factorisation and the triangular solves run entirely in
    // VGPRs with v_readlane broadcasts (no LDS round trips, no barriers); the factor is written back to LDS (diagonal
    // slot = 1/L_kk) for the per-row forward substitutions of the constraint solve and the column reads of the
    // backward substitution.  All loops are fully unrolled (ND is the compile-time dof count of the kernel class), so
    // every register-array index is static.
    //
    // Factor H = L L^T and x := H^-1 x for the LDS vector x (length D).  The row lives in NP2 packed register pairs
    // so that the rank-1 updates issue as v_pk_fma_f32 with the two broadcast factors in one SGPR pair.
    DM_DEV void chol_solve(Real* xvec) {
        const int D = m.D;
        R2 h2[NP2];
        const int lr = l < ND ? l : 0;
        // own row; entries right of the diagonal are never consumed (they only feed this lane's own dead entries)
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            R2 v = *reinterpret_cast<const R2*>(&s.Lt[L::lrow(lr) + 2 * p]);
            if (!(l < D)) { v[0] = (l == 2 * p) ? (Real)1 : (Real)0; v[1] = (l == 2 * p + 1) ? (Real)1 : (Real)0; }
            h2[p] = v;
        }
        // Column k of L (one value per lane) is published through LDS and read back as wave-uniform broadcasts: two
        // ds_read_b128 feed four packed FMAs, where v_readlane would cost one VALU slot per scalar.  The buffer aliases the
        // Newton-Euler accumulators (dead once the mass matrix is built) and is double-buffered over k, so the in-order LDS
        // queue of the single wave is all the synchronisation needed.
        Real* colbuf = &s.f[0][0];
        Real dinv = 1;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int pk = k >> 1, ck = k & 1;
            Real hk = h2[pk][ck];
            Real piv = lane_bcast(hk, k);
            Real inv = dm_rsqrt(piv);
            Real lik = hk * inv;
            h2[pk][ck] = lik;
            if (l == k) dinv = inv;
            if (k + 1 < ND) {
                Real* cb = colbuf + (k & 1) * kWave;
                cb[l] = lik;
                sync();
                if (ck == 0) h2[pk][1] -= lik * cb[k + 1];
                const R2 l2 = {lik, lik};
#pragma unroll
                for (int p = (k + 2) >> 1; p < NP2; ++p) h2[p] -= l2 * *reinterpret_cast<const R2*>(&cb[2 * p]);
            }
        }
        if (l < ND) {
            Real* row = &s.Lt[L::lrow(l)];
#pragma unroll
            for (int p = 0; p < NP2; ++p) {
                R2 v = h2[p];
                if (l == 2 * p) v[0] = dinv;
                if (l == 2 * p + 1) v[1] = dinv;
                if (2 * p <= l) *reinterpret_cast<R2*>(&row[2 * p]) = v;
            }
        }
        Real x = (l < D) ? xvec[l] : (Real)0;
        // forward substitution: the entries of the own row on and right of the diagonal are zeroed once, so a step is mul, broadcast,
        // FMA for every lane (no per-step lane compares); the lane's own scaling by 1/L_kk moves behind the loop
#pragma unroll
        for (int p = 0; p < NP2; ++p) { if (!(2 * p < l)) h2[p][0] = 0; if (!(2 * p + 1 < l)) h2[p][1] = 0; }
#pragma unroll
        for (int k = 0; k < ND; ++k) { const Real xk = lane_bcast(x * dinv, k); x -= h2[k >> 1][k & 1] * xk; }
        x *= dinv;
        sync();
        x = back_substitute(x, dinv);
        if (l < D) xvec[l] = x;
        sync();
    }
    // x_l := (L^-T x)_l with column l of L read back from LDS
    DM_DEV Real back_substitute(Real x, Real dinv) {
        Real c[ND];
        const int lr = l < ND ? l : 0;
#pragma unroll
        for (int k = 0; k < ND; ++k) c[k] = (l < ND && k > l) ? s.Lt[L::lrow(k) + lr] : (Real)0;
#pragma unroll
        for (int k = ND - 1; k >= 0; --k) { const Real xk = lane_bcast(x * dinv, k); x -= c[k] * xk; }     // c[k] = 0 for k <= l
        return x * dinv;
    }

    // ------------------------------------------------------------------ branch-sparse factor H = L^T L on a compiled topology (TREE classes)
    // Replaces chol_solve / back_substitute above for a class whose dof tree is a compile-time table (dm_types.h TopoTables; dense
    // counterpart in the reference: the Eigen LDLT of sim/ImpPDController.cpp:162-188).  Lane j owns COLUMN j of the lower triangle:
    // c(i) = H_ij, i > j, in a statically indexed register array, the diagonal in its own register.  Elimination runs leaves first,
    // one LEVEL of the tree per step: every lane turns its entries of the level's pivot rows into L_kj = H_kj / L_kk (the pivots'
    // 1 / L_kk come by v_readlane from the pivot lanes, which all take their rsqrt in the same instruction), publishes them through LDS
    // (one value per lane and pivot) and subtracts L_kj L_ki for the ancestors i of each pivot k -- a static list, read back as wave-uniform
    // broadcasts.  Lanes that are no ancestor of k carry L_kj = 0 and do harmless work; a lane's entries above its own diagonal collect
    // garbage that only ever feeds that lane's own dead entries (pivot k is processed before any of its ancestors).  22 dependent steps
    // and 388 packed FMAs for dog3d against 64 columns and 1024 of the dense code.  Also solves x := H^-1 x for the LDS vector xvec.
    DM_DEV void tree_solve(Real* xvec) {
        typedef typename C::Topo TP;
        static_assert(TP::N == ND, "topology table and kernel class disagree");
        R2 c2[NP2]; Real hd;
        tree_load_col(c2, hd);
        sync();
        uint64_t z = 0; DM_OPAQUE_S(z);
        Real dinv = 1;
        tree_elim<0>(c2, hd, dinv, z);
        // the factor goes to LDS by columns for the row lanes' Y = L^-T J^T and the row reads of tree_fwd: quads from the own diagonal
        // down (what sits above the diagonal in the first quad is never read), then 1 / L_ll into the diagonal slot
        if (l < ND) {
            Real* col = &s.Lt[L::lcb(l)];
#pragma unroll
            for (int t = 0; t < (NP2 + 1) / 2; ++t) {        // (ND = 4 q + 2: the last quad's upper pair does not exist and is stored as zeros)
                const R2 up = (2 * t + 1 < NP2) ? c2[(2 * t + 1 < NP2) ? 2 * t + 1 : 0] : R2{(Real)0, (Real)0};
                const R4 w = {c2[2 * t][0], c2[2 * t][1], up[0], up[1]};
                if ((TP::T.quadmask[t] >> l) & 1ull) *reinterpret_cast<R4*>(&col[4 * t]) = w;
            }
            col[l] = dinv;
        }
        // entries on and above the diagonal are zeroed once: the substitution below then runs without lane compares
        static_for<0, ND>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            c2[i >> 1][i & 1] = lane_sel<((1ull << i) - 1ull)>(c2[i >> 1][i & 1], (Real)0, l, z);      // lanes below i keep their L_il
        });
        // L^T z = x, deepest level first: z_j = (x_j - sum_{i in desc(j)} L_ij z_i) / L_jj
        Real x = (l < ND) ? xvec[l] : (Real)0;
        tree_bwd<0>(c2, x, dinv);
        x *= dinv;
        sync();
        x = tree_fwd(x, dinv);
        if (l < ND) xvec[l] = x;
        sync();
    }
    // level V of the schedule (0 = the deepest): every loop bound below is a template constant, so all register indices are immediates
    template <int V> DM_DEV void tree_elim(R2 (&c2)[NP2], Real& hd, Real& dinv, uint64_t z) {
        typedef typename C::Topo TP;
        if constexpr (V < TP::T.nlev) {
            constexpr int S0 = TP::T.lev_start[V], W = TP::T.lev_start[V + 1] - S0;
            // [pivot of the level][dof]: aliases the dead Newton-Euler accumulators f, n, Iw, Fs, Ns -- NOT Ic behind them, where the
            // stable-PD joint forces live during the SPD solve (xs() = Ic); the stride is the dof count rounded to a quad
            constexpr int CS = (ND + 3) & ~3;
            static_assert(TP::T.maxw * CS <= 18 * NJ, "level buffer must end before Ic (= xs during the SPD solve): f | n | Iw | Fs | Ns are dead by now");
            Real* cbuf = &s.f[0][0];
            const Real inv = dm_rsqrt(hd);
            dinv = lane_sel<TP::T.levmask[V]>(inv, dinv, l, z);
            Real lk[W];
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int k = TP::T.order[S0 + q];
                const Real ik = lane_bcast(inv, k);
                const Real x = c2[k >> 1][k & 1] * ik;
                c2[k >> 1][k & 1] = x; lk[q] = x;
                if (TP::T.anc[k] != 0 && (ND >= kWave || l < ND)) cbuf[q * CS + l] = x;       // (no lane test when every lane is a dof: a divergent store here makes the optimizer sink the level's arithmetic behind it)
            }
            sync();
#if DM_ELPREF
            // (round 6) the published entries of pivot q + 1 are requested before the rank-1 update of pivot q runs (two register sets by the parity of q): left to
            // itself the scheduler keeps two or three reads in flight, a third of what the LDS latency needs.  Same updates in the same order per entry.
            constexpr int NQ = (NP2 + 1) / 2;
            R2 ea[2][NQ], eb[2][NQ]; Real ek[2] = {(Real)0, (Real)0};
            auto eload = [&](auto qc) {
                constexpr int q = decltype(qc)::value, k = TP::T.order[S0 + q], sl = q & 1;
                if constexpr (TP::T.anc[k] != 0) {
                    static_for<0, NQ>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        constexpr bool own0 = (2 * t == (k >> 1)), own1 = (2 * t + 1 == (k >> 1));
                        constexpr bool on0 = !own0 && ((TP::T.anc[k] >> (4 * t)) & 3ull), on1 = (2 * t + 1 < NP2) && !own1 && ((TP::T.anc[k] >> (4 * t + 2)) & 3ull);
                        if constexpr (on0 && on1) { const R4 r = *reinterpret_cast<const R4*>(&cbuf[q * CS + 4 * t]); ea[sl][t] = R2{r[0], r[1]}; eb[sl][t] = R2{r[2], r[3]}; }
                        else if constexpr (on0) ea[sl][t] = *reinterpret_cast<const R2*>(&cbuf[q * CS + 4 * t]);
                        else if constexpr (on1) eb[sl][t] = *reinterpret_cast<const R2*>(&cbuf[q * CS + 4 * t + 2]);
                        if constexpr ((own0 || own1) && (k & 1) && ((TP::T.anc[k] >> (k - 1)) & 1ull)) ek[sl] = cbuf[q * CS + k - 1];
                    });
                }
            };
            eload(std::integral_constant<int, 0>{});
            static_for<0, W>([&](auto qc) {
                constexpr int q = decltype(qc)::value, k = TP::T.order[S0 + q], sl = q & 1;
                if constexpr (q + 1 < W) { eload(std::integral_constant<int, (q + 1 < W ? q + 1 : 0)>{}); DM_SCHED_FENCE(); }
                if constexpr (TP::T.anc[k] != 0) {
                    hd -= lk[q] * lk[q];
                    const R2 l2 = {lk[q], lk[q]};
                    static_for<0, NQ>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        constexpr bool own0 = (2 * t == (k >> 1)), own1 = (2 * t + 1 == (k >> 1));
                        constexpr bool on0 = !own0 && ((TP::T.anc[k] >> (4 * t)) & 3ull), on1 = (2 * t + 1 < NP2) && !own1 && ((TP::T.anc[k] >> (4 * t + 2)) & 3ull);
                        if constexpr (on0) c2[2 * t] -= l2 * ea[sl][t];
                        if constexpr (on1) c2[(2 * t + 1 < NP2) ? 2 * t + 1 : 0] -= l2 * eb[sl][t];
                        if constexpr ((own0 || own1) && (k & 1) && ((TP::T.anc[k] >> (k - 1)) & 1ull)) c2[k >> 1][0] -= lk[q] * ek[sl];
                    });
                }
                if constexpr (q + 1 < W) DM_SCHED_FENCE();
            });
#else
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int k = TP::T.order[S0 + q];
                if (TP::T.anc[k] == 0) continue;
                hd -= lk[q] * lk[q];
                const R2 l2 = {lk[q], lk[q]};
                // (the pair that holds k itself takes its lower entry alone: lane k's slot k -- the diagonal position, kept apart in hd --
                // has collected - sum L^2 from deeper pivots, and what lane k published for it must not reach the fresh L_kj of the others)
#pragma unroll
                for (int t = 0; t < (NP2 + 1) / 2; ++t) {   // quads of dofs 4t .. 4t + 3 = pairs 2t, 2t + 1: one 16-B broadcast read when both take part
                    const bool own0 = (2 * t == (k >> 1)), own1 = (2 * t + 1 == (k >> 1));
                    const bool on0 = !own0 && ((TP::T.anc[k] >> (4 * t)) & 3ull), on1 = (2 * t + 1 < NP2) && !own1 && ((TP::T.anc[k] >> (4 * t + 2)) & 3ull);
                    if (on0 && on1) {
                        const R4 r = *reinterpret_cast<const R4*>(&cbuf[q * CS + 4 * t]);
                        const R2 ra = {r[0], r[1]}, rb = {r[2], r[3]};
                        c2[2 * t] -= l2 * ra; c2[(2 * t + 1 < NP2) ? 2 * t + 1 : 0] -= l2 * rb;
                    } else if (on0) c2[2 * t] -= l2 * *reinterpret_cast<const R2*>(&cbuf[q * CS + 4 * t]);
                    else if (on1) c2[(2 * t + 1 < NP2) ? 2 * t + 1 : 0] -= l2 * *reinterpret_cast<const R2*>(&cbuf[q * CS + 4 * t + 2]);
                    if ((own0 || own1) && (k & 1) && ((TP::T.anc[k] >> (k - 1)) & 1ull)) c2[k >> 1][0] -= lk[q] * cbuf[q * CS + k - 1];
                }
            }
#endif
            tree_elim<V + 1>(c2, hd, dinv, z);
        }
    }
    template <int V> DM_DEV void tree_bwd(const R2 (&c2)[NP2], Real& x, Real dinv) {
        typedef typename C::Topo TP;
        if constexpr (V < TP::T.nlev) {
            constexpr int S0 = TP::T.lev_start[V], W = TP::T.lev_start[V + 1] - S0;
            const Real xs = x * dinv;
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int k = TP::T.order[S0 + q];
                if (TP::T.anc[k] != 0) x -= c2[k >> 1][k & 1] * lane_bcast(xs, k);
            }
            tree_bwd<V + 1>(c2, x, dinv);
        }
    }
    // x_k := (L^-1 x)_k, root level first, with ROW k of L read back from the column store (per-lane reads; only the lanes below a
    // column's dof in the tree hold a nonzero there: a compile-time lane mask)
    DM_DEV Real tree_fwd(Real x, Real dinv) {
        typedef typename C::Topo TP;
        uint64_t z = 0; DM_OPAQUE_S(z);
        const int lr = l < ND ? l : 0;
        Real r[ND];
#if DM_TFPREF
        // (round 6) all row reads are requested first, the lane selects follow behind one wait: interleaved, each select (an opaque asm) waited for its own read
        static_for<0, ND>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (TP::T.desc[j] != 0) r[j] = s.Lt[L::lcb(j) + lr]; else r[j] = 0;
        });
        DM_SCHED_FENCE();
        static_for<0, ND>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (TP::T.desc[j] != 0) r[j] = lane_sel<TP::T.desc[j]>(r[j], (Real)0, l, z);
        });
#else
        static_for<0, ND>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (TP::T.desc[j] != 0) r[j] = lane_sel<TP::T.desc[j]>(s.Lt[L::lcb(j) + lr], (Real)0, l, z); else r[j] = 0;
        });
#endif
        tree_fwd_lev<TP::T.nlev - 1>(r, x, dinv);
        return x * dinv;
    }
    template <int V> DM_DEV void tree_fwd_lev(const Real (&r)[ND], Real& x, Real dinv) {
        typedef typename C::Topo TP;
        if constexpr (V >= 0) {
            constexpr int S0 = TP::T.lev_start[V], W = TP::T.lev_start[V + 1] - S0;
            const Real xs = x * dinv;
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int j = TP::T.order[S0 + q];
                if (TP::T.desc[j] != 0) x -= r[j] * lane_bcast(xs, j);
            }
            tree_fwd_lev<V - 1>(r, x, dinv);
        }
    }
    // one stage of the transposing wave reduction: N per-lane partial sums -> (N+1)/2, lanes split on bit MASK
    template <int N, int MASK> DM_DEV void tr_stage(Real (&w)[NP2]) {
        const bool bit = (l & MASK) != 0;
#pragma unroll
        for (int i = 0; i < (N + 1) / 2; ++i) {
            Real a = w[2 * i], b = (2 * i + 1 < N) ? w[2 * i + 1] : (Real)0;
            Real keep = bit ? b : a, send = bit ? a : b;
            w[i] = keep + wave_shfl_xor_c<MASK>(send);
        }
    }

    DM_DEV v3 gravity_a0() const { return mk3(-m.gravity[0], -m.gravity[1], -m.gravity[2]); }

    // ------------------------------------------------------------------ stable-PD torques
    // reference BuildCjRoot quirk == uniform extra base acceleration v0 x (E w - w)  
    DM_DEV v3 spd_a0() const {
        v3 v0 = ld3(s.vel), w0 = ld3(s.vel + 3);
        m3 E = quat_to_rot(ldq(s.pose + 3));
        return gravity_a0() + cross(v0, E * w0 - w0);
    }
    // pose error per joint -> xs = Kp e + Kd (0 - qd); rhs = xs - C
    DM_DEV void spd_rhs(Real dt) {
        spd_rhs_pre(dt);
        for (int i = l; i < m.D; i += LW) s.rhs[i] = xs()[i] - s.dofrec[i][7];
        sync();
    }
    DM_DEV void spd_rhs_pre(Real dt) {
        if (l < m.J && l > 0) {
            int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li), dof = DM_LI_DOFF(li);
            if (jt == JT_SPHERICAL) {
                q4 q = ldq(s.pose + off); v3 om = ld3(s.vel + off);
                q4 dq = quat_diff_mul(q, om);
                q4 qh = qnormalize(mkq(q.w + dt * dq.w, q.x + dt * dq.x, q.y + dt * dq.y, q.z + dt * dq.z));
                v3 e = quat_to_rotvec(qmul(qconj(qh), ldq(s.tar + off)), (Real)0.000001);
                for (int k = 0; k < 3; ++k) xs()[dof + k] = s.mdl.kp[l] * comp(e, k) + s.mdl.kd[l] * (-s.vel[off + k]);
            } else if (jt == JT_REVOLUTE) {
                Real th = normalize_angle(s.pose[off]);
                Real e = s.tar[off] - (th + dt * s.vel[off]);
                xs()[dof] = s.mdl.kp[l] * e + s.mdl.kd[l] * (-s.vel[off]);
            }
        }
        if (l < 6) xs()[l] = 0;
        sync();
    }
    // rhs holds qddot: tau = Kp e + Kd (e_v - dt qddot), clamped per joint (SimBodyJoint.cpp:299-307)
    DM_DEV void spd_post(Real dt) {
        for (int i = l; i < m.D; i += LW) s.tau[i] = (i < 6) ? (Real)0 : xs()[i] - s.mdl.kd[DM_DI_JOINT(s.mdl.dof_info[i])] * dt * s.rhs[i];
        sync();
        spd_clamp();
    }
    DM_DEV void spd_clamp() {
        if (l < m.J && l > 0) {
            int jt = DM_LI_JTYPE(li), dof = DM_LI_DOFF(li); Real lim = s.mdl.torque_lim[l];
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
    // ------------------------------------------------------------------ contacts (DM-physics v1)
    // One self-collision candidate: capsule models of links i and j (mdl.cap), closest points of the two segments (Ericson,
    // Real-Time Collision Detection 5.1.9).  Returns whether it is active; x = midpoint of the two surface points, n from j to i.
    // (round 6: split into the reads and the arithmetic, so that a caller can request the next pass's operands before it evaluates the current one; same operations in the same order)
    struct PairIn { R4 ci, cj; m3 Ri, Rj; v3 pi, pj; Real ti, tj; };
    DM_DEV PairIn pair_load(int i, int j) const {
        PairIn in;
        { const Real* ci = s.mdl.cap[i]; const Real* cj = s.mdl.cap[j]; in.ci = R4{ci[0], ci[1], ci[2], ci[3]}; in.cj = R4{cj[0], cj[1], cj[2], cj[3]}; }      // (the table is not 16-B aligned: element reads)
        in.Ri = ldm3(Rbp(i)); in.Rj = ldm3(Rbp(j));
        in.pi = ld3(s.com[i]); in.pj = ld3(s.com[j]);
        in.ti = s.mdl.thresh[i]; in.tj = s.mdl.thresh[j];
        return in;
    }
    DM_DEV bool pair_eval(const PairIn& in, v3& x, v3& n, Real& dist) const {
        const v3 ui = in.Ri * mk3(in.ci[0], in.ci[1], in.ci[2]), uj = in.Rj * mk3(in.cj[0], in.cj[1], in.cj[2]);
        const v3 p1 = in.pi + ui, p2 = in.pj + uj;
        const v3 d1 = (Real)-2 * ui, d2 = (Real)-2 * uj, r = p1 - p2;
        const Real eps = (Real)1e-12;
        const Real a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
        // reciprocals once (fp32: v_rcp_f32 + Newton step), clamps by v_med3
        const Real ia = (a > eps) ? dm_rcp(a) : (Real)0, ie = (e > eps) ? dm_rcp(e) : (Real)0;
        Real sp, tp;
        if (a <= eps && e <= eps) { sp = 0; tp = 0; }
        else if (a <= eps) { sp = 0; tp = dm_med3((Real)0, f * ie, (Real)1); }
        else {
            const Real c = dot(d1, r);
            if (e <= eps) { tp = 0; sp = dm_med3((Real)0, -c * ia, (Real)1); }
            else {
                const Real b = dot(d1, d2), denom = a * e - b * b;
                sp = (denom > eps) ? dm_med3((Real)0, (b * f - c * e) * dm_rcp(denom), (Real)1) : (Real)0;
                tp = (b * sp + f) * ie;
                if (tp < 0) { tp = 0; sp = dm_med3((Real)0, -c * ia, (Real)1); }
                else if (tp > 1) { tp = 1; sp = dm_med3((Real)0, (b - c) * ia, (Real)1); }
            }
        }
        const v3 ca = p1 + sp * d1, cb = p2 + tp * d2, dl = ca - cb;
        const Real d2n = dot(dl, dl), ri = in.ci[3], rj = in.cj[3];
        const Real idn = (d2n > (Real)1e-18) ? dm_rsqrt(d2n) : (Real)0, d = d2n * idn;
        dist = d - ri - rj;
        n = (d > (Real)1e-9) ? idn * dl : mk3((Real)0, (Real)1, (Real)0);
        x = (Real)0.5 * ((ca - ri * n) + (cb + rj * n));
        return dist < dm_min(in.ti, in.tj);
    }
    DM_DEV bool self_pair(int i, int j, v3& x, v3& n, Real& dist) const { return pair_eval(pair_load(i, j), x, n, dist); }
    // btPlaneSpace1: two tangents of a unit normal ((-1,0,0), (0,0,1) for the ground normal)
    static DM_DEV void plane_space(v3 n, v3& p, v3& q) {
        if (dm_abs(n.z) > (Real)0.7071067811865475244) {
            const Real a = n.y * n.y + n.z * n.z, k = (Real)1 / dm_sqrt(a);
            p = mk3((Real)0, -n.z * k, n.y * k); q = mk3(a * k, -n.x * p.z, n.x * p.y);
        } else {
            const Real a = n.x * n.x + n.y * n.y, k = (Real)1 / dm_sqrt(a);
            p = mk3(-n.y * k, n.x * k, (Real)0); q = mk3(-n.z * p.y, n.z * p.x, a * k);
        }
    }
    // constraint row `r` of this lane from the contact slots: bias b, chain masks (dofs moving the point with link a, minus
    // those moving it with link b: common ancestors cancel exactly), X = (x - p0) x d and d
    // link id 254 = the free body (OBJ classes), 255 = the ground.  ball_sg: +1 the point moves with the ball as body a, -1 as body b, 0 none
    DM_DEV void contact_row(int r, int NL, int nc, Real h, Real& brow, uint32_t& m_lo, uint32_t& m_hi, uint32_t& g_lo, uint32_t& g_hi, v3& xd, v3& dd, int* ball_sg = nullptr, v3* cx = nullptr) const {
        int slot, kindr;
        if (r < NL + nc) { slot = r - NL; kindr = 0; } else { int fi = r - NL - nc; slot = fi >> 1; kindr = 1 + (fi & 1); }
        const Real* ct = s.ct[slot];
        const int info = (int)ct[7];            // link ids are small integers, exact in fp32
        const int la = info & 0xff, lb = (info >> 8) & 0xff;
        const v3 n = ld3(ct + 3);
        v3 t1, t2; plane_space(n, t1, t2);
        dd = (kindr == 0) ? n : ((kindr == 1) ? t1 : t2);
        xd = cross(ld3(ct) - ld3(s.p[0]), dd);
        uint32_t a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;
        if (!C::OBJ || la != 254) { a_lo = s.mdl.chain_lo[la < NJ ? la : 0]; a_hi = s.mdl.chain_hi[la < NJ ? la : 0]; }
        if (lb != 255 && (!C::OBJ || lb != 254)) { b_lo = s.mdl.chain_lo[lb < NJ ? lb : 0]; b_hi = s.mdl.chain_hi[lb < NJ ? lb : 0]; }
        if (C::OBJ && ball_sg) { *ball_sg = (la == 254) ? 1 : ((lb == 254) ? -1 : 0); *cx = ld3(ct); }
        m_lo = a_lo ^ b_lo; m_hi = a_hi ^ b_hi; g_lo = b_lo & m_lo; g_hi = b_hi & m_hi;
        if (kindr == 0) { const Real dc = ct[6]; brow = (dc > 0) ? -dc / h : -m.erp * dc / h; }
    }
    DM_DEV void store_contact(int slot, v3 x, v3 n, Real dist, int la, int lb) {
        Real* ct = s.ct[slot];
        st3(ct, x); st3(ct + 3, n); ct[6] = dist; ct[7] = (Real)(la | (lb << 8));
    }

    // ---- DM-physics v2: link-vs-ground contacts through one persistent manifold per link ([EXT-BULLET, recalled] btPersistentManifold::
    // refreshContactPoints, btConvexPlaneCollisionAlgorithm::collideSingleContact, getCacheEntry / sortCachedPoints; restated in
    // oracle/orc_scene.h manifold_update).  lane = link; the manifolds live in HBM (EnvState::manif, L2-resident for the 40 substeps of
    // a call) -- v2 is the parity mode, not the fast path.  Returns the number of ground contact slots written (in (link, slot) order).
    DM_DEV int ground_manifolds(Real* manif) {
        const int J = m.J;
        int cnt = 0; Real lp[4][3], bxz[4][2], dist[4];
        Real* mfp = manif + (l < J ? l : 0) * MF_STRIDE;
        const bool has_body = l < J && s.mdl.thresh[l < J ? l : 0] > (Real)0;
        // every lane evaluates its candidate points first (lane = candidate): distance into cdistc for the links' argmin below.  The candidates of a
        // link are contiguous (build_host_model adds them link by link): each candidate lane reports its index to its link's [first, count) cell in
        // LDS, so the link lane scans its own <= 8 distances instead of all NC candidate records in global memory (round 4: v2 two-per-wave +7.8 %)
        int* crange = &s.csel[0];                      // 2 x 32 ints (the cap path below reuses csel afterwards)
        static_assert(NCAP >= 64 && NJ <= 32, "csel holds the per-link candidate ranges");
        if (l < 32) { crange[l] = 0x7fffffff; crange[32 + l] = 0; }
        sync();
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = l + kWave * q;
            if (c < m.NC) {
                const int link = cand_link[q];
                v3 x = ld3(s.com[link]) + ldm3(Rbp(link)) * mk3(cand_loc[q][0], cand_loc[q][1], cand_loc[q][2]);
                s.cdistc[c] = x.y - cand_rad[q];
                dm_atomic_min(&crange[link], c); dm_atomic_add(&crange[32 + link], 1);
            }
        }
        sync();
        const int cfirst = crange[l & 31], ccount = crange[32 + (l & 31)];
        sync();
        if (has_body) {
            const Real thr = s.mdl.thresh[l];
            const v3 com = ld3(s.com[l]); const m3 Rb = ldm3(Rbp(l));
            cnt = (int)mfp[0];
            for (int i = 0; i < 4; ++i) { for (int k = 0; k < 3; ++k) lp[i][k] = mfp[1 + i * MF_PT + k]; bxz[i][0] = mfp[1 + i * MF_PT + 3]; bxz[i][1] = mfp[1 + i * MF_PT + 4]; dist[i] = mfp[1 + i * MF_PT + 5]; }
            // refresh, last to first (a dropped point's successors move down, as std::vector::erase does in the oracle)
            for (int i = cnt - 1; i >= 0; --i) {
                const v3 xa = com + Rb * mk3(lp[i][0], lp[i][1], lp[i][2]);
                dist[i] = xa.y;
                const Real dx = bxz[i][0] - xa.x, dz = bxz[i][1] - xa.z;
                if (!(dist[i] <= thr) || dx * dx + dz * dz > thr * thr) {
                    for (int k = i; k + 1 < cnt; ++k) { for (int a = 0; a < 3; ++a) lp[k][a] = lp[k + 1][a]; bxz[k][0] = bxz[k + 1][0]; bxz[k][1] = bxz[k + 1][1]; dist[k] = dist[k + 1]; }
                    --cnt;
                }
            }
            // the new point: the deepest candidate of this link (its support point along -n), first on ties
            int best = -1; Real bd = 0;
            for (int c = cfirst; c < cfirst + ccount; ++c) { const Real d = s.cdistc[c]; if (best < 0 || d < bd) { best = c; bd = d; } }
            if (best >= 0 && bd < thr) {
                v3 x = com + Rb * mk3(m.cand_loc[best * 3], m.cand_loc[best * 3 + 1], m.cand_loc[best * 3 + 2]);
                x.y -= m.cand_rad[best];
                const v3 d0 = x - com;
                const v3 nl = mk3(Rb.m[0] * d0.x + Rb.m[3] * d0.y + Rb.m[6] * d0.z, Rb.m[1] * d0.x + Rb.m[4] * d0.y + Rb.m[7] * d0.z, Rb.m[2] * d0.x + Rb.m[5] * d0.y + Rb.m[8] * d0.z);   // Rb^T (x - com)
                int slot = -1; Real shortest = thr * thr;
                for (int i = 0; i < cnt; ++i) { const Real ex = lp[i][0] - nl.x, ey = lp[i][1] - nl.y, ez = lp[i][2] - nl.z, d2 = ex * ex + ey * ey + ez * ez; if (d2 < shortest) { shortest = d2; slot = i; } }
                if (slot < 0) {
                    if (cnt < 4) slot = cnt++;
                    else {      // sortCachedPoints: never the deepest; of the others the one that leaves the largest quadrilateral
                        int deepest = -1; Real maxpen = x.y;
                        for (int i = 0; i < 4; ++i) if (dist[i] < maxpen) { deepest = i; maxpen = dist[i]; }
                        Real res[4] = { 0, 0, 0, 0 };
                        const v3 p0 = mk3(lp[0][0], lp[0][1], lp[0][2]), p1 = mk3(lp[1][0], lp[1][1], lp[1][2]), p2 = mk3(lp[2][0], lp[2][1], lp[2][2]), p3 = mk3(lp[3][0], lp[3][1], lp[3][2]);
                        if (deepest != 0) { const v3 c = cross(nl - p1, p3 - p2); res[0] = dot(c, c); }
                        if (deepest != 1) { const v3 c = cross(nl - p0, p3 - p2); res[1] = dot(c, c); }
                        if (deepest != 2) { const v3 c = cross(nl - p0, p3 - p1); res[2] = dot(c, c); }
                        if (deepest != 3) { const v3 c = cross(nl - p0, p2 - p1); res[3] = dot(c, c); }
                        slot = 0;
                        for (int i = 1; i < 4; ++i) if (res[i] > res[slot]) slot = i;
                    }
                }
                lp[slot][0] = nl.x; lp[slot][1] = nl.y; lp[slot][2] = nl.z; bxz[slot][0] = x.x; bxz[slot][1] = x.z; dist[slot] = x.y;
            }
            mfp[0] = (Real)cnt;
            for (int i = 0; i < 4; ++i) { for (int k = 0; k < 3; ++k) mfp[1 + i * MF_PT + k] = lp[i][k]; mfp[1 + i * MF_PT + 3] = bxz[i][0]; mfp[1 + i * MF_PT + 4] = bxz[i][1]; mfp[1 + i * MF_PT + 5] = dist[i]; }
            for (int i = 0; i < cnt; ++i) if (dist[i] <= m.report_dist) dm_atomic_or(&s.flg[FLG_CONTACT], 1 << l);
        }
        // cap at max_contacts, deepest first (ties: (link, slot) order), as v1 does for its candidates
        sync();
        bool keep[4] = { false, false, false, false };
        for (int i = 0; i < 4; ++i) keep[i] = i < cnt;
        int total = 0;
        uint64_t km[4];
        for (int i = 0; i < 4; ++i) { km[i] = wave_ballot(keep[i]); total += dm_popc64(km[i]); }
        if (total > m.max_contacts) {
            for (int i = 0; i < 4; ++i) if (l < J) { s.csel[l * 4 + i] = keep[i] ? 1 : 0; s.cdistc[l * 4 + i] = dist[i]; }
            sync();
            for (int i = 0; i < 4; ++i) {
                int rank = 0;
                if (keep[i]) for (int k = 0; k < 4 * J; ++k) if (s.csel[k] && (s.cdistc[k] < dist[i] || (s.cdistc[k] == dist[i] && k < l * 4 + i))) ++rank;
                keep[i] = keep[i] && rank < m.max_contacts;
            }
            sync();
            total = 0;
            for (int i = 0; i < 4; ++i) { km[i] = wave_ballot(keep[i]); total += dm_popc64(km[i]); }
        }
        // slots in (link, slot) order
        const uint64_t ltm = (l == 0) ? 0ull : (~0ull >> (64 - l));
        int base = 0;
        for (int i = 0; i < 4; ++i) base += dm_popc64(km[i] & ltm);
        if (l < J) {
            const v3 com = ld3(s.com[l]); const m3 Rb = ldm3(Rbp(l));
            int k = 0;
            for (int i = 0; i < 4; ++i) if (keep[i]) { store_contact(base + k, com + Rb * mk3(lp[i][0], lp[i][1], lp[i][2]), mk3((Real)0, (Real)1, (Real)0), dist[i], l, 255); ++k; }
        }
        return total;
    }
    // a reset (and the state setter) starts every manifold empty: cWorld::Reset clears the broadphase pair cache (sim/World.cpp:81-90)
    DM_DEV void manif_clear(const EnvState<Real>& st, int e) {
        if (st.manif && l < m.J) st.manif[((size_t)e * m.J + l) * MF_STRIDE] = (Real)0;
    }
    // integrate positions with the new velocity (semi-implicit Euler, exponential map on rotations); lane = link
    DM_DEV void integrate(Real h) {
        if (l < m.J) {
            int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li);
            if (l == 0) {
                for (int k = 0; k < 3; ++k) s.pose[k] += h * s.vel[k];
                q4 q = qnormalize(qmul(quat_exp(h * ld3(s.vel + 3)), ldq(s.pose + 3)));
                stq(s.pose + 3, q);
            } else if (jt == JT_SPHERICAL) {
                q4 q = qnormalize(qmul(ldq(s.pose + off), quat_exp(h * ld3(s.vel + off))));
                stq(s.pose + off, qstandardize(q));
            } else if (jt == JT_REVOLUTE) s.pose[off] += h * s.vel[off];
        }
    }
    // s.rhs holds qddot of the unconstrained dynamics; s.L the Cholesky factor of H.
    // PLAIN: the tap-free imitate instantiation without perturbations / manifolds; a class may give it a wider row file (C::RREG_PLAIN:
    // the registers the AMP / v2 code needs elsewhere are free there)
    // TREE classes: the number of 16-B quads of column k of the factor that hold a descendant of dof k (what the row lanes' substitution reads of it)
    static constexpr int y_nquads(int k) {
        if constexpr (C::TREE) {
            typedef typename C::Topo TP;
            int n = 0;
            for (int t = (k >> 2); t < (NP2 + 1) / 2; ++t) {
                const bool a = (2 * t > (k >> 1)) && ((TP::T.desc[k] >> (4 * t)) & 3ull), b = (2 * t + 1 < NP2) && (2 * t + 1 > (k >> 1)) && ((TP::T.desc[k] >> (4 * t + 2)) & 3ull);
                n += (a || b) ? 1 : 0;
            }
            return n;
        } else return 0;
    }
    template <bool V2 = false, bool PLAIN = false>
    // nc_ground_in >= 0 (V2, the fallback of the two-per-wave kernel): the caller has already updated the manifolds of this substep and stored the ground
    // contact slots in s.ct -- a second refresh would not be idempotent
    DM_DEV void substep_post(Real h, DebugTaps<Real> dbg, int e, Real* aovf, Real* manif = nullptr, int nc_ground_in = -1) {
        constexpr int RREG = PLAIN ? C::RREG_PLAIN : C::RREG;
        static_assert(RREG >= C::RREG, "the overflow block is sized for C::RREG");
        const int D = m.D, J = m.J;
        Real vstar = 0; int vidx = 0;
        if (l < D) { vidx = DM_DI_VIDX(s.mdl.dof_info[l]); vstar = clamp_vel(s.vel[vidx] + h * s.rhs[l], l); s.dofrec[l][6] = vstar; }
        const bool ground_done = V2 && nc_ground_in >= 0;
        if (l == 0 && !ground_done) s.flg[FLG_CONTACT] = 0;
        sync();
        if (TAPS && dbg.vstar && l < D) dbg.vstar[(size_t)e * D + l] = vstar;
        mark(7);
        // OBJ classes: the free body's unconstrained velocity (btRigidBody::applyDamping, then the gravity impulse), every lane
        v3 bpos = zero3(), bvs = zero3(), bws = zero3();
        if (C::OBJ) {
            const Real* ob = s.obj;
            bpos = ld3(ob + (C::OBJ ? OB_PX : 0));
            bvs = (Real)exp((double)h * m.ball_ln_lin) * ld3(ob + (C::OBJ ? OB_VX : 0)) + h * mk3(m.gravity[0], m.gravity[1], m.gravity[2]);
            bws = (Real)exp((double)h * m.ball_ln_ang) * ld3(ob + (C::OBJ ? OB_WX : 0));
        }
        // ---- collision detection
        const bool phys2 = V2 && manif != nullptr;
        int nc_ground = 0;
        if (ground_done) nc_ground = nc_ground_in;
        else if (phys2) nc_ground = ground_manifolds(manif);
        else {
        // ---- collision detection: lane = candidate point (CPL candidates per lane when NC > 64)
        bool active[CPL]; Real dist[CPL]; v3 cxp[CPL]; uint64_t amask[CPL];
        int nact = 0;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = l + kWave * q;
            active[q] = false; dist[q] = 0; cxp[q] = zero3();
            if (c < m.NC) {
                int link = cand_link[q];
                v3 x = ld3(s.com[link]) + ldm3(Rbp(link)) * mk3(cand_loc[q][0], cand_loc[q][1], cand_loc[q][2]);
                x.y -= cand_rad[q];
                dist[q] = x.y; cxp[q] = x;
                active[q] = x.y < s.mdl.thresh[link];
                if (x.y <= m.report_dist) dm_atomic_or(&s.flg[FLG_CONTACT], 1 << link);
            }
            amask[q] = wave_ballot(active[q]);
            nact += dm_popc64(amask[q]);
        }
        if (nact > m.max_contacts) {
            // manifold reduction (rare): keep the max_contacts deepest, ties to the lower index
#pragma unroll
            for (int q = 0; q < CPL; ++q) { const int c = l + kWave * q; s.csel[c] = active[q] ? 1 : 0; s.cdistc[c] = dist[q]; }
            sync();
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int c = l + kWave * q; int rank = 0;
                if (active[q]) for (int k = 0; k < m.NC; ++k) if (s.csel[k] && (s.cdistc[k] < dist[q] || (s.cdistc[k] == dist[q] && k < c))) ++rank;
                active[q] = active[q] && rank < m.max_contacts;
            }
            sync();
            nact = 0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { amask[q] = wave_ballot(active[q]); nact += dm_popc64(amask[q]); }
        }
#ifdef DM_PROF_COLLISION
        mark(13);
#endif
        const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
        {   // ground contacts -> slots, in candidate-index order
            int base = 0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                if (active[q]) store_contact(base + dm_popc64(amask[q] & lt), cxp[q], mk3((Real)0, (Real)1, (Real)0), dist[q], cand_link[q], 255);
                base += dm_popc64(amask[q]);
            }
        }
        nc_ground = nact;
        }
        const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
        int nc = V2 ? lane_bcast(nc_ground, 0) : nc_ground;       // (wave-uniform by construction; in the V2 instantiations the read-lane tells the compiler)
        // ---- self collision: lane = link pair; active pairs take the slots the ground left, in pair order
        if constexpr (C::BROAD) {
            // bounding-sphere cull first (a capsule is symmetric about its link's COM): the pairs that survive it -- a handful of the dog's
            // 231 in a mocap pose -- are compacted in pair order into a list and go through ONE pass of the segment-segment test instead
            // of four.  Same contacts in the same order as the full sweep below.
            int* plist = &s.csel[0];                   // NPAIRCAP ints over csel | cdistc (manifold-reduction scratch, free by now)
            static_assert(C::NPAIRCAP <= 2 * NCAP, "pair list must fit the manifold-reduction scratch");
            sync();
            int nsurv = 0;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                if (q * kWave < m.NPAIR) {
                    const int code = pair_code[q];
                    bool near = false;
                    if (code >= 0) {
                        const int i = code & 0xff, j = code >> 8;
                        const v3 d = ld3(s.com[i]) - ld3(s.com[j]);
                        const Real rr = s.brad[C::BROAD ? i : 0] + s.brad[C::BROAD ? j : 0];
                        near = dot(d, d) < rr * rr;
                    }
                    const uint64_t mk = wave_ballot(near);
                    if (near) plist[nsurv + dm_popc64(mk & lt)] = code;
                    nsurv += dm_popc64(mk);
                }
            }
            sync();
            for (int base = 0; base < nsurv; base += kWave) {
                const int code = (base + l < nsurv) ? plist[base + l] : -1;
                v3 x = zero3(), n = zero3(); Real dsc = 0; bool act = false;
#if DM_PAIRPREF
                {   // every lane reads (an idle lane pair 0 / 0): with the reads under the lanes' `code >= 0` branch they went out one dependent round trip at a time
                    const PairIn pin = pair_load(code >= 0 ? (code & 0xff) : 0, code >= 0 ? (code >> 8) : 0);
                    act = pair_eval(pin, x, n, dsc) && code >= 0;
                }
#else
                if (code >= 0) act = self_pair(code & 0xff, code >> 8, x, n, dsc);
#endif
                const uint64_t mk = wave_ballot(act);
                if (mk != 0) {
                    const int slot = nc + dm_popc64(mk & lt);
                    if (act && slot < m.max_contacts) store_contact(slot, x, n, dsc, code & 0xff, code >> 8);
                    nc = dm_min(m.max_contacts, nc + dm_popc64(mk));
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            if (q * kWave < m.NPAIR) {
                const int code = pair_code[q];
                v3 x = zero3(), n = zero3(); Real dsc = 0; bool act = false;
                if (code >= 0) act = self_pair(code & 0xff, code >> 8, x, n, dsc);
                const uint64_t mk = wave_ballot(act);
                if (mk != 0) {
                    const int slot = nc + dm_popc64(mk & lt);
                    if (act && slot < m.max_contacts) store_contact(slot, x, n, dsc, code & 0xff, code >> 8);
                    nc = dm_min(m.max_contacts, nc + dm_popc64(mk));
                }
            }
        }
        }
#ifdef DM_PROF_COLLISION
        mark(14);
#endif
        if (C::OBJ) {
            // the free body takes the slots that are left: lane 0 tests it against the ground, lane 1 + j against link j (capsule models, as
            // link against link); slot order = lane order.  Link ids in the slot: 254 = the body, 255 = the ground.
            const Real rb = m.ball_radius, thr_b = m.ball_thresh;
            v3 x = zero3(), n = mk3((Real)0, (Real)1, (Real)0); Real dsc = 0; bool act = false; int la = 254, lb = 255;
            if (l == 0) { dsc = bpos.y - rb; act = dsc < thr_b; x = bpos; x.y -= rb; }
            else if (l <= m.J) {
                const int j = l - 1;
                const Real* cj = s.mdl.cap[j];
                const v3 uj = ldm3(Rbp(j)) * ld3(cj), p1 = ld3(s.com[j]) + uj, d1 = (Real)-2 * uj, r = p1 - bpos;
                const Real a = dot(d1, d1);
                const Real sp = (a > (Real)1e-12) ? dm_med3((Real)0, -dot(d1, r) * dm_rcp(a), (Real)1) : (Real)0;       // closest point of the segment to the centre
                const v3 ca = p1 + sp * d1, dl = ca - bpos;
                const Real d2n = dot(dl, dl), rj = cj[3];
                const Real idn = (d2n > (Real)1e-18) ? dm_rsqrt(d2n) : (Real)0, d = d2n * idn;
                dsc = d - rj - rb;
                n = (d > (Real)1e-9) ? idn * dl : mk3((Real)0, (Real)1, (Real)0);
                x = (Real)0.5 * ((ca - rj * n) + (bpos + rb * n));
                act = (s.mdl.thresh[j] > (Real)0) && dsc < dm_min(s.mdl.thresh[j], thr_b);
                la = j; lb = 254;
            }
            const uint64_t mk = wave_ballot(act);
            if (mk != 0) {
                const int slot = nc + dm_popc64(mk & lt);
                if (act && slot < m.max_contacts) store_contact(slot, x, n, dsc, la, lb);
                nc = dm_min(m.max_contacts, nc + dm_popc64(mk));
            }
        }
        if (V2) nc = lane_bcast(nc, 0);       // (still wave-uniform: every term came from a ballot)
        const int NL = m.NL;
        const int R = NL + 3 * nc;
        if (l == 0) { s.flg[FLG_NROWS] = R; s.flg[FLG_NCONT] = nc; }
        sync();

        mark(8);
        // ---- constraint rows: lane = row.  limits | normals | frictions (2 per contact)
        // J_r[k] = +-(a_k . ((x - p0) x d) + g_k . d) on the dofs of the chain root..link (minus the other link's chain for a
        // self contact), 0 elsewhere.  A limit row (+-e_dof) is the same formula with d = 0 and X := +-a_dof on the chain {dof}.
        Real b = 0;
        uint32_t ch_lo = 0, ch_hi = 0, ng_lo = 0, ng_hi = 0; v3 xd = zero3(), dd = zero3();
        int ball_sg = 0; v3 ball_cx = zero3();
        if (l < R) {
            if (l < NL) {
                const int lr = phys2 ? (l >> 1) : l;        // v2: both rows of a limit (q - lo, then hi - q), v1: the nearer bound
                int j = s.mdl.lim_joint[lr]; int lj = s.mdl.link_info[j]; int off = DM_LI_POFF(lj);
                const int limdof = DM_LI_DOFF(lj);
                Real th = s.pose[off], pen_lo = th - s.mdl.lim_lo[lr], pen_hi = s.mdl.lim_hi[lr] - th;
                Real pen, sgn;
                if (phys2 ? !(l & 1) : (pen_lo <= pen_hi)) { sgn = 1; pen = pen_lo; } else { sgn = -1; pen = pen_hi; }
                b = (pen > 0) ? -pen / h : -m.erp * pen / h;
                xd = sgn * ld3(&s.dofrec[limdof][0]);
                if (limdof < 32) ch_lo = 1u << limdof; else ch_hi = 1u << (limdof - 32);
            } else contact_row(l, NL, nc, h, b, ch_lo, ch_hi, ng_lo, ng_hi, xd, dd, &ball_sg, &ball_cx);
        }
        // friction coefficient of this row's contact; the free body's Jacobian columns: d . v_b + ((x - p_b) x d) . w_b, signed by its side
        Real mu_row = m.friction;
        v3 jbl = zero3(), jba = zero3();
        if (C::OBJ && ball_sg != 0) { mu_row = m.ball_friction; jbl = (Real)ball_sg * dd; jba = (Real)ball_sg * cross(ball_cx - bpos, dd); }
        // y := L^-1 J_l^T in registers (static indices; dof records and L rows are wave-uniform LDS broadcasts)
        R2 y2[NP2X]; Real cvec = 0;
        dm_setprio<prio_of(DM_PRIO_ONE_Y)>();
        if constexpr (C::TREE) {
            // H = L^T L: y = L^-T J^T runs from the last dof down, against COLUMN k of L (wave-uniform broadcasts); only the pairs that
            // hold a descendant of k are touched (744 multiply-adds per row for dog3d instead of 2 016)
            typedef typename C::Topo TP;
#if DM_YPREF
            // Software-pipelined over the dofs (round 6): the records of dof k - 1 (its axis record, the quads of column k - 1 of L that hold a
            // descendant, the diagonal) are REQUESTED before the arithmetic of dof k, which needs none of them -- the two LDS round trips a dof
            // used to wait for in turn (record -> Jacobian entry, then column -> substitution) now pass behind the previous dof's chain.  Two
            // register sets, chosen by the parity of k (static indices); the root's six columns (15 quads each) are read where they are used.
            // Same operations in the same order per row: bit-identical to the loop below.
            constexpr int NQ = (NP2 + 1) / 2, QPRE = 8;
            R4 pr0[2], pr1[2]; R2 pqa[2][NQ], pqb[2][NQ]; Real pdg[2], pk1[2] = {(Real)0, (Real)0};
            // (y_nquads(k): quads of column k that hold a descendant of k)
            auto yquads = [&](auto kc) {
                constexpr int k = decltype(kc)::value, sl = k & 1;
                const Real* lc = &s.Lt[L::lcb(k)];
                static_for<(k >> 2), NQ>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr bool on0 = (2 * t > (k >> 1)) && ((TP::T.desc[k] >> (4 * t)) & 3ull), on1 = (2 * t + 1 < NP2) && (2 * t + 1 > (k >> 1)) && ((TP::T.desc[k] >> (4 * t + 2)) & 3ull);
                    if constexpr (on0 && on1) { const R4 r = *reinterpret_cast<const R4*>(&lc[4 * t]); pqa[sl][t] = R2{r[0], r[1]}; pqb[sl][t] = R2{r[2], r[3]}; }
                    else if constexpr (on0) pqa[sl][t] = *reinterpret_cast<const R2*>(&lc[4 * t]);
                    else if constexpr (on1) pqb[sl][t] = *reinterpret_cast<const R2*>(&lc[4 * t + 2]);
                });
            };
            auto yload = [&](auto kc) {
                constexpr int k = decltype(kc)::value, sl = k & 1;
                pr0[sl] = *reinterpret_cast<const R4*>(&s.dofrec[k][0]); pr1[sl] = *reinterpret_cast<const R4*>(&s.dofrec[k][4]);
                const Real* lc = &s.Lt[L::lcb(k)];
                if constexpr (y_nquads(k) <= QPRE) yquads(kc);
                if constexpr (!(k & 1) && ((TP::T.desc[k] >> (k + 1)) & 1ull)) pk1[sl] = lc[k + 1];
                pdg[sl] = lc[k];
            };
            yload(std::integral_constant<int, ND - 1>{});
            static_for<0, ND>([&](auto kkc) {
                constexpr int k = ND - 1 - decltype(kkc)::value, sl = k & 1;
                if constexpr (k > 0) { yload(std::integral_constant<int, (k > 0 ? k - 1 : 0)>{}); if (DM_YPREF >= 2) DM_SCHED_FENCE(); }      // (requests first: the waits below then count the older ones only)
                const R4 r0 = pr0[sl], r1 = pr1[sl];
                Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
                DM_OPAQUE_V(val);
                const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
                const Real raw = on ? (ng ? -val : val) : (Real)0;
                cvec += raw * r1[2];
                if constexpr (y_nquads(k) > QPRE) yquads(std::integral_constant<int, k>{});
                R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;
                static_for<(k >> 2), NQ>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr bool on0 = (2 * t > (k >> 1)) && ((TP::T.desc[k] >> (4 * t)) & 3ull), on1 = (2 * t + 1 < NP2) && (2 * t + 1 > (k >> 1)) && ((TP::T.desc[k] >> (4 * t + 2)) & 3ull);
                    if constexpr (on0) acc2 += pqa[sl][t] * y2[2 * t];
                    if constexpr (on1) acc3 += pqb[sl][t] * y2[(2 * t + 1 < NP2) ? 2 * t + 1 : 0];
                });
                acc2 += acc3;
                Real acc = raw - (acc2[0] + acc2[1]);
                if constexpr (!(k & 1) && ((TP::T.desc[k] >> (k + 1)) & 1ull)) acc -= pk1[sl] * y2[k >> 1][1];
                Real yk = acc * pdg[sl];
                DM_OPAQUE_V(yk);
                y2[k >> 1][k & 1] = yk;
                DM_SCHED_FENCE();
            });
#else
#pragma unroll
            for (int kk = 0; kk < ND; ++kk) {
                const int k = ND - 1 - kk;
                const R4 r0 = *reinterpret_cast<const R4*>(&s.dofrec[k][0]), r1 = *reinterpret_cast<const R4*>(&s.dofrec[k][4]);
                Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
                DM_OPAQUE_V(val);      // computed unconditionally: a branch per dof here makes the optimizer sink the substitution chains below the loop (and spill every column)
                const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
                const Real raw = on ? (ng ? -val : val) : (Real)0;
                cvec += raw * r1[2];
                R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;
                const Real* lc = &s.Lt[L::lcb(k)];                                // lc[i] = L_ik
#pragma unroll
                for (int t = (k >> 2); t < (NP2 + 1) / 2; ++t) {      // quads of dofs: one 16-B broadcast read when both pairs hold a descendant of k
                    const bool on0 = (2 * t > (k >> 1)) && ((TP::T.desc[k] >> (4 * t)) & 3ull), on1 = (2 * t + 1 < NP2) && (2 * t + 1 > (k >> 1)) && ((TP::T.desc[k] >> (4 * t + 2)) & 3ull);
                    if (on0 && on1) {
                        const R4 r = *reinterpret_cast<const R4*>(&lc[4 * t]);
                        const R2 ra = {r[0], r[1]}, rb = {r[2], r[3]};
                        acc2 += ra * y2[2 * t]; acc3 += rb * y2[(2 * t + 1 < NP2) ? 2 * t + 1 : 0];
                    } else if (on0) acc2 += *reinterpret_cast<const R2*>(&lc[4 * t]) * y2[2 * t];
                    else if (on1) acc3 += *reinterpret_cast<const R2*>(&lc[4 * t + 2]) * y2[(2 * t + 1 < NP2) ? 2 * t + 1 : 0];
                }
                acc2 += acc3;
                Real acc = raw - (acc2[0] + acc2[1]);
                if (!(k & 1) && ((TP::T.desc[k] >> (k + 1)) & 1ull)) acc -= lc[k + 1] * y2[k >> 1][1];
                Real yk = acc * s.Lt[L::lcb(k) + k];
                DM_OPAQUE_V(yk);
                y2[k >> 1][k & 1] = yk;
                DM_SCHED_FENCE();      // the branches of the tree are independent chains: without a fence the scheduler hoists their loads and spills
            }
#endif
        } else if constexpr (DM_YPREF != 0 && C::FULLD) {      // (the fallback class of the two-per-wave kernel only: at the one-per-wave biped kernels' 128 registers the second row buffer spills)
        // (round 6) the dense loop software-pipelined like the two-per-wave kernel's: row k + 1 of the factor and its dof record are requested before the
        // accumulation chain of step k -- this is the loop of the 64-lane fallback of a two-per-wave pair, i.e. of the waves a closed-loop launch waits for
        R2 lrp[2][NP2]; R4 rrp[2][2];
        auto ydload = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            rrp[k & 1][0] = *reinterpret_cast<const R4*>(&s.dofrec[k][0]); rrp[k & 1][1] = *reinterpret_cast<const R4*>(&s.dofrec[k][4]);
            const R2* lrow_ = reinterpret_cast<const R2*>(&s.Lt[L::lrow(k)]);
            static_for<0, (k >> 1) + 1>([&](auto pc) { constexpr int p = decltype(pc)::value; lrp[k & 1][p] = lrow_[p]; });
        };
        static_assert(L::LPAD % 2 == 0, "row pairs are read as 8-byte words");
        ydload(std::integral_constant<int, 0>{});
        static_for<0, ND>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            Real yk = 0;
            // (C::FULLD: no `k < D` tests -- as wave-uniform branches around the requests they make the compiler wait for everything in flight at every dof, DuoSim::substep_post)
            if constexpr (k + 1 < ND) { if (C::FULLD || k + 1 < D) ydload(std::integral_constant<int, (k + 1 < ND ? k + 1 : 0)>{}); }
            if (C::FULLD || k < D) {
                const R4 r0 = rrp[k & 1][0], r1 = rrp[k & 1][1];
                Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
                if (C::FULLD) DM_OPAQUE_V(val);
                const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
                const Real raw = on ? (ng ? -val : val) : (Real)0;
                cvec += raw * r1[2];
                R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;      // two independent accumulation chains
                static_for<0, (k >> 1)>([&](auto pc) { constexpr int p = decltype(pc)::value; if constexpr (p & 1) acc3 += lrp[k & 1][p] * y2[p]; else acc2 += lrp[k & 1][p] * y2[p]; });
                acc2 += acc3;
                Real acc = raw - (acc2[0] + acc2[1]);
                if constexpr (k & 1) acc -= lrp[k & 1][k >> 1][0] * y2[k >> 1][0];
                yk = acc * lrp[k & 1][k >> 1][k & 1];
                if (C::FULLD) DM_OPAQUE_V(yk);
            }
            y2[k >> 1][k & 1] = yk;
            if (DM_YPREF_DENSE_FENCE) DM_SCHED_FENCE();
        });
        } else {
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            Real yk = 0;
            if (k < D) {
                const R4 r0 = *reinterpret_cast<const R4*>(&s.dofrec[k][0]), r1 = *reinterpret_cast<const R4*>(&s.dofrec[k][4]);
                const Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
                const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
                const Real raw = on ? (ng ? -val : val) : (Real)0;
                cvec += raw * r1[2];
                R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;      // two independent accumulation chains
                const R2* lrow = reinterpret_cast<const R2*>(&s.Lt[L::lrow(k)]);
#pragma unroll
                for (int p = 0; p < (k >> 1); ++p) { if (p & 1) acc3 += lrow[p] * y2[p]; else acc2 += lrow[p] * y2[p]; }
                acc2 += acc3;
                Real acc = raw - (acc2[0] + acc2[1]);
                if (k & 1) acc -= s.Lt[L::lrow(k) + k - 1] * y2[k >> 1][0];
                yk = acc * s.Lt[L::lrow(k) + k];
            }
            y2[k >> 1][k & 1] = yk;
        }
        }
        dm_setprio<prio_of(0)>();
        if (C::OBJ) {
            // the free body's block of the mass matrix is diagonal: its rows of Y = M^-1/2 J^T are a scaling; J v* gains its share
            cvec += dot(jbl, bvs) + dot(jba, bws);
            const Real sm = dm_sqrt(m.ball_inv_mass), si = dm_sqrt(m.ball_inv_inertia);
            y2[C::OBJ ? NP2 : 0][0] = sm * jbl.x; y2[C::OBJ ? NP2 : 0][1] = sm * jbl.y; y2[C::OBJ ? NP2 + 1 : 0][0] = sm * jbl.z;
            y2[C::OBJ ? NP2 + 1 : 0][1] = si * jba.x; y2[C::OBJ ? NP2 + 2 : 0][0] = si * jba.y; y2[C::OBJ ? NP2 + 2 : 0][1] = si * jba.z;
        }
        mark(9);
        // projected Gauss-Seidel in impulse space; u = J v* + A lambda is kept per lane.
        // Sweep order: limits, normals, frictions; a friction row is boxed by mu * (current normal impulse of its contact).
        const int RN = NL + nc;
        const bool is_fric = l >= RN && l < R;
        Real lam = 0;
        // no row can leave lambda = 0 unless some limit / normal row starts violated: skip A and the sweeps otherwise
        if (wave_ballot(l < RN && (b - cvec) > 0) != 0) {
            // A = Y^T Y: lane l keeps row l of A in a register file indexed by the row id
            // (rows >= RREG of a heavily contacted character overflow to a per-env HBM scratch block, [row][lane])
            RowFile<Real, RREG> arow;
            Real adiag;
            { R2 a2 = {(Real)0, (Real)0};
#pragma unroll
              for (int p = 0; p < NP2X; ++p) a2 += y2[p] * y2[p];
              adiag = a2[0] + a2[1]; }
            // rows are pre-scaled by 1/A_ll and the diagonal is zeroed: the sweep keeps t_l = lambda_l + (b_l - u_l)/A_ll, the
            // pre-clamp target of row l, which a visit to row l itself leaves unchanged -- one uniform FMA per row, no add
            const Real inv_adiag = (l < R) ? (Real)1 / adiag : (Real)0;
            if (R <= 32) {
                Real g[32];
                // (y is made opaque to the optimizer first: without it clang's middle end does not terminate on the select
                // chains feeding the MFMAs)
#pragma unroll
                for (int p = 0; p < NP2X; ++p) DM_OPAQUE_V(y2[p]);
                wave_gram32<NP2X>(y2, g);
#pragma unroll
                for (int r = 0; r < 32; ++r) arow.set(r, (l == r) ? (Real)0 : g[r] * inv_adiag);
            } else if (RREG >= kMaxRows && ND <= 34) {
                // the wide biped class (fallback of the two-per-wave kernel): all 64 rows on the matrix core, straight into registers
                Real g[64];
#pragma unroll
                for (int p = 0; p < NP2X; ++p) DM_OPAQUE_V(y2[p]);
                wave_gram64<NP2X>(y2, g);
#pragma unroll
                for (int r = 0; r < 64; ++r) arow.set(r, (l == r) ? (Real)0 : g[r] * inv_adiag);
            } else if (C::GRAM64) {
                // narrow row file, Gram on the matrix core: rows 32..63 go to the overflow block ([row][lane]; all of them, the sweep reads
                // rows < R only)
                static_assert(RREG >= 32 || !C::GRAM64, "two halves of 32 entries");
#pragma unroll
                for (int p = 0; p < NP2X; ++p) DM_OPAQUE_V(y2[p]);
                {
                    Real g[32];
                    wave_gram64_half<NP2X, 1>(y2, g);            // entries 32..63 first: those past RREG leave for the overflow block at once
#pragma unroll
                    for (int r = 0; r < 32; ++r) if (32 + r < RREG) arow.set(32 + r < RREG ? 32 + r : 0, (l == r + 32) ? (Real)0 : g[r] * inv_adiag);
                    if (RREG < kMaxRows && R > RREG) {
#pragma unroll
                        for (int r = 0; r < 32; ++r) if (32 + r >= RREG) aovf[(32 + r - RREG) * kWave + l] = (l == r + 32) ? (Real)0 : g[r] * inv_adiag;
                    }
                }
#pragma unroll
                for (int p = 0; p < NP2X; ++p) DM_OPAQUE_V(y2[p]);
                {
                    Real g[32];
                    wave_gram64_half<NP2X, 0>(y2, g);
#pragma unroll
                    for (int r = 0; r < 32; ++r) arow.set(r, (l == r) ? (Real)0 : g[r] * inv_adiag);
                }
            } else {
                for (int r = 0; r < R; ++r) {
                    R2 a2 = {(Real)0, (Real)0};
#pragma unroll
                    for (int p = 0; p < NP2X; ++p) { const R2 bb = {lane_bcast(y2[p][0], r), lane_bcast(y2[p][1], r)}; a2 += y2[p] * bb; }
                    const Real v = (l == r) ? (Real)0 : (a2[0] + a2[1]) * inv_adiag;
                    if (RREG >= kMaxRows || r < RREG) arow.set(r, v);
                    else aovf[(r - RREG) * kWave + l] = v;
                }
            }
            mark(10);
            Real t = (b - cvec) * inv_adiag;
            const int nrm_lane = is_fric ? NL + ((l - RN) >> 1) : 0;
            Real lo = 0, hi = is_fric ? (Real)0 : ((l < NL) ? m.lim_max_impulse : (Real)1e30);      // (limit rows: maxAppliedImpulse)
            int Rv = R, RNv = RN, lv = l;
            constexpr int PFD = C::PFD;             // rows of look-ahead for the overflow block (a ring of PFD + 1 registers)
            Real pre[PFD + 1] = {};
#if DM_PRIO
            // wave priority by load for the duration of the sweep (see DuoSim::substep_post; dog3d +1.6 %): thresholds at the median / p90 row count of the class
            { constexpr int PLO = (ND > 34) ? 28 : 16, PHI = (ND > 34) ? 40 : 22;
              if (Rv > PHI) dm_setprio<3>(); else if (Rv > PLO) dm_setprio<prio_of(2)>(); else dm_setprio<prio_of(1)>(); }
#endif
            for (int it = 0; it < m.solver_iters; ++it) {
                DM_OPAQUE_S(Rv); DM_OPAQUE_S(RNv); DM_OPAQUE_V(lv);
                uint32_t one = 1; DM_OPAQUE_S(one);
                // statically unrolled over the row id (register-file index and lane select are immediates); rows >= R are
                // skipped in blocks of 4 by wave-uniform branches (a lane beyond R has t = 0, lambda = 0 and changes nothing)
                static_for<0, kMaxRows / 4>([&](auto blkc) {
                    constexpr int blk = decltype(blkc)::value;
                    if (blk * 4 < Rv) {
                        static_for<0, 4>([&](auto ic) {
                            constexpr int r = blk * 4 + decltype(ic)::value;
                            if (__builtin_expect(r == RNv, 0)) { const Real ln = wave_shfl(lam, nrm_lane); if (is_fric) { hi = mu_row * ln; lo = -hi; } }
                            // rows >= RREG live in the HBM/L2 overflow block: row r + PFD is requested PFD rows ahead (a ring of
                            // registers), so the load latency sits beside the sweep's dependent chain instead of on it
                            if (RREG < kMaxRows && r + PFD >= RREG && r + PFD < kMaxRows) { const Real* ap = aovf + lv; DM_OPAQUE_V(ap); pre[(r + PFD) % (PFD + 1)] = ap[(r + PFD - RREG) * kWave]; }
                            const Real nl = dm_med3(lo, t, hi);
                            const Real delta = lane_bcast(nl - lam, r);
                            const Real ar = (r < RREG) ? arow.get(r < RREG ? r : 0) : pre[r % (PFD + 1)];
                            t -= ar * delta;
                            if constexpr (C::PGS_MASKSEL) lam = row_sel_c<r>(lam, nl, lv, one);        // lane r takes its new lambda: a select on a scalar-unit lane mask (no v_cmp)
                            else if (lv == r) lam = nl;
                        });
                    }
                });
            }
            if (l >= R) lam = 0;
        } else mark(10);
        dm_setprio<prio_of(DM_PRIO_ONE_BACK)>();
        mark(11);
        if (TAPS && dbg.lambda) { dbg.lambda[(size_t)e * kMaxRows + l] = lam; if (l < 2) dbg.rows[(size_t)e * 2 + l] = s.flg[FLG_NROWS + l]; }
        // delta v = L^-T (Y lambda): transposing wave reduction of y_r[k] lambda_r, dof k's total lands in lane k
        Real z;
        {
            Real w[NP2];
            const bool bit = (l & 1) != 0;
#pragma unroll
            for (int p = 0; p < NP2; ++p) {
                const Real a = y2[p][0] * lam, bb = y2[p][1] * lam;
                w[p] = (bit ? bb : a) + wave_shfl_xor_c<1>(bit ? a : bb);
            }
            tr_stage<NP2, 2>(w);
            tr_stage<(NP2 + 1) / 2, 4>(w);
            tr_stage<(NP2 + 3) / 4, 8>(w);
            tr_stage<(NP2 + 7) / 8, 16>(w);
            tr_stage<(NP2 + 15) / 16, 32>(w);
            z = w[0];
        }
        if constexpr (C::TREE) z = tree_fwd(z, (l < ND) ? s.Lt[L::lcb(l < ND ? l : 0) + (l < ND ? l : 0)] : (Real)1);       // delta v = L^-1 (Y lambda)
        else {
            Real dinv = (l < ND) ? Lx(l < ND ? l : 0, l < ND ? l : 0) : (Real)1;
            z = back_substitute(z, dinv);
        }
        if (l < D) s.vel[vidx] = clamp_vel(vstar + z, l);
        if (C::OBJ) {
            // delta v of the free body = M^-1/2 (Y_b lambda): six wave sums; then semi-implicit Euler with the exponential map
            Real dv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                dv[k] = wave_sum(y2[C::OBJ ? NP2 + (k >> 1) : 0][k & 1] * lam);
            }
            if (l == 0) {
                const Real sm = dm_sqrt(m.ball_inv_mass), si = dm_sqrt(m.ball_inv_inertia);
                const v3 bv = bvs + sm * mk3(dv[0], dv[1], dv[2]), bw = bws + si * mk3(dv[3], dv[4], dv[5]);
                Real* ob = s.obj;
                st3(ob + (C::OBJ ? OB_VX : 0), bv); st3(ob + (C::OBJ ? OB_WX : 0), bw);
                st3(ob + (C::OBJ ? OB_PX : 0), bpos + h * bv);
                stq(ob + (C::OBJ ? OB_QW : 0), qnormalize(qmul(quat_exp(h * bw), ldq(ob + (C::OBJ ? OB_QW : 0)))));
            }
        }
        sync();
        integrate(h);
        sync();
        dm_setprio<prio_of(0)>();
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
    // RAW: cMotion::CalcFrame / CalcFrameVel only (no cycle offset, no origin transform, root quaternion not standardised)
    template <bool RAW = false> DM_DEV void kin_sample(double time, Real* kp, Real* kv) {
        int idx, cyc; double blend; kin_index_blend(time, idx, blend, cyc);
        Real b = (Real)(blend < 0 ? 0 : (blend > 1 ? 1 : blend));
        const Real* f0 = m.frames + (size_t)idx * m.P; const Real* f1 = f0 + m.P;
        q4 orot = ldq(s.kin + 3);
        if (l < m.J) {
            int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li);
            if (l == 0) {
                v3 rp = ((Real)1 - b) * ld3(f0) + b * ld3(f1);
                q4 rr = qnormalize(qslerp(ldq(f0 + 3), b, ldq(f1 + 3), m.slerp_one));
                if (!RAW) {
                    if (m.loop) rp = rp + (Real)cyc * mk3(m.cycle_delta[0], m.cycle_delta[1], m.cycle_delta[2]);
                    rr = qstandardize(qmul(orot, rr));
                    rp = qrot(orot, rp) + ld3(s.kin);
                }
                st3(kp, rp); stq(kp + 3, rr);
            } else if (jt == JT_SPHERICAL) stq(kp + off, qslerp(ldq(f0 + off), b, ldq(f1 + off), m.slerp_one));
            else if (jt == JT_REVOLUTE) kp[off] = ((Real)1 - b) * f0[off] + b * f1[off];
        }
        bool over = !m.loop && time >= m.duration;
        const Real* v0 = m.frame_vel + (size_t)idx * m.P; const Real* v1 = v0 + m.P;
        Real bv = (Real)blend;
        for (int i = l; i < m.P; i += LW) kv[i] = over ? (Real)0 : ((Real)1 - bv) * v0[i] + bv * v1[i];
        sync();
        if (!RAW && l == 0) {
            v3 v = qrot(orot, ld3(kv)), w = qrot(orot, ld3(kv + 3));
            st3(kv, v); st3(kv + 3, w);
        }
        sync();
    }
    // root rotation of the kin character at `time`, every lane (cKinCharacter::CalcPose root slot)
    DM_DEV q4 kin_root_rot(double time) const {
        int idx, cyc; double blend; kin_index_blend(time, idx, blend, cyc);
        Real b = (Real)(blend < 0 ? 0 : (blend > 1 ? 1 : blend));
        const Real* f0 = m.frames + (size_t)idx * m.P; const Real* f1 = f0 + m.P;
        q4 rr = qnormalize(qslerp(ldq(f0 + 3), b, ldq(f1 + 3), m.slerp_one));
        return qstandardize(qmul(ldq(s.kin + 3), rr));
    }
    // cKinCharacter::RotateOrigin about the kin root position rp by the heading difference dh (rotation about +y); lane 0
    DM_DEV void kin_rotate_origin(Real dh, v3 rp) {
        Real sh, ch; dm_sincos((Real)0.5 * dh, sh, ch);
        q4 drot = mkq(ch, (Real)0, sh, (Real)0);
        stq(s.kin + 3, qnormalize(qmul(drot, ldq(s.kin + 3))));
        st3(s.kin, rp + qrot(drot, ld3(s.kin) - rp));
    }
    // cSceneImitate::UpdateKinChar: advance the clip clock; on phase wrap SyncKinCharNewCycle (SceneImitate.cpp:420-444)
    // turns the kin character to the sim heading (sync_char_root_rot) and snaps its root x, z to the sim root (sync_char_root_pos)
    // CLIPS (the AMP / tap instantiations): with a multi-clip dataset the kinematic character runs on the env's own clip (cClipsController's active motion) --
    // its duration and loop mode decide where a cycle ends, its frames where the root is at that moment.  (Round 5: until then every env's kinematic clock wrapped
    // on clip 0's period; found by holding the kinematic pose against the compiled cKinCharacter after every update, tests/test_ref_draw_order.py.)
    template <bool CLIPS = false>
    DM_DEV void kin_update(double dt) {
        DM_OPAQUE_V(l);
        double t0 = s.clk[CLK_KIN], t1 = t0 + dt;
        const bool multi = CLIPS && m.num_clips > 1;
        double ph0, ph1;
        if (multi) {
            const double dur = m.clip_dur[clip]; const bool loop = m.clip_loop[clip] != 0;
            ph0 = t0 / dur; ph1 = t1 / dur;
            if (loop) { ph0 -= floor(ph0); ph1 -= floor(ph1); } else { ph0 = ph0 < 0 ? 0 : (ph0 > 1 ? 1 : ph0); ph1 = ph1 < 0 ? 0 : (ph1 > 1 ? 1 : ph1); }
        } else { ph0 = kin_phase(t0); ph1 = kin_phase(t1); }
        sync();
        if (l == 0) s.clk[CLK_KIN] = t1;
        if (ph1 < ph0 && (m.sync_root_pos || m.sync_root_rot)) {
            v3 kr; q4 krot = mkq((Real)1, (Real)0, (Real)0, (Real)0);
            if (multi) {
                const ModelDev<Real> mc = model_of_clip(clip);
                EnvSim<Real, C, TAPS, LW> cs(mc, s, l);
                kr = cs.kin_root_pos(t1); if (m.sync_root_rot) krot = cs.kin_root_rot(t1);
            } else { kr = kin_root_pos(t1); if (m.sync_root_rot) krot = kin_root_rot(t1); }
            Real dh = 0;
            if (m.sync_root_rot) dh = calc_heading(ldq(s.pose + 3)) - calc_heading(krot);
            if (l == 0) {
                if (m.sync_root_rot) kin_rotate_origin(dh, kr);          // rotation about the root: kr itself does not move
                if (m.sync_root_pos) {
                    v3 sp = ld3(s.pose);
                    Real hgt = kr.y - s.kin[1];
                    v3 target = mk3(sp.x, (Real)0 + hgt, sp.z);
                    v3 delta = target - kr;
                    s.kin[0] += delta.x; s.kin[1] += delta.y; s.kin[2] += delta.z;
                }
            }
        }
        sync();
    }

    // ------------------------------------------------------------------ action -> PD targets
    DM_DEV void set_action(const float* a) {
        if (l < m.J && l > 0) {
            int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li), ao = m.act_off[l];
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
        Real* kp = scratch(); Real* kv = scratch() + NP;
        kin_sample(s.clk[CLK_KIN], kp, kv);
        if (l < m.J && l > 0) {
            int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li);
            if (jt == JT_SPHERICAL) {
                v3 ev = quat_to_rotvec(ldq(kp + off), (Real)0.000001);
                stq(s.tar + off, qnormalize(exp_map_to_quat(ev)));
            } else if (jt == JT_REVOLUTE) s.tar[off] = kp[off];
        }
        sync();
    }

    // ------------------------------------------------------------------ one scene update (cSceneSimChar::Update)
    // phase 0 = stable-PD solve, phases 1..n = rigid-body substeps; the three share one copy of the kinematics /
    // dynamics / Cholesky code (the instruction stream of the 20-update loop must fit the instruction cache).
    // reuse_kin: pose / vel are unchanged since the previous phase's kinematics (stable-PD solve -> first substep); only
    // the base acceleration differs, and it enters every joint-origin acceleration as the same additive constant.
    template <bool PERT = false, bool V2 = false>
    DM_DEV void dyn_phase(int ph, Real dt, Real h, DebugTaps<Real> dbg, int e, bool tap_only, bool reuse_kin, Real* aovf, const double* pert = nullptr, Real* manif = nullptr, bool kin_done = false) {
        // lane id / link word are re-materialised per phase: keeps the optimizer from hoisting every per-lane LDS address
        // out of the 20-update loop (dozens of long-lived VGPRs that end up in scratch)
        DM_OPAQUE_V(l); DM_OPAQUE_V(li);
        mark(ph == 0 ? 0 : 4);
        if (reuse_kin) {
            if (l < m.J) { v3 da = gravity_a0() - spd_a0(); st3(s.aj[l], ld3(s.aj[l]) + da); }
            sync();
        } else if (!kin_done) { dm_setprio<DM_PRIO_ONE_KIN>(); kinematics(s.pose, s.vel, ph == 0 ? spd_a0() : gravity_a0()); dm_setprio<0>(); }      // (kin_done: kin_pre() ran it for this state)
        mark(ph == 0 ? 1 : 5);
        dynamics(ph == 0 ? 0 : 1, ph == 0 ? dt : (Real)0);
        mark(ph == 0 ? 2 : 6);
        if (TAPS && dbg.H) {
            const int D = m.D;
            if constexpr (C::TREE) {
                R2 c2[NP2]; Real hd;
                tree_load_col(c2, hd);
                if (l < D) {
                    Real* Hd = dbg.H + (size_t)e * D * D;
#pragma unroll
                    for (int i = 0; i < ND; ++i) if (i > l) { Hd[i * D + l] = c2[i >> 1][i & 1]; Hd[l * D + i] = c2[i >> 1][i & 1]; }
                    Hd[l * D + l] = hd - ((ph == 0) ? dt * s.mdl.kd[DM_DI_JOINT(s.mdl.dof_info[l])] : (Real)0);
                }
            } else
            for (int i = l; i < D * D; i += LW) { int r = i / D, c = i % D; Real v = (c <= r) ? Lx(r, c) : Lx(c, r); if (r == c && ph == 0) v -= dt * s.mdl.kd[DM_DI_JOINT(s.mdl.dof_info[r])]; dbg.H[(size_t)e * D * D + i] = v; }
            if (l < D) dbg.C[(size_t)e * D + l] = s.dofrec[l][7];
        }
        if (tap_only) return;
        if (ph == 0) spd_rhs(dt);
        else { if (l < m.D) { Real r = s.tau[l] - s.dofrec[l][7]; if (PERT && pert) r += pert_gen_force(l); s.rhs[l] = r; } sync(); }
        DM_OPAQUE_V(l);
        dm_setprio<DM_PRIO_ONE_CHOL>();
        if constexpr (C::TREE) tree_solve(s.rhs); else chol_solve(s.rhs);
        dm_setprio<0>();
        DM_OPAQUE_V(l); DM_OPAQUE_V(li);
        if (ph == 0) { mark(3); spd_post(dt); }
        else substep_post<V2, !TAPS && !PERT && !V2>(h, dbg, e, aovf, manif);
    }
    // Top of a scene update, before anything of it changes the env: link kinematics of the current state (the stable-PD pass of this
    // update reuses them, kin_done) and cSceneSimChar::CheckValidEpisode on them (cSimCharacter::HasVelExploded, sim/SimCharacter.cpp:
    // 571-586: any link velocity component beyond 100) -- i.e. the validity test the reference's driver runs after the PREVIOUS update
    // (DeepMimic.py:62-80).  An invalid state is latched in bit 1 of FLG_OVER for the rest of the launch: with DM_END_EPISODE_EARLY the
    // step kernels stop before this update, emit() reports valid = 0 either way.
    DM_DEV void kin_pre(bool act = true) {
        mark(0);
        kinematics(s.pose, s.vel, spd_a0());
        if (act && l < m.J) {
            const v3 vc = link_vcom(l), w = ld3(s.w[l]);
            Real mx = dm_max(dm_max(dm_abs(vc.x), dm_abs(vc.y)), dm_abs(vc.z));
            mx = dm_max(mx, dm_max(dm_max(dm_abs(w.x), dm_abs(w.y)), dm_abs(w.z)));
            if (mx > (Real)100) dm_atomic_or(&s.flg[FLG_OVER], 2);
        }
        sync();
    }
    template <bool PERT = false, bool V2 = false>
    DM_DEV void update(double dt, DebugTaps<Real> dbg, int e, Real* aovf, double* pert = nullptr, Real* manif = nullptr, bool kin_done = false) {
        if (l == 0) { s.clk[CLK_TIMER] += dt; s.clk[CLK_CTRL] += dt; s.flg[FLG_NEED_ACTION] = 0; }
        if (PERT && pert) { if (l == 0) pert_tick(pert, e, dt); sync(); }
        kin_update<PERT>(dt);
        const Real h = (Real)(dt / m.num_sim_substeps);
        for (int ph = 0; ph <= m.num_sim_substeps; ++ph) dyn_phase<PERT, V2>(ph, (Real)dt, h, dbg, e, false, ph == 1, aovf, pert, V2 ? manif : nullptr, kin_done && ph == 0);
        if (l == 0) {                      // cCtController::CheckNeedNewAction (CtController.cpp:221-227)
            double cur = s.clk[CLK_CTRL] + s.clk[CLK_INIT_OFF], pad = 0.001 * dt;
            int c1 = (int)floor((cur + pad) / m.query_period), c0 = (int)floor((cur + pad - dt) / m.query_period);
            s.flg[FLG_NEED_ACTION] = (c1 != c0) ? 1 : 0;
            bool over = episode_over_now();
            if (PERT) { if (m.enable_root_rot_fail && m.enable_fall_end && !over) over = root_rot_failed_now(); }
            s.flg[FLG_OVER] = (s.flg[FLG_OVER] & 2) | (over ? 1 : 0);      // read by the step kernels when DM_END_EPISODE_EARLY is set (bit 1: kin_pre's invalid latch)
        }
        sync();
    }

    // ------------------------------------------------------------------ termination
    // cSceneSimChar::HasFallenContact; cSceneHeadingAMPGetup's override (SceneHeadingAMPGetup.cpp:255-264): never while getting up
    DM_DEV bool has_fallen_contact() const {
        if (m.scene_goal == 3 && s.getup) return false;
        return m.enable_contact_fall && (s.flg[FLG_CONTACT] & s.fall_mask) != 0;
    }
    DM_DEV bool has_fallen(const Real* kp) const {
        bool f = has_fallen_contact();
        if (m.enable_root_rot_fail && kp) f = f || (quat_theta(qmul(ldq(kp + 3), qconj(ldq(s.pose + 3)))) > (Real)(0.5 * DM_PI));
        return f;
    }

    // ------------------------------------------------------------------ reward + observation + flags
    DM_DEV void emit(const StepIO<Real>& io, DebugTaps<Real> dbg, int e, bool write_flags) {
        DM_OPAQUE_V(l); DM_OPAQUE_V(li);
        const int J = m.J;
        Real* kp = scratch(); Real* kv = scratch() + NP;
        Real* ee_k = scratch() + 2 * NP;                 // J x 3 kin joint positions
        Real* red = scratch() + 2 * NP + 3 * NJ;         // J x 4 per-joint reduction terms
        const v3 zero = zero3();
        kin_sample(s.clk[CLK_KIN], kp, kv);
        if (TAPS && dbg.kin_pose) for (int i = l; i < m.P; i += LW) { dbg.kin_pose[(size_t)e * m.P + i] = kp[i]; dbg.kin_vel[(size_t)e * m.P + i] = kv[i]; }
        // kin character: joint positions and COM velocity (cRBDUtil::CalcCoM)
        kinematics(kp, kv, zero);
        if (l < J) {
            st3(ee_k + l * 3, ld3(s.p[l]));
            v3 vc = ld3(s.vj[l]) + cross(ld3(s.w[l]), ld3(s.com[l]) - ld3(s.p[l]));
            st3(red + l * 4, s.mdl.mass[l] * vc);
        }
        sync();
        if (l == 0) {
            v3 acc = zero; Real tm = 0;
            for (int j = 0; j < J; ++j) { acc = acc + ld3(red + j * 4); tm += s.mdl.mass[j]; }
            st3(s.sc + 0, ((Real)1 / tm) * acc);
        }
        sync();
        // sim character
        kinematics(s.pose, s.vel, zero);
        v3 vcom = zero;
        if (l < J) {
            vcom = ld3(s.vj[l]) + cross(ld3(s.w[l]), ld3(s.com[l]) - ld3(s.p[l]));
            st3(red + l * 4, s.mdl.mass[l] * vcom);
        }
        sync();
        if (l == 0) {
            v3 acc = zero; Real tm = 0;
            for (int j = 0; j < J; ++j) { acc = acc + ld3(red + j * 4); tm += s.mdl.mass[j]; }
            st3(s.sc + 3, ((Real)1 / tm) * acc);
        }
        sync();
        if (TAPS && dbg.links && l < J) {
            Real* o = dbg.links + ((size_t)e * J + l) * 21;
            st3(o, ld3(s.com[l])); for (int k = 0; k < 9; ++k) o[3 + k] = Rbp(l)[k];
            st3(o + 12, vcom); st3(o + 15, ld3(s.w[l])); st3(o + 18, ld3(s.p[l]));
        }
        // origin frames (cKinTree::BuildOriginTrans): rotation about y by -heading, translation by -root(x,z)
        Real head0 = calc_heading(ldq(s.pose + 3)), head1 = calc_heading(ldq(kp + 3));
        m3 O0 = rot_y(-head0), O1 = rot_y(-head1);
        // per-joint reward terms
        if (l < J) {
            int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li);
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
            if (DM_LI_IS_EE(li)) {
                v3 p0 = ld3(s.p[l]), p1 = ld3(ee_k + l * 3);
                v3 rel0 = p0 - ld3(s.pose), rel1 = p1 - ld3(kp);
                rel0.y = p0.y - (Real)0; rel1.y = p1.y - s.kin[1];
                v3 dlt = O1 * rel1 - O0 * rel0;
                ee = dot(dlt, dlt);
            }
            Real dw = m.diffw[l];
            red[l * 4 + 0] = dw * pe; red[l * 4 + 1] = dw * ve; red[l * 4 + 2] = ee;
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
            if (fallen || m.scene_amp) r = 0;          // cSceneImitateAMP::CalcReward is 0 outside the test-mode time-warp score
            // cSceneImitateAMP::CheckTerminate keeps only the fall test (SceneImitateAMP.cpp:184-188)
            bool fail = (m.enable_fall_end && fallen) || (!m.scene_amp && !m.loop && s.clk[CLK_KIN] >= m.duration);
            bool end = fail || (s.clk[CLK_TIMER] >= s.clk[CLK_TIMER_MAX]);
            if (write_flags) {
                if (io.rewards) io.rewards[e] = (float)r;
                if (io.terminate) io.terminate[e] = fail ? TERM_FAIL : TERM_NULL;
                if (io.episode_end) io.episode_end[e] = end ? 1 : 0;
            }
            s.sc[6] = end ? (Real)1 : (Real)0;
            if (TAPS && dbg.reward_terms) { Real* o = dbg.reward_terms + (size_t)e * 5; o[0] = pose_err; o[1] = vel_err; o[2] = ee_err; o[3] = root_err; o[4] = com_err; }
        }
        // CheckValidEpisode: any link velocity component beyond 100 (SimCharacter.cpp:571-586)
        if (l == 0) s.flg[FLG_VALID] = (s.flg[FLG_OVER] & 2) ? 0 : 1;      // (an update inside this launch left the env invalid: kin_pre)
        sync();
        if (l < J) {
            v3 w = ld3(s.w[l]);
            Real mx = dm_max(dm_max(dm_abs(vcom.x), dm_abs(vcom.y)), dm_abs(vcom.z));
            mx = dm_max(mx, dm_max(dm_max(dm_abs(w.x), dm_abs(w.y)), dm_abs(w.z)));
            if (mx > (Real)100) s.flg[FLG_VALID] = 0;
        }
        sync();
        if (l == 0 && write_flags && !s.flg[FLG_VALID]) s.sc[6] = (Real)1;       // the driver resets after an invalid episode too (DeepMimic.py:62-80): DM_AUTO_RESET follows
        if (l == 0 && io.valid && write_flags) io.valid[e] = s.flg[FLG_VALID];
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
                m3 Rb = ldm3(Rbp(l));
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
            if (C::OBJ && m.scene_goal == 5 && l == 0) {
                // cSceneDribbleAMP::RecordTaskState (SceneDribbleAMP.cpp:554-590): the ball in the origin frame -- position, normal and
                // tangent of its rotation, linear and angular velocity
                const Real* ob = s.obj;
                const v3 bp = ld3(ob + (C::OBJ ? OB_PX : 0));
                const v3 pc = O0 * mk3(bp.x - rpos.x, bp.y, bp.z - rpos.z);
                const m3 Rq = O0 * quat_to_rot(ldq(ob + (C::OBJ ? OB_QW : 0)));
                const v3 nrm = col(Rq, 1), tan = col(Rq, 0), bv = O0 * ld3(ob + (C::OBJ ? OB_VX : 0)), bw = O0 * ld3(ob + (C::OBJ ? OB_WX : 0));
                float* o = out + base + 1 + 15 * J;
                o[0] = (float)pc.x; o[1] = (float)pc.y; o[2] = (float)pc.z; o[3] = (float)nrm.x; o[4] = (float)nrm.y; o[5] = (float)nrm.z;
                o[6] = (float)tan.x; o[7] = (float)tan.y; o[8] = (float)tan.z; o[9] = (float)bv.x; o[10] = (float)bv.y; o[11] = (float)bv.z;
                o[12] = (float)bw.x; o[13] = (float)bw.y; o[14] = (float)bw.z;
            }
        }
        sync();
    }

    // ------------------------------------------------------------------ AMP observations (scenes/SceneImitateAMP.cpp)
    // cSceneImitateAMP::UpdateHist at the action latch (cRLSceneSimChar::PreUpdate -> NewActionUpdate, RLSceneSimChar.cpp:263-275):
    // the history lives in HBM only; called before every update, writes when the controller wants a new action
    DM_DEV void latch_hist(const EnvState<Real>& st, int e, bool act = true) {
        if (act && s.flg[FLG_NEED_ACTION]) {
            Real* h = st.hist + (size_t)e * 2 * m.P;
            for (int i = l; i < m.P; i += LW) { h[i] = s.pose[i]; h[m.P + i] = s.vel[i]; }
        }
        sync();                            // the flag is cleared by lane 0 at the top of update()
    }
    // cSceneImitateAMP::InitHist (SceneImitateAMP.cpp:152-164) after a reset: kin character one control period earlier
    DM_DEV void init_hist(const EnvState<Real>& st, int e) {
        Real* kp = scratch(); Real* kv = scratch() + NP;
        kin_sample(s.clk[CLK_CTRL] - m.query_period, kp, kv);
        Real* h = st.hist + (size_t)e * 2 * m.P;
        for (int i = l; i < m.P; i += LW) { h[i] = kp[i]; h[m.P + i] = kv[i]; }
        sync();
    }
    // one pose block of RecordAMPObsPose (:279-338) for the pose whose kinematics() results are in LDS (s.com = body part
    // positions, cKinTree::CalcBodyPartPos); O = heading rotation of the *current* pose (cKinTree::CalcHeadingRot)
    DM_DEV void amp_pose_block(const Real* pose, const m3& O, Real ground_h, bool sim_pose, float* out) {
        if (l < m.J) {
            const int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li);
            const v3 rpos = ld3(pose);
            if (l == 0) {
                out[0] = (float)(rpos.y - ground_h);
                const q4 q = ldq(pose + 3);                           // cMathUtil::CalcNormalTangent: q * e_y, q * e_x
                v3 nrm = qrot(q, mk3((Real)0, (Real)1, (Real)0)), tan = qrot(q, mk3((Real)1, (Real)0, (Real)0));
                if (m.amp_local_root) { nrm = O * nrm; tan = O * tan; }
                out[1] = (float)nrm.x; out[2] = (float)nrm.y; out[3] = (float)nrm.z; out[4] = (float)tan.x; out[5] = (float)tan.y; out[6] = (float)tan.z;
            } else if (jt == JT_SPHERICAL) {
                const q4 q = ldq(pose + off);
                v3 nrm = qrot(q, mk3((Real)0, (Real)1, (Real)0)), tan = qrot(q, mk3((Real)1, (Real)0, (Real)0));
                float* o = out + m.amp_off[l];
                o[0] = (float)nrm.x; o[1] = (float)nrm.y; o[2] = (float)nrm.z; o[3] = (float)tan.x; o[4] = (float)tan.y; o[5] = (float)tan.z;
            } else if (jt == JT_REVOLUTE) out[m.amp_off[l]] = (float)(sim_pose ? normalize_angle(pose[off]) : pose[off]);
            const int eo = m.amp_ee[l];
            if (eo >= 0) {
                v3 rel = O * (ld3(s.com[l]) - rpos);
                float* o = out + m.amp_ee_base + 3 * eo;
                o[0] = (float)rel.x; o[1] = (float)rel.y; o[2] = (float)rel.z;
            }
        }
    }
    DM_DEV void amp_vel_block(const Real* vel, const m3& O, float* out) {          // RecordAMPObsVel (:340-371)
        if (l == 0) {
            v3 v = ld3(vel), w = ld3(vel + 3);
            if (m.amp_local_root) { v = O * v; w = O * w; }
            out[0] = (float)v.x; out[1] = (float)v.y; out[2] = (float)v.z; out[3] = (float)w.x; out[4] = (float)w.y; out[5] = (float)w.z;
        }
        for (int i = 7 + l; i < m.P; i += LW) out[6 + i - 7] = (float)vel[i];
    }
    // BuildAMPObs (:259-277): [pose_t, pose_t-1, vel_t, vel_t-1]; (pp, pv) = history, (p, v) = current, all in LDS
    DM_DEV void amp_build(const Real* pp, const Real* pv, const Real* p, const Real* v, Real ground_h, bool sim_pose, float* out) {
        const m3 O = rot_y(-calc_heading(ldq(p + 3)));
        const v3 zero = zero3();
        const int ps = m.amp_pose_size, vs = m.amp_vel_size;
        kinematics(p, v, zero);
        amp_pose_block(p, O, ground_h, sim_pose, out);
        sync();
        kinematics(pp, pv, zero);
        amp_pose_block(pp, O, ground_h, sim_pose, out + ps);
        amp_vel_block(v, O, out + 2 * ps);
        amp_vel_block(pv, O, out + 2 * ps + vs);
        sync();
    }
    // RecordAMPObsAgent (:101-113): sim character now + at the last action latch; ground height of the plane = 0
    DM_DEV void emit_amp(const StepIO<Real>& io, const EnvState<Real>& st, int e) {
        DM_OPAQUE_V(l); DM_OPAQUE_V(li);
        Real* pp = scratch(); Real* pv = scratch() + NP;
        const Real* h = st.hist + (size_t)e * 2 * m.P;
        for (int i = l; i < m.P; i += LW) { pp[i] = h[i]; pv[i] = h[m.P + i]; }
        sync();
        amp_build(pp, pv, s.pose, s.vel, (Real)0, true, io.amp_obs + (size_t)e * 2 * (m.amp_pose_size + m.amp_vel_size));
    }
    // RecordAMPObsExpert (:115-138): raw clip frames (no origin transform, no cycle offset) at `time` and one control
    // period earlier; ground height := the caller's kin origin y.  Uses this wave's LDS record as scratch only.
    DM_DEV void amp_expert(double time, Real ground_h, float* out) {
        Real* p = scratch(); Real* v = scratch() + NP; Real* pp = scratch() + 2 * NP; Real* pv = scratch() + 3 * NP;
        if (l == 0) { s.kin[0] = s.kin[1] = s.kin[2] = 0; s.kin[3] = 1; s.kin[4] = s.kin[5] = s.kin[6] = 0; }
        sync();
        kin_sample<true>(time, p, v);
        kin_sample<true>(time - m.query_period, pp, pv);
        amp_build(pp, pv, p, v, ground_h, false, out);
    }

    // ------------------------------------------------------------------ random perturbations (`--enable_rand_perturbs`)
    // cSceneSimChar::UpdateRandPerturb / ApplyRandForce / GetRandPerturbPartID / ResetRandPertrub (scenes/SceneSimChar.cpp:205-256, 618-626,
    // 952-956), tPerturb + cPerturbManager (sim/Perturb.cpp, sim/PerturbManager.cpp:39-53), applied by cWorld::Update before stepSimulation
    // (sim/World.cpp:93-96) as btMultiBody::addLinkForce at the part's centre of mass (sim/SimBodyLink.cpp:97-112; local_pos = 0).
    // State: one row of doubles per env in HBM (EnvState::pert), touched by lane 0; the forces that act during the current update are
    // handed to the dof lanes through s.sc[0..6] (free inside the update loop).  Draws: dm_rand01(seed, global env id, draw counter,
    // stream 5), the counter kept in the row.  Compiled into the AMP / tap instantiations of the kernels only.
    // (dm_set_draw_tape: the reference's generators; a one-env route, i.e. never the two-per-wave kernel -- compiled out of it)
    DM_DEV double* tape(int e) const { return (LW == kWave && m.draw_tape) ? m.draw_tape + (size_t)e * TP_STRIDE : nullptr; }
    DM_DEV double pert_u01(double* p, int e) const {
        const bool own = p[PT_KSEED] != 0.0;            // (dm_set_env_keys: the env draws the stream of a one-env context of its own)
        const double u = dm_rand01(own ? (uint64_t)(p[PT_KSEED] - 1.0) : m.seed, own ? 0ull : (uint64_t)(e + m.env_off), (uint64_t)p[PT_DRAWS], 5); p[PT_DRAWS] += 1; return u;
    }
    DM_DEV double pert_uniform(double* p, int e, double lo, double hi) const { if (double* T = tape(e)) return tape_uniform(T, 1, lo, hi); const double u = pert_u01(p, e); DM_OPAQUE_D(lo); DM_OPAQUE_D(hi); return (hi > lo && hi < 1e300) ? lo + (hi - lo) * u : hi; }
    // ResetRandPertrub (timer := 0, next := U[time_min, time_max]) and cWorld::Reset -> mPerturbManager.Clear(); lane 0
    DM_DEV void pert_reset(double* p, int e) const {
        p[PT_TIMER] = 0; p[PT_NEXT] = pert_uniform(p, e, m.perturb_time_min, m.perturb_time_max);
        for (int i = 0; i < PT_SLOTS; ++i) p[PT_SLOT0 + i * PT_SLOT_W + PT_LINK] = 0;
    }
    // one scene update: UpdateRandPerturb, then the manager's update inside cWorld::Update (an expired force leaves before it is applied;
    // the others advance their clock and act for the whole update, both substeps); lane 0
    DM_DEV void pert_tick(double* p, int e, double dt) {
        double t = p[PT_TIMER] + dt;
        if (t >= p[PT_NEXT]) {
            const uint32_t mask = m.perturb_part_mask;
            const int n = mask ? dm_popc64((uint64_t)mask) : m.J;
            int idx;                                                                           // cRand::RandInt(0, n)
            if (double* T = tape(e)) idx = tape_int_range(T, 0, n);
            else { idx = (int)(pert_u01(p, e) * n); idx = idx < n ? idx : n - 1; }
            int part = idx;
            if (mask) { uint32_t mm = mask; for (int i = 0; i < idx; ++i) mm &= mm - 1; part = dm_ctz32(mm); }
            const double dx = pert_uniform(p, e, -1, 1), dy = pert_uniform(p, e, -1, 1), dz = pert_uniform(p, e, -1, 1);
            const double mag = pert_uniform(p, e, m.perturb_min, m.perturb_max), dur = pert_uniform(p, e, m.perturb_dur_min, m.perturb_dur_max);
            const double sc = mag / sqrt(dx * dx + dy * dy + dz * dz);
            // a free slot; with none free (excluded at create time: PT_SLOTS * perturb_time_min >= max duration) the one closest to its end
            int slot = 0; double best = -1e300;
            for (int i = 0; i < PT_SLOTS; ++i) {
                const double* ps = p + PT_SLOT0 + i * PT_SLOT_W;
                const double key = (ps[PT_LINK] == 0.0) ? 1e300 : ps[PT_TIME] - ps[PT_DUR];
                if (key > best) { best = key; slot = i; }
            }
            double* ps = p + PT_SLOT0 + slot * PT_SLOT_W;
            ps[PT_LINK] = (double)(part + 1); ps[PT_FX] = sc * dx; ps[PT_FY] = sc * dy; ps[PT_FZ] = sc * dz; ps[PT_DUR] = dur; ps[PT_TIME] = 0;
            t = 0; p[PT_NEXT] = pert_uniform(p, e, m.perturb_time_min, m.perturb_time_max);
        }
        p[PT_TIMER] = t;
        int codes = 0;
        for (int i = 0; i < PT_SLOTS; ++i) {
            double* ps = p + PT_SLOT0 + i * PT_SLOT_W;
            int lk = (int)ps[PT_LINK];
            v3 f = zero3();
            if (lk > 0) {
                if (ps[PT_TIME] >= ps[PT_DUR]) { ps[PT_LINK] = 0; lk = 0; }
                else { ps[PT_TIME] += dt; f = mk3((Real)ps[PT_FX], (Real)ps[PT_FY], (Real)ps[PT_FZ]); }
            }
            st3(s.sc + 3 * i, f); codes |= lk << (8 * i);
        }
        s.sc[6] = (Real)codes;
    }
    // generalized force of the acting perturbations on dof k: J_k^T f with the point Jacobian of the part's centre of mass (same
    // (axis, g) records and chain masks as the contact rows)
    DM_DEV Real pert_gen_force(int k) const {
        const int codes = (int)s.sc[6];
        Real q = 0;
        for (int i = 0; i < PT_SLOTS; ++i) {
            const int lk = (codes >> (8 * i)) & 0xff;
            if (lk == 0) continue;
            const int link = lk - 1;
            const v3 f = ld3(s.sc + 3 * i), xd = cross(ld3(s.com[link]) - ld3(s.p[0]), f);
            const uint32_t ch = (k < 32) ? s.mdl.chain_lo[link] : s.mdl.chain_hi[link];
            if ((ch >> (k & 31)) & 1u) q += dot(ld3(&s.dofrec[k][0]), xd) + dot(ld3(&s.dofrec[k][3]), f);
        }
        return q;
    }

    // ------------------------------------------------------------------ goal-conditioned AMP task scenes
    // target_amp (scenes/SceneTargetAMP.cpp) and heading_amp (scenes/SceneHeadingAMP.cpp).  The goal state lives in HBM only
    // (EnvState::goal, one row of doubles per env) and is touched by lane 0; these functions are called from the AMP / tap
    // instantiations of the kernels only, the plain imitate kernel carries none of it.  The reference draws from the scene's
    // std::default_random_engine (cRand); here every draw is dm_rand01(seed, global env id, draw counter, stream 2), the counter
    // kept in the goal row, so that a trajectory depends neither on the batch nor on the partition.
    // eng: which of the reference's generators makes the draw -- 1 the scene's mRand, 0 cMathUtil::gRand; it only matters on the draw tape
    DM_DEV double goal_u01(double* g, int e, int eng = 1) const {
        if (double* T = tape(e)) return tape_u01(T, eng);
        const bool own = g[GS_KON] != 0.0;
        const double u = dm_rand01(own ? (uint64_t)g[GS_KSEED] : m.seed, own ? 0ull : (uint64_t)(e + m.env_off), (uint64_t)g[GS_DRAWS], 2); g[GS_DRAWS] += 1; return u;
    }
    DM_DEV double goal_uniform(double* g, int e, double lo, double hi, int eng = 1) const { if (double* T = tape(e)) return tape_uniform(T, eng, lo, hi); const double u = goal_u01(g, e); DM_OPAQUE_D(lo); DM_OPAQUE_D(hi); return lo + (hi - lo) * u; }       // cRand::RandDouble(min, max)
    DM_DEV double goal_normal(double* g, int e, double mean, double stdev) const {                                              // cRand::RandDoubleNorm: Box-Muller here
        if (double* T = tape(e)) return tape_normal(T, mean, stdev);
        const double u1 = 1.0 - goal_u01(g, e), u2 = goal_u01(g, e);
        return mean + stdev * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    DM_DEV bool target_like() const { return m.scene_goal == 1 || m.scene_goal == 4 || m.scene_goal == 5; }
    // ---- dribble_amp: the ball as target object (SceneDribbleAMP.cpp:422-440, 493-508), lane 0
    DM_DEV void obj_timer_reset(double* g, int e) const { g[GS_OTIMER] = 0; g[GS_OTIMER_MAX] = goal_uniform(g, e, m.obj_time_min, m.obj_time_max, 0); }      // cTimer::Reset draws from gRand
    DM_DEV void reset_tar_objs(double* g, int e) {
        const double r = goal_uniform(g, e, m.min_tar_obj_dist, m.max_tar_obj_dist), theta = goal_uniform(g, e, -3.141592653589793, 3.141592653589793);
        const double px = (double)s.pose[0] + r * cos(theta), pz = (double)s.pose[2] + r * sin(theta);
        double ax = goal_uniform(g, e, -1, 1), ay = goal_uniform(g, e, -1, 1), az = goal_uniform(g, e, -1, 1);   // SetTarObjPos: on the ground, random orientation, at rest
        if (tape(e)) { const double t = ax; ax = az; az = t; }      // tVector(Rand, Rand, Rand, 0) (:497): the reference's compiler (gcc) evaluates the constructor's arguments right to left
        const double an = sqrt(ax * ax + ay * ay + az * az), th = goal_uniform(g, e, -3.141592653589793, 3.141592653589793, 0);      // (rot_theta: cMathUtil::RandDouble, :498)
        Real* ob = s.obj;
        const Real sh = dm_sin((Real)0.5 * (Real)th), ch = dm_cos((Real)0.5 * (Real)th);
        ob[C::OBJ ? OB_PX : 0] = (Real)px; ob[C::OBJ ? OB_PX + 1 : 0] = m.ball_radius; ob[C::OBJ ? OB_PX + 2 : 0] = (Real)pz;
        ob[C::OBJ ? OB_QW : 0] = ch; ob[C::OBJ ? OB_QW + 1 : 0] = sh * (Real)(ax / an); ob[C::OBJ ? OB_QW + 2 : 0] = sh * (Real)(ay / an); ob[C::OBJ ? OB_QW + 3 : 0] = sh * (Real)(az / an);
        for (int k = 0; k < 6; ++k) ob[C::OBJ ? OB_VX + k : 0] = 0;
        g[GS_PBX] = (double)ob[C::OBJ ? OB_PX : 0]; g[GS_PBY] = (double)ob[C::OBJ ? OB_PX + 1 : 0]; g[GS_PBZ] = (double)ob[C::OBJ ? OB_PX + 2 : 0];      // ResetAgentTarObjRecord
    }
    // cSceneDribbleAMP::Reset (:160-167) runs these BEFORE the scene reset: the ball lands around the root of the previous episode's last state
    DM_DEV void obj_reset(const EnvState<Real>& st, int e) {
        if (C::OBJ && m.scene_goal == 5) {
            if (l == 0) { double* g = st.goal + (size_t)e * GS_WIDTH; obj_timer_reset(g, e); reset_tar_objs(g, e); }
            sync();
        }
    }
    DM_DEV v3 ball_p() const { return ld3(s.obj + (C::OBJ ? OB_PX : 0)); }
    DM_DEV bool dribble_dist_fail(const double* g) const {               // CheckTarObjDistFail / CheckCharObjDistFail (:351-379), part of HasFallen (:343-349)
        const v3 bp = ball_p();
        const Real dx = (Real)g[GS_TX] - bp.x, dz = (Real)g[GS_TZ] - bp.z, t1 = 2 * m.max_target_dist;
        const Real ex = bp.x - s.pose[0], ez = bp.z - s.pose[2], t2 = 2 * (Real)m.max_tar_obj_dist;
        return dx * dx + dz * dz > t1 * t1 || ex * ex + ez * ez > t2 * t2;
    }
    DM_DEV bool dribble_succ(const double* g) const {                    // CheckTargetSucc (:454-464)
        const v3 bp = ball_p();
        const Real dx = (Real)g[GS_TX] - bp.x, dz = (Real)g[GS_TZ] - bp.z;
        return dx * dx + dz * dz < m.target_succ_dist * m.target_succ_dist;
    }
    DM_DEV bool heading_like() const { return m.scene_goal == 2 || m.scene_goal == 3; }
    // cSceneStrikeAMP::SetTargetHit (:249-258): the hit time is the scene clock (cScene::GetTime) of the first hit
    DM_DEV void set_target_hit(double* g, bool hit) const { if (g[GS_AUX0] == 0.0 && hit) g[GS_AUX1] = s.clk[CLK_TIMER]; g[GS_AUX0] = hit ? 1.0 : 0.0; }
    // cSceneTargetAMP::SampleRandTargetPos (:285-299); cSceneStrikeAMP::ResetTargetPos / ...Far / ...Near (:318-374)
    DM_DEV void goal_reset_target_pos(double* g, int e) const {
        if (m.scene_goal == 4) {
            const bool far = goal_u01(g, e) < m.tar_far_prob;
            // (ResetTargetPosFar / Near draw from cMathUtil, the coin above from the scene's generator)
            const double theta = far ? goal_uniform(g, e, -3.141592653589793, 3.141592653589793, 0) : goal_uniform(g, e, m.target_min[0], m.target_max[0], 0);
            const double hgt = goal_uniform(g, e, m.target_min[1], m.target_max[1], 0);
            const double dist = far ? goal_uniform(g, e, m.target_min[2], (double)m.max_target_dist, 0) : goal_uniform(g, e, m.target_min[2], m.target_max[2], 0);
            g[GS_TX] = dist * cos(theta) + (double)s.pose[0]; g[GS_TY] = hgt; g[GS_TZ] = dist * -sin(theta) + (double)s.pose[2];
            set_target_hit(g, false);
            return;
        }
        if (C::OBJ && m.scene_goal == 5) {   // cSceneDribbleAMP::SampleRandTargetPos (:510-524): around the ball
            const double r = goal_uniform(g, e, (double)m.ball_radius, (double)m.max_target_dist), theta = goal_uniform(g, e, -3.141592653589793, 3.141592653589793);
            const v3 bp = ball_p();
            g[GS_TX] = (double)bp.x + r * cos(theta); g[GS_TY] = 0; g[GS_TZ] = (double)bp.z + r * sin(theta);
            return;
        }
        const double dist = goal_uniform(g, e, 0.0, (double)m.max_target_dist), theta = goal_uniform(g, e, 0.0, 6.283185307179586);
        g[GS_TX] = (double)s.pose[0] + dist * cos(theta); g[GS_TY] = 0; g[GS_TZ] = (double)s.pose[2] + dist * sin(theta);
    }
    DM_DEV void goal_timer_reset(double* g, int e) const { g[GS_TIMER] = 0; g[GS_TIMER_MAX] = goal_uniform(g, e, m.goal_time_min, m.goal_time_max, 0); }   // cTimer::Reset, uniform
    // the get-up flag of the env's LDS record follows the goal row (after load, after every change of the timer)
    DM_DEV void goal_sync_flags(const EnvState<Real>& st, int e, bool act = true) {
        if (m.scene_goal == 3) {
            if (act && l == 0) s.getup = !(st.goal[(size_t)e * GS_WIDTH + GS_AUX0] >= m.getup_time) ? 1 : 0;      // CheckGettingUp (:301-304)
            sync();
        }
    }
    // cSceneTargetAMP::Reset after the scene reset (:130-135): mTargetTimer.Reset(); ResetTarget(); and cCtController::SetInitTime /
    // cDeepMimicCharController::ResetParams for the action bookkeeping (mPrevActionTime = time, mPrevActionCOM = 0)
    DM_DEV void goal_reset(const EnvState<Real>& st, int e) {
        if (l == 0) {
            double* g = st.goal + (size_t)e * GS_WIDTH;
            g[GS_AUX0] = 0; g[GS_AUX1] = -1.0;
            goal_timer_reset(g, e);
            goal_reset_target_pos(g, e);
            if (heading_like()) {          // cSceneHeadingAMP::ResetTarget (:230-239)
                g[GS_HEADING] = 0;
                const double sp = goal_uniform(g, e, (double)m.tar_speed_min, (double)m.tar_speed_max);
                g[GS_SPEED] = fmin(fmax(sp, (double)m.tar_speed_min), (double)m.tar_speed_max);
            } else g[GS_SPEED] = (double)m.tar_speed;
            g[GS_PCOMX] = g[GS_PCOMY] = g[GS_PCOMZ] = 0; g[GS_PTIME] = s.clk[CLK_CTRL];
            if (m.scene_goal == 3) {       // ResetTimers -> ResetGetupTimer (ended), then SyncGetupTimer (:190-207): a get-up clip starts mid get-up
                g[GS_AUX0] = m.getup_time;
                if ((m.getup_clip_mask >> (int)g[GS_CLIP]) & 1) g[GS_AUX0] = s.clk[CLK_KIN];
                s.getup = !(g[GS_AUX0] >= m.getup_time) ? 1 : 0;
            }
            if (m.scene_goal == 4) {       // cSceneStrikeAMP::ResetTarget (:301-316): ResetTargetHit (:376-383), then the hit clock
                if (!m.mode_test && m.init_hit_prob > 0) set_target_hit(g, goal_u01(g, e, 0) < m.init_hit_prob);      // cMathUtil::FlipCoin
                g[GS_AUX1] = (g[GS_AUX0] != 0.0) ? goal_uniform(g, e, s.clk[CLK_TIMER] - m.hit_reset_time, s.clk[CLK_TIMER]) : -1.0;
            }
        }
        sync();
    }
    // cSceneHeadingAMPGetup::Reset (:109-121): in train mode an episode that ended in a fall continues, with probability
    // recover_episode_prob, as a recovery episode -- ResetRecoveryEpisode (:40-56) resets the timers and the controller only, the
    // characters stay where they are (`bool mIsRecoveryEpisode = ...` declares a local there: the member stays false and recovery
    // episodes chain).  Returns true (to every lane of the character) when it did that instead of a scene reset.
    DM_DEV bool try_recovery_reset(const EnvState<Real>& st, int e, double max_time) {
        if (!(m.scene_goal == 3 && !m.mode_test && m.recover_prob > 0)) return false;
        sync();
        if (l == 0) {
            double* g = st.goal + (size_t)e * GS_WIDTH;
            bool rec = m.enable_fall_end && has_fallen(nullptr);                     // CheckTerminate(0) == eTerminateFail
            if (rec) rec = goal_u01(g, e) < m.recover_prob;                          // mRand.FlipCoin
            if (rec && !(max_time == max_time)) max_time = tape_time_limit(tape(e), 1);      // (draw tape: ResetRecoveryEpisode -> ResetTimers, one cTimer::Reset)
            if (rec) {
                s.clk[CLK_TIMER] = 0; s.clk[CLK_TIMER_MAX] = max_time;               // ResetTimers
                g[GS_AUX0] = 0; s.getup = 1;                                         // ResetGetupTimer, BeginGetup
                s.clk[CLK_CTRL] = 0; s.clk[CLK_INIT_OFF] = 0;                        // cCtController::ResetParams (CtController.cpp:103-108)
                s.flg[FLG_NEED_ACTION] = 1; s.flg[FLG_EPISODE] += 1;
                g[GS_PCOMX] = g[GS_PCOMY] = g[GS_PCOMZ] = 0; g[GS_PTIME] = 0;
            }
            s.sc[7] = rec ? (Real)1 : (Real)0;
        }
        sync();
        const bool rec = s.sc[7] != (Real)0;
        if (rec) for (int i = l; i < m.D; i += LW) s.tau[i] = 0;
        sync();
        return rec;
    }
    // mass-weighted COM of the links whose kinematics() results are in LDS -> every lane (cSimCharacter::CalcCOM)
    DM_DEV v3 com_of_links() {
        Real* red = scratch();
        if (l < m.J) { const Real mj = s.mdl.mass[l]; red[l * 4] = mj * s.com[l][0]; red[l * 4 + 1] = mj * s.com[l][1]; red[l * 4 + 2] = mj * s.com[l][2]; red[l * 4 + 3] = mj; }
        sync();
        v3 acc = zero3(); Real tm = 0;
        for (int j = 0; j < m.J; ++j) { acc = acc + ld3(red + j * 4); tm += red[j * 4 + 3]; }
        sync();
        return ((Real)1 / tm) * acc;
    }
    DM_DEV v3 link_vcom(int j) const { return ld3(s.vj[j]) + cross(ld3(s.w[j]), ld3(s.com[j]) - ld3(s.p[j])); }   // cSimObj::GetLinearVelocity of a link
    // Before an update: cScene::UpdateTimers for the get-up timer (SceneHeadingAMPGetup.cpp:163-167), then
    // cDeepMimicCharController::HandleNewAction (DeepMimicCharController.cpp:262-267), run by UpdateCalcTau of the first update after
    // an action boundary: mPrevActionTime = mTime (already advanced by this update), mPrevActionCOM = CalcCOM() (state before it)
    DM_DEV void goal_latch(const EnvState<Real>& st, int e, double dt, bool act = true) {
        if (m.scene_goal == 3 && act && l == 0) { double* g = st.goal + (size_t)e * GS_WIDTH; g[GS_AUX0] += dt; s.getup = !(g[GS_AUX0] >= m.getup_time) ? 1 : 0; }
        if (C::OBJ && m.scene_goal == 5 && act && l == 0 && s.flg[FLG_NEED_ACTION]) {      // cSceneDribbleAMP::NewActionUpdate (:337-341)
            double* g = st.goal + (size_t)e * GS_WIDTH; const v3 bp = ball_p(); g[GS_PBX] = (double)bp.x; g[GS_PBY] = (double)bp.y; g[GS_PBZ] = (double)bp.z;
        }
        if (s.flg[FLG_NEED_ACTION]) {          // wave-uniform for one character per wave; per half otherwise (link positions of this state: kin_pre())
            const v3 c = com_of_links();
            if (act && l == 0) { double* g = st.goal + (size_t)e * GS_WIDTH; g[GS_PCOMX] = c.x; g[GS_PCOMY] = c.y; g[GS_PCOMZ] = c.z; g[GS_PTIME] = s.clk[CLK_CTRL] + dt; }
        }
        sync();
    }
    // strike_amp, lane 0, link kinematics of the current state in LDS.  CheckTargetHit (:441-478): a strike body inside the target
    // sphere moving towards the target fast enough; CheckTarContactFail (:485-503): one of the forbidden bodies inside it
    DM_DEV bool strike_hit(const double* g) const {
        const v3 tp = mk3((Real)g[GS_TX], (Real)g[GS_TY], (Real)g[GS_TZ]);
        v3 d = tp - ld3(s.pose); d.y = 0;
        const Real n = norm(d);
        const v3 dir = (n > (Real)1e-5) ? ((Real)1 / n) * d : zero3();
        for (int j = 0; j < m.J; ++j) if ((m.strike_mask >> j) & 1) {
            const v3 dj = tp - ld3(s.com[j]);
            if (dot(dj, dj) < m.target_radius * m.target_radius) { const Real sp = dot(dir, link_vcom(j)); if (sp >= m.hit_tar_speed || m.hit_tar_speed == (Real)0) return true; }
        }
        return false;
    }
    DM_DEV bool strike_contact_fail(const double* g) const {
        const v3 tp = mk3((Real)g[GS_TX], (Real)g[GS_TY], (Real)g[GS_TZ]);
        for (int j = 0; j < m.J; ++j) if ((m.fail_tar_mask >> j) & 1) { const v3 dj = tp - ld3(s.com[j]); if (dot(dj, dj) < m.target_radius * m.target_radius) return true; }
        return false;
    }
    DM_DEV bool strike_succ(const double* g) const { return g[GS_AUX0] != 0.0 && (s.clk[CLK_TIMER] - g[GS_AUX1]) >= m.hit_reset_time; }   // CheckTarHitSucc (:505-520)
    DM_DEV bool goal_dist_fail(const double* g) const {                  // cSceneTargetAMP::CheckTarDistFail (:306-317); heading scenes: false
        if (!target_like()) return false;
        const Real dx = s.pose[0] - (Real)g[GS_TX], dz = s.pose[2] - (Real)g[GS_TZ];
        return dx * dx + dz * dz > m.tar_fail_dist * m.tar_fail_dist;
    }
    // cSceneTargetAMP::Update after the scene update (:137-146) with cSceneHeadingAMP::UpdateTarget (:214-228), cSceneStrikeAMP::UpdateTarget
    // (:290-299), cSceneHeadingAMPGetup::UpdateTestGetup (:244-253); then the goal scenes' part of IsEpisodeEnd for DM_END_EPISODE_EARLY
    DM_DEV void goal_update(const EnvState<Real>& st, int e, double dt, bool act = true) {
        if (m.scene_goal == 4) kinematics(s.pose, s.vel, zero3());             // link positions / velocities of the state after this update
        if (act && l == 0) {
            double* g = st.goal + (size_t)e * GS_WIDTH;
            if (C::OBJ && m.scene_goal == 5) {                                 // cSceneDribbleAMP::UpdateObjs (:310-319), inside the scene update
                g[GS_OTIMER] += dt;
                if (g[GS_OTIMER] >= g[GS_OTIMER_MAX]) { reset_tar_objs(g, e); obj_timer_reset(g, e); }
            }
            g[GS_TIMER] += dt;
            const bool end = g[GS_TIMER] >= g[GS_TIMER_MAX];
            if (end && m.scene_goal != 4) goal_reset_target_pos(g, e);         // EnableRandTargetPos(); cSceneStrikeAMP::CheckTargetReset is false (:385-388)
            if (m.scene_goal == 4 && g[GS_AUX0] == 0.0) set_target_hit(g, strike_hit(g));
            if (heading_like() && end) {
                // UpdateTargetHeading (:182-203, EnableTargetPos() false): sharp turn with probability p, else a Gaussian step
                const bool sharp = goal_u01(g, e) < (double)m.sharp_turn_prob;                 // cRand::FlipCoin
                g[GS_HEADING] += sharp ? goal_uniform(g, e, -3.141592653589793, 3.141592653589793) : goal_normal(g, e, 0.0, (double)m.max_heading_turn_rate);
                // UpdateTargetSpeed (:205-212, EnableRandSpeed() true)
                if (goal_u01(g, e) < (double)m.speed_change_prob) {
                    const double sp = goal_uniform(g, e, (double)m.tar_speed_min, (double)m.tar_speed_max);
                    g[GS_SPEED] = fmin(fmax(sp, (double)m.tar_speed_min), (double)m.tar_speed_max);
                }
            }
            if (end) goal_timer_reset(g, e);
            if (m.scene_goal == 3 && m.mode_test && has_fallen_contact() && !s.getup) { g[GS_AUX0] = 0; s.getup = 1; }     // BeginGetup
            bool over = episode_over_now() || goal_dist_fail(g);
            if (m.scene_goal == 4) over = over || strike_contact_fail(g) || strike_succ(g);
            if (C::OBJ && m.scene_goal == 5) over = over || (m.enable_fall_end && dribble_dist_fail(g)) || dribble_succ(g);
            s.flg[FLG_OVER] = (s.flg[FLG_OVER] & 2) | (over ? 1 : 0);
        }
        sync();
    }
    // RecordGoal + CalcReward + CheckTerminate of the task scenes, after emit() (kinematics of the sim pose is in LDS, emit's flags
    // in io / s.sc[6]).  write_flags = false: the observation pass after an auto reset (goal only).
    DM_DEV void emit_goal(const StepIO<Real>& io, const EnvState<Real>& st, int e, bool write_flags) {
        const v3 com = com_of_links();
        if (l == 0) {
            const double* g = st.goal + (size_t)e * GS_WIDTH;
            const v3 root = ld3(s.pose), tar = mk3((Real)g[GS_TX], (Real)g[GS_TY], (Real)g[GS_TZ]);
            const Real heading = calc_heading(ldq(s.pose + 3));
            const Real tar_speed = (Real)g[GS_SPEED];
            v3 rel = tar - root; rel.y = 0;
            const Real dist_sq = dot(rel, rel);
            const bool dist_fail = target_like() && dist_sq > m.tar_fail_dist * m.tar_fail_dist;      // CheckTarDistFail (:306-317); heading: false
            if (io.goals) {
                float* o = io.goals + (size_t)e * m.goal_dim;
                if (C::OBJ && m.scene_goal == 5) {   // cSceneDribbleAMP::RecordGoal (:276-302): ball -> target in the origin frame
                    const v3 bp = ball_p();
                    v3 rb = tar - bp; rb.y = 0;
                    const Real d = norm(rb);
                    v3 r = mk3((Real)1, (Real)0, (Real)0);
                    if (d > (Real)0.0001) r = ((Real)1 / d) * (rot_y(-heading) * rb);
                    o[0] = (float)r.x; o[1] = (float)r.z; o[2] = (float)d;
                } else if (m.scene_goal == 1) {       // cSceneTargetAMP::RecordGoal (:195-223)
                    const Real d = dm_sqrt(dist_sq);
                    v3 r = mk3((Real)1, (Real)0, (Real)0);
                    if (d > (Real)0.0001) r = ((Real)1 / d) * (rot_y(-heading) * rel);
                    o[0] = (float)r.x; o[1] = (float)r.z; o[2] = (float)d;
                } else if (m.scene_goal == 4) {  // cSceneStrikeAMP::RecordGoal (:414-434): target in the origin frame (BuildOriginTrans), hit phase
                    v3 t = tar - root; t.y = tar.y;
                    t = rot_y(-heading) * t;
                    double ph = 0;
                    if (g[GS_AUX0] != 0.0) ph = fmin(fmax((s.clk[CLK_TIMER] - g[GS_AUX1]) / m.hit_reset_time, 0.0), 1.0);   // CalcHitPhase (:390-401)
                    o[0] = (float)t.x; o[1] = (float)t.y; o[2] = (float)t.z; o[3] = (float)ph;
                } else {                       // cSceneHeadingAMP::RecordGoal (:150-166)
                    Real sh, ch; dm_sincos((Real)g[GS_HEADING] - heading, sh, ch);
                    o[0] = (float)ch; o[1] = (float)-sh; o[2] = (float)tar_speed;
                    // cSceneHeadingAMPGetup::RecordGoal (:123-130): + CalcGetupPhase (:293-299)
                    if (m.scene_goal == 3) o[3] = (float)fmin(fmax(1.0 - g[GS_AUX0] / m.getup_time, 0.0), 1.0);
                }
            }
            if (write_flags) {
                bool fallen = has_fallen(nullptr);
                if (C::OBJ && m.scene_goal == 5) fallen = fallen || dribble_dist_fail(g);      // cSceneDribbleAMP::HasFallen (:343-349)
                const Real step_dur = (Real)(s.clk[CLK_CTRL] - g[GS_PTIME]);
                const v3 dcom = com - mk3((Real)g[GS_PCOMX], (Real)g[GS_PCOMY], (Real)g[GS_PCOMZ]);
                int term = TERM_NULL;          // what the goal scene adds to CheckTerminate when nothing else terminated
                if (dist_fail) term = TERM_FAIL;                                 // cSceneTargetAMP::CheckTerminate (:319-345)
                else if (C::OBJ && m.scene_goal == 5) {                          // the fall test of cRLSceneSimChar::CheckTerminate sees cSceneDribbleAMP::HasFallen
                    if (m.enable_fall_end && dribble_dist_fail(g)) term = TERM_FAIL;
                    else if (dribble_succ(g)) term = TERM_SUCC;                  // cSceneDribbleAMP::CheckTerminateTarget (:466-475)
                }
                else if (m.scene_goal == 4) {                                    // cSceneStrikeAMP::CheckTerminateTarget (:522-541)
                    if (strike_contact_fail(g)) term = TERM_FAIL;
                    else if (strike_succ(g)) term = TERM_SUCC;
                }
                Real r = 0;
                if (m.scene_goal == 1) {       // cSceneTargetAMP::CalcReward (:3-81)
                    if (!dist_fail && !fallen) {
                        const Real pos_reward = dm_exp(-m.pos_reward_scale * dist_sq);
                        Real vel_reward = 0;
                        if (dist_sq < m.target_succ_dist * m.target_succ_dist) vel_reward = 1;
                        else {
                            v3 ct = tar - com; ct.y = 0;
                            const Real cd = norm(ct);
                            v3 dir = zero3();
                            if (cd > (Real)0.0001) dir = ((Real)1 / cd) * ct;
                            const Real avg_vel = dot(dir, dcom) / step_dur;
                            Real vel_err = tar_speed - avg_vel;
                            if (!(avg_vel < 0)) { if (m.enable_min_tar_vel) vel_err = dm_max(vel_err, (Real)0); vel_reward = dm_exp(-((Real)4 / (tar_speed * tar_speed)) * vel_err * vel_err); }
                        }
                        r = (Real)0.6 * pos_reward + (Real)0.4 * vel_reward;
                    }
                } else if (C::OBJ && m.scene_goal == 5) {  // cSceneDribbleAMP::CalcReward (:6-122)
                    const bool ended = s.sc[6] != (Real)0 || term != TERM_NULL;
                    const bool succ = !(m.enable_fall_end && fallen) && term == TERM_SUCC;
                    if (m.mode_test) { if (ended && succ) r = (Real)(s.clk[CLK_TIMER_MAX] - s.clk[CLK_TIMER]); }
                    else if (!fallen) {
                        const v3 bp = ball_p(), pac = mk3((Real)g[GS_PCOMX], (Real)g[GS_PCOMY], (Real)g[GS_PCOMZ]), pbp = mk3((Real)g[GS_PBX], (Real)g[GS_PBY], (Real)g[GS_PBZ]);
                        v3 com_delta = com - pac, cbd = bp - pac; com_delta.y = 0; cbd.y = 0;
                        const Real com_ball_dist = dot(cbd, cbd);
                        const v3 cb_dir = ((Real)1 / dm_sqrt(com_ball_dist)) * cbd;           // Eigen normalized(): no guard in the reference either
                        Real e0 = dm_min((Real)0, dot(cb_dir, com_delta) / step_dur - tar_speed); e0 *= e0;
                        v3 ball_delta = bp - pbp, btd = tar - pbp; ball_delta.y = 0; btd.y = 0;
                        const v3 bt_dir = ((Real)1 / norm(btd)) * btd;
                        Real e2 = dm_min((Real)0, dot(bt_dir, ball_delta) / step_dur - tar_speed); e2 *= e2;
                        const v3 tb = tar - bp;
                        const Real cur_bt = dot(tb, tb);
                        Real r0 = dm_exp(-(Real)1.5 * e0), r1 = dm_exp(-(Real)0.5 * com_ball_dist), r2 = dm_exp(-e2), r3 = dm_exp(-(Real)0.5 * dm_sqrt(cur_bt));
                        if (cur_bt < m.target_succ_dist * m.target_succ_dist && com_ball_dist < (Real)4) r0 = r1 = r2 = r3 = 1;
                        r = (Real)0.1 * r0 + (Real)0.1 * r1 + (Real)0.3 * r2 + (Real)0.5 * r3;
                    }
                } else if (m.scene_goal == 4) {  // cSceneStrikeAMP::CalcReward (:9-187)
                    const bool ended = s.sc[6] != (Real)0 || term != TERM_NULL;
                    const bool succ = !(m.enable_fall_end && fallen) && term == TERM_SUCC;
                    if (m.mode_test) { if (ended && succ) r = (Real)(s.clk[CLK_TIMER_MAX] - s.clk[CLK_TIMER]); }      // CalcRewardTest (:57-72)
                    else if (g[GS_AUX0] != 0.0) r = (Real)1;                                                          // 0.3 + 0.3 + 0.4
                    else if (dist_sq < m.tar_near_dist * m.tar_near_dist) {                                           // CalcRewardTargetNear (:74-112)
                        const Real n = dm_sqrt(dist_sq);
                        const v3 dir = (n > (Real)1e-5) ? ((Real)1 / n) * rel : zero3();
                        Real best = 0;
                        for (int j = 0; j < m.J; ++j) if ((m.strike_mask >> j) & 1) {
                            const v3 dj = tar - ld3(s.com[j]);
                            const Real dr = dm_exp(-m.tar_reward_scale * dot(dj, dj));
                            Real vr = dm_min(dm_max(dot(dir, link_vcom(j)) / m.hit_tar_speed, (Real)0), (Real)1); vr *= vr;
                            best = dm_max(best, (Real)0.2 * dr + (Real)0.8 * vr);
                        }
                        r = (Real)0.3 + (Real)0.3 * best;
                    } else if (!fallen) {                                                                             // CalcRewardTargetFar (:114-187)
                        const Real rt = dm_sqrt(dist_sq), de = dm_max(rt - m.tar_near_dist, (Real)0);
                        const Real pos_reward = dm_exp(-m.pos_reward_scale * de * de);
                        Real vel_reward = 0;
                        v3 ct = tar - com; ct.y = 0;
                        const Real cd = norm(ct);
                        const v3 dir = (cd > (Real)0.0001) ? ((Real)1 / cd) * ct : zero3();
                        const Real avg_vel = dot(dir, dcom) / step_dur;
                        Real vel_err = tar_speed - avg_vel;
                        if (!(avg_vel < 0)) { if (m.enable_min_tar_vel) vel_err = dm_max(vel_err, (Real)0); vel_reward = dm_exp(-((Real)4 / (tar_speed * tar_speed)) * vel_err * vel_err); }
                        r = (Real)0.3 * ((Real)0.7 * pos_reward + (Real)0.3 * vel_reward);
                    }
                } else if (m.scene_goal == 3 && s.getup) {                       // cSceneHeadingAMPGetup::CalcRewardGetup (:19-38)
                    const Real nr = dm_min(dm_max(s.pose[1] / m.getup_height_root, (Real)0), (Real)1);
                    const Real nh = dm_min(dm_max(s.com[m.head_id][1] / m.getup_height_head, (Real)0), (Real)1);
                    r = (Real)0.2 * nr + (Real)0.8 * nh;
                } else if (!fallen) {          // cSceneHeadingAMP::CalcReward (:3-43)
                    Real sh, ch; dm_sincos((Real)g[GS_HEADING], sh, ch);
                    v3 av = ((Real)1 / step_dur) * dcom; av.y = 0;
                    const Real avg_speed = ch * av.x - sh * av.z;
                    if (avg_speed > 0) { Real vel_err = tar_speed - avg_speed; if (m.enable_min_tar_vel) vel_err = dm_max(vel_err, (Real)0); r = dm_exp(-m.vel_reward_scale * vel_err * vel_err); }
                }
                if (io.rewards) io.rewards[e] = (float)r;
                if (term != TERM_NULL) {       // the scene's own termination, only when nothing else terminated
                    if (io.terminate && io.terminate[e] == TERM_NULL) io.terminate[e] = term;
                    if (io.episode_end) io.episode_end[e] = 1;
                    s.sc[6] = (Real)1;
                }
            }
        }
        sync();
    }
    // Multi-clip datasets: a copy of the model whose clip members describe clip c (cClipsController::ActivateMotion)
    DM_DEV ModelDev<Real> model_of_clip(int c) const {
        ModelDev<Real> mc = m;
        if (m.num_clips > 1) {
            const int r0 = m.clip_start[c];
            mc.frame_time = m.frame_time + r0; mc.frames = m.frames + (size_t)r0 * m.P; mc.frame_vel = m.frame_vel + (size_t)r0 * m.P;
            mc.F = m.clip_start[c + 1] - r0; mc.duration = m.clip_dur[c]; mc.loop = m.clip_loop[c];
            for (int k = 0; k < 3; ++k) mc.cycle_delta[k] = m.clip_delta[c * 3 + k];
        }
        return mc;
    }
    // cClipsController::SelectNewMotion (:226-243): upper_bound of a uniform draw in the weight CDF
    DM_DEV int draw_clip(double u) const {
        int c = 0;
        if (m.num_clips > 1) { while (c < m.num_clips - 1 && !(u < m.clip_cdf[c])) ++c; }
        return c;
    }

    // ------------------------------------------------------------------ reset
    // `yaw`: enable_rand_rot_reset (cSceneImitate::ResetKinChar, SceneImitate.cpp:331-349): after Pose(rand_time) with the identity
    // origin the kin character is turned about its root by a random angle about +y (cKinCharacter::RotateOrigin)
    DM_DEV void reset_env(double kin_time, double max_time, Real yaw = (Real)0) {
        DM_OPAQUE_V(l); DM_OPAQUE_V(li);
        Real* kp = scratch(); Real* kv = scratch() + NP; Real* red = scratch() + 2 * NP;
        sync();                            // every lane has read the episode counter / flags the caller derived its arguments from
        if (l == 0) {
            s.clk[CLK_TIMER] = 0; s.clk[CLK_TIMER_MAX] = max_time;
            s.clk[CLK_KIN] = kin_time; s.clk[CLK_CTRL] = kin_time; s.clk[CLK_INIT_OFF] = -kin_time;
            s.flg[FLG_NEED_ACTION] = 1; s.flg[FLG_CONTACT] = 0; s.flg[FLG_VALID] = 1; s.flg[FLG_OVER] = 0; s.flg[FLG_EPISODE] += 1;
            s.kin[0] = s.kin[1] = s.kin[2] = 0; s.kin[3] = 1; s.kin[4] = s.kin[5] = s.kin[6] = 0;
        }
        for (int i = l; i < m.D; i += LW) s.tau[i] = 0;
        sync();
        if (yaw != (Real)0) {
            const v3 rp = kin_root_pos(kin_time);
            sync();
            if (l == 0) kin_rotate_origin(yaw, rp);
            sync();
        }
        kin_sample(kin_time, kp, kv);
        // sim := kin (cSimCharacter::SetPose/SetVel then BuildPose/BuildVel): unit quaternions, spherical w >= 0
        for (int i = l; i < m.P; i += LW) { s.pose[i] = kp[i]; s.vel[i] = kv[i]; }
        sync();
        if (l < m.J) {
            int jt = DM_LI_JTYPE(li), off = DM_LI_POFF(li);
            if (l == 0) { stq(s.pose + 3, qnormalize(ldq(s.pose + 3))); s.vel[6] = 0; if (m.enable_rand_placement) { s.pose[0] = 0; s.pose[2] = 0; } }
            else if (jt == JT_SPHERICAL) { stq(s.pose + off, qstandardize(qnormalize(ldq(s.pose + off)))); s.vel[off + 3] = 0; }
        }
        sync();
        // ResolveCharGroundIntersect: lift the root so that every link AABB clears the ground by 1 mm
        kinematics(s.pose, s.vel, zero3());
        Real viol = 0;
        if (l < m.J) {
            const Real* he = m.aabb_he + l * 4; m3 Rb = ldm3(Rbp(l));
            Real ext = (he[3] != 0) ? he[0] : dm_abs(Rb.m[3]) * he[0] + dm_abs(Rb.m[4]) * he[1] + dm_abs(Rb.m[5]) * he[2];
            viol = dm_min((Real)0, s.com[l][1] - ext - (Real)0.001);
        }
        red[l] = viol;
        sync();
        if (l == 0) {
            Real mv = 0; for (int j = 0; j < m.J; ++j) mv = dm_min(mv, red[j]);
            if (mv < 0) s.pose[1] += -mv;
            // SyncKinCharRoot (SceneImitate.cpp:401-418): kin heading := sim heading (sync_char_root_rot), kin root := sim root
            if (m.sync_root_rot) kin_rotate_origin(calc_heading(ldq(s.pose + 3)) - calc_heading(ldq(kp + 3)), ld3(kp));
            for (int k = 0; k < 3; ++k) s.kin[k] += s.pose[k] - kp[k];
        }
        sync();
    }
};

// Reset of an env of a goal scene / multi-clip dataset (cSceneTargetAMP::Reset -> cSceneImitate::ResetKinChar with a
// cClipsController and enable_rand_rot_reset): the clip is drawn by weight (stream 3 of the reset generator), the clip time
// uniformly in that clip (stream 0), the yaw uniformly in [-pi, pi) (stream 4); the reset itself runs on a copy of the model whose
// clip members describe the drawn clip.  kin_time != null: caller-given clip time (clip 0 unless the goal row names one).
// `pert`: the env's perturbation row (ResetRandPertrub is part of the scene reset).  With a draw tape bound (and no clip time handed in) every draw of
// the reset is looked up in the reference's call order on its two generators: cSceneDribbleAMP::Reset (:160-167: object timer, ball) -> cScene::Reset ->
// cRLSceneSimChar::ResetScene (4 x cTimer::Reset; max_time NaN = draw them here) -> ResetRandPertrub -> cSceneImitate::ResetKinChar (:331-349: the clip
// time over the duration of the clip that was active BEFORE the reset, then cClipsController::Reset picks the new clip, then the random yaw) ->
// cSceneTargetAMP::Reset (:125-130: target timer, ResetTarget).
template <typename Real, typename C, bool TAPS, int LW>
DM_DEV void reset_goal_env(EnvSim<Real, C, TAPS, LW>& sim, const ModelDev<Real>& m, Lds<Real, C>& lds, const EnvState<Real>& st, int e, uint64_t ep,
                           const double* kin_time, double max_time, bool act = true, double* pert = nullptr) {
    const double* grow = st.goal + (size_t)e * GS_WIDTH;
    double* T = kin_time ? nullptr : sim.tape(e);
    int clip; double kt; Real yaw;
    if (T) {
        sim.obj_reset(st, e);
        if (act && sim.l == 0) {
            double mt = max_time;
            if (!(mt == mt)) mt = tape_time_limit(T, 4);
            if (pert) sim.pert_reset(pert, e);
            const double dur_prev = sim.model_of_clip((int)grow[GS_CLIP]).duration;
            lds.clk[0] = tape_uniform(T, 0, 0.0, dur_prev);                                             // CalcRandKinResetTime (:494-500)
            lds.clk[2] = (T[TP_CLIPDRAW] != 0.0) ? (double)sim.draw_clip(tape_uniform(T, 0, 0.0, 1.0)) : 0.0;
            lds.clk[1] = m.enable_rand_rot_reset ? tape_uniform(T, 1, -3.141592653589793, 3.141592653589793) : 0.0;
            lds.clk[4] = mt;
        }
        sim.sync();
        kt = lds.clk[0]; yaw = (Real)lds.clk[1]; clip = (int)lds.clk[2]; max_time = lds.clk[4];
        sim.sync();
    } else {
        const bool own = grow[GS_KON] != 0.0;              // (dm_set_env_keys)
        const uint64_t gid = own ? 0ull : (uint64_t)(e + m.env_off), ksd = own ? (uint64_t)grow[GS_KSEED] : m.seed;
        clip = kin_time ? 0 : sim.draw_clip(dm_rand01(ksd, gid, ep, 3));
        // the clip time is drawn over the duration of the clip that was active BEFORE the reset (cSceneImitate::ResetKinChar draws it first, SceneImitate.cpp:331-335;
        // cKinCharacter::Reset -> cClipsController::Reset selects the new clip afterwards); before the first reset that is the clip cClipsController::Init selected (stream 6)
        const int prev = (ep == 0) ? sim.draw_clip(dm_rand01(ksd, gid, 0, 6)) : (int)grow[GS_CLIP];
        kt = kin_time ? *kin_time : sim.model_of_clip(prev).duration * dm_rand01(ksd, gid, ep, 0);
        yaw = (m.enable_rand_rot_reset && !kin_time) ? (Real)(-3.141592653589793 + 6.283185307179586 * dm_rand01(ksd, gid, ep, 4)) : (Real)0;
        sim.obj_reset(st, e);                      // dribble_amp: the ball first, around the OLD root (cSceneDribbleAMP::Reset)
    }
    const ModelDev<Real> mc = sim.model_of_clip(clip);
    EnvSim<Real, C, TAPS, LW> rs(mc, lds, sim.l);
    rs.li = sim.li;
    rs.reset_env(kt, max_time, yaw);
    sim.manif_clear(st, e);
    if (st.hist) rs.init_hist(st, e);
    if (act && sim.l == 0) st.goal[(size_t)e * GS_WIDTH + GS_CLIP] = (double)clip;
    sim.clip = clip;
    if (m.scene_goal) sim.goal_reset(st, e);
    if (!T && pert && act && sim.l == 0) sim.pert_reset(pert, e);
}

// ============================================================================ kernels
// grid = number of envs, block = one wavefront.
// Occupancy target of the step kernel: fp32 biped 4 waves / SIMD (<= 128 VGPRs, <= 10 KB LDS: 4096 envs are resident at
// once on 256 CUs x 4 SIMDs), fp32 large class 2, fp64 parity builds unconstrained.
template <typename Real, typename C> struct StepWaves { static constexpr int value = 1; };
template <> struct StepWaves<float, ClsBiped> { static constexpr int value = 4; };
template <> struct StepWaves<float, ClsLarge> { static constexpr int value = 2; };
#ifndef DM_LT_WAVES
#define DM_LT_WAVES 2
#endif
template <> struct StepWaves<float, ClsLargeTree> { static constexpr int value = DM_LT_WAVES; };
#ifndef DM_BT_WAVES
#define DM_BT_WAVES 4
#endif
template <> struct StepWaves<float, ClsBipedTree> { static constexpr int value = DM_BT_WAVES; };
#ifndef DM_OBJ_WAVES
#define DM_OBJ_WAVES 2
#endif
template <> struct StepWaves<float, ClsBipedObj> { static constexpr int value = DM_OBJ_WAVES; };
#ifdef DM_EMU
#define DM_WAVES_PER_EU(n)
#else
#define DM_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif
// cTimer::Reset (util/Timer.cpp:55-73) from the counter-based stream 1 of (seed, global env id, episode): uniform U[min, max], or -- `--timer_type exp`,
// EXP instantiations only (the AMP / tap step kernels, which the host selects for such a scene, and the reset kernel) --
// min(min + Exp(rate 1 / timer_exp), max) with std::exponential_distribution's inverse-CDF form -ln(1 - u) / rate.  A pinned limit (min == max: test
// mode, every shipped imitate arg file) draws nothing.
template <bool EXP, typename Real> DM_DEV double draw_time_limit(const ModelDev<Real>& m, int e, uint64_t ep, const double* grow = nullptr) {
    if (!(m.time_lim_max > m.time_lim_min)) return m.time_lim_max;
    const bool own = grow && grow[GS_KON] != 0.0;      // (goal row of an env with a draw key of its own, dm_set_env_keys)
    const double u = dm_rand01(own ? (uint64_t)grow[GS_KSEED] : m.seed, own ? 0ull : (uint64_t)(e + m.env_off), ep, 1);
    if (EXP) { if (m.timer_exp > 0) { const double t = m.time_lim_min - m.timer_exp * log1p(-u); return t < m.time_lim_max ? t : m.time_lim_max; } }
    return m.time_lim_min + (m.time_lim_max - m.time_lim_min) * u;
}
// AMP: the `--scene imitate_amp` instantiation (pose history latch inside the update loop, AMP observation at the end); the
// plain production kernel carries none of it.  The tap build (tests, profiling) serves both scene kinds.
// PHYS2: DM-physics v2 -- its own instantiation, so that the AMP kernels of the shipped scenes do not carry it
template <typename Real, typename C, bool TAPS, bool AMP = false, bool PHYS2 = false>
__global__ void __launch_bounds__(64) DM_WAVES_PER_EU((StepWaves<Real, C>::value)) k_env_step(ModelDev<Real> m, EnvState<Real> st, StepIO<Real> io, DebugTaps<Real> dbg) {
    constexpr bool HIST = TAPS || AMP, V2 = TAPS || PHYS2;
    __shared__ Lds<Real, C> lds;
    const int e = io.env_ids ? io.env_ids[blockIdx.x] : dm_wg_unit(), l = threadIdx.x;      // (env_ids: a subset of the ctx's envs, dm_step_envs)
    EnvSim<Real, C, TAPS> sim(m, lds, l);
    if (TAPS && dbg.prof) sim.prof_begin(dbg.prof + (size_t)e * 16);
    sim.load(st, e);
    if (io.open_loop) sim.set_action_from_clip();
    else if (io.actions) sim.set_action(io.actions + (size_t)(io.env_ids ? (int)blockIdx.x : e) * m.A);
    sim.mark(15);
    Real* aovf = st.aovf ? st.aovf + (size_t)e * (kMaxRows - C::RREG) * kWave : nullptr;
    const bool goal = HIST && st.goal && m.scene_goal;
    if (HIST && st.goal) sim.clip = (int)st.goal[(size_t)e * GS_WIDTH + GS_CLIP];
    double* pert = (HIST && st.pert) ? st.pert + (size_t)e * PT_WIDTH : nullptr;      // enable_rand_perturbs
    Real* manif = (V2 && st.manif) ? st.manif + (size_t)e * m.J * MF_STRIDE : nullptr;   // physics 2
    if (goal) sim.goal_sync_flags(st, e);
    for (int u = 0; u < io.n_updates; ++u) {
        // (AMP / tap builds: the env id is re-materialised per update, so the per-env HBM addresses of the rare paths -- history row, goal
        // row, perturbation state -- and the keys of their counter-based draws are formed where they are used instead of living across the loop)
        int eo = e; if (HIST) DM_OPAQUE_S(eo);
        double* po = (HIST && st.pert) ? st.pert + (size_t)eo * PT_WIDTH : nullptr;
        sim.kin_pre();
        if (io.end_early && (lds.flg[FLG_OVER] & 2)) break;     // invalid since the previous update: the driver ends the episode there
        if (HIST && st.hist) sim.latch_hist(st, eo);
        if (goal) sim.goal_latch(st, eo, io.dt);
        sim.template update<HIST, V2>(io.dt, dbg, eo, aovf, po, manif, true);
        if (goal) sim.goal_update(st, eo, io.dt);
        if (io.end_early && lds.flg[FLG_OVER]) break;           // wave-uniform: latched by lane 0 at the end of update()
    }
    if (io.emit) {
        // pass 0 writes reward / flags / observation.  With auto-reset (mirrors DeepMimic.py:70-79) an env whose episode
        // ended starts its next episode and pass 1 hands back the observation the agent needs for its first action
        // (RecordState after Reset); one copy of the emit code serves both.
        DebugTaps<Real> tap = dbg;
        for (int pass = 0; pass < 2; ++pass) {
            sim.emit(io, tap, e, pass == 0);
            if (goal) sim.emit_goal(io, st, e, pass == 0);                                // task reward / target-distance failure / RecordGoal
            const bool ended = lds.sc[6] != (Real)0;
            if (HIST && pass == 0 && io.amp_obs && st.hist) sim.emit_amp(io, st, e);       // end-of-path observation of a finished episode included
            if (pass == 1 || !(io.auto_reset && ended)) break;
            uint64_t ep = (uint64_t)lds.flg[FLG_EPISODE];
            double mt = draw_time_limit<HIST>(m, e, ep, (HIST && st.goal) ? st.goal + (size_t)e * GS_WIDTH : nullptr);
            bool rec = false;
            if (HIST && st.goal) {           // clip by weight, random yaw, goal reset -- unless the episode goes on as a recovery episode
                rec = sim.try_recovery_reset(st, e, mt);
                if (!rec) reset_goal_env<Real, C, TAPS>(sim, m, lds, st, e, ep, nullptr, mt, true, pert);      // (ResetScene -> ResetRandPertrub; a recovery episode only resets the timers)
            }
            else {
                double kt = m.duration * dm_rand01(m.seed, (uint64_t)(e + m.env_off), ep, 0);
                sim.reset_env(kt, mt);
                if (V2) sim.manif_clear(st, e);
                if (HIST && st.hist) sim.init_hist(st, e);
                if (HIST && pert && l == 0) sim.pert_reset(pert, e);
            }
            tap = DebugTaps<Real>();
        }
        sim.mark(13);
    }
    sim.store(st, e);
    sim.mark(14);
    sim.prof_flush();
}

// reset the envs listed in env_ids (or all when env_ids == null); kin_times / max_times optional per listed env
template <typename Real, typename C>
__global__ void __launch_bounds__(64) k_env_reset(ModelDev<Real> m, EnvState<Real> st, const int* env_ids, const double* kin_times, const double* max_times) {
    __shared__ Lds<Real, C> lds;
    const int b = blockIdx.x, l = threadIdx.x;
    const int e = env_ids ? env_ids[b] : b;
    EnvSim<Real, C> sim(m, lds, l);
    sim.load(st, e);
    uint64_t ep = (uint64_t)lds.flg[FLG_EPISODE];
    // (draw tape bound and no clip time given: NaN = the timer draws are looked up inside the reset, in the reference's order)
    double mt = max_times ? max_times[b]
              : (st.goal && !kin_times && sim.tape(e)) ? (double)NAN
              : draw_time_limit<true>(m, e, ep, st.goal ? st.goal + (size_t)e * GS_WIDTH : nullptr);
    double* pert = st.pert ? st.pert + (size_t)e * PT_WIDTH : nullptr;
    bool rec = false;
    if (st.goal) {
        sim.goal_sync_flags(st, e);
        rec = !kin_times && sim.try_recovery_reset(st, e, mt);
        if (!rec) reset_goal_env<Real, C, true>(sim, m, lds, st, e, ep, kin_times ? &kin_times[b] : nullptr, mt, true, pert);
    } else {
        double kt = kin_times ? kin_times[b] : m.duration * dm_rand01(m.seed, (uint64_t)(e + m.env_off), ep, 0);
        sim.reset_env(kt, mt);
        sim.manif_clear(st, e);
        if (st.hist) sim.init_hist(st, e);
        if (pert && l == 0) sim.pert_reset(pert, e);
    }
    sim.store(st, e);
}

// observation / reward / flags for the current state without stepping (RecordState, CalcReward, CheckTerminate)
template <typename Real, typename C>
__global__ void __launch_bounds__(64) k_env_query(ModelDev<Real> m, EnvState<Real> st, StepIO<Real> io, DebugTaps<Real> dbg) {
    __shared__ Lds<Real, C> lds;
    const int e = blockIdx.x, l = threadIdx.x;
    EnvSim<Real, C> sim(m, lds, l);
    sim.load(st, e);
    if (st.goal && m.scene_goal) sim.goal_sync_flags(st, e);
    sim.emit(io, dbg, e, true);
    if (st.goal && m.scene_goal) sim.emit_goal(io, st, e, true);
    if (io.amp_obs && st.hist) sim.emit_amp(io, st, e);
}

// n expert AMP observations from the clip (RecordAMPObsExpert): sample b at clip time times[b], ground height ground_h[b] (or 0)
template <typename Real, typename C>
__global__ void __launch_bounds__(64) k_amp_expert(ModelDev<Real> m, const double* times, const double* ground_h, float* out, const int* clips) {
    __shared__ Lds<Real, C> lds;
    const int b = blockIdx.x, l = threadIdx.x;
    EnvSim<Real, C> sim(m, lds, l);
    sim.load_model();
    float* o = out + (size_t)b * 2 * (m.amp_pose_size + m.amp_vel_size);
    if (clips && m.num_clips > 1) {            // SampleExpertMotion with a cClipsController: the sample's own clip
        const ModelDev<Real> mc = sim.model_of_clip(clips[b]);
        EnvSim<Real, C> cs(mc, lds, l);
        cs.li = sim.li;
        cs.amp_expert(times[b], ground_h ? (Real)ground_h[b] : (Real)0, o);
    } else sim.amp_expert(times[b], ground_h ? (Real)ground_h[b] : (Real)0, o);
}

// component taps for parity tests: SPD torque for the stored state / one substep with the stored torque
template <typename Real, typename C>
__global__ void __launch_bounds__(64) k_env_probe(ModelDev<Real> m, EnvState<Real> st, DebugTaps<Real> dbg, int what, double dt) {
    __shared__ Lds<Real, C> lds;
    const int e = blockIdx.x, l = threadIdx.x;
    EnvSim<Real, C> sim(m, lds, l);
    sim.load(st, e);
    // what: 0 SPD torque, 1 one substep of length dt with the latched torque, 2 SPD-model mass matrix / bias force taps only
    sim.template dyn_phase<false, true>(what == 1 ? 1 : 0, (Real)dt, (Real)dt, dbg, e, what == 2, false, st.aovf ? st.aovf + (size_t)e * (kMaxRows - C::RREG) * kWave : nullptr,
                                 nullptr, st.manif ? st.manif + (size_t)e * m.J * MF_STRIDE : nullptr);
    sim.store(st, e);
}

}  // namespace dmk
