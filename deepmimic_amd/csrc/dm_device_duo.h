// Two characters per wavefront ("duo") variant of the step kernel for the biped class.
//
// Lanes 0..31 simulate character 2b, lanes 32..63 character 2b+1 of workgroup b, through one instruction stream:
//   lanes <-> links (15 of 32)            kinematics, Newton-Euler pass, integration, reward/observation  (EnvSim pieces)
//   lanes <-> dofs 3..33 (31 of 32)       mass-matrix rows, Cholesky rows, triangular solves.  Rows 0..2 of H belong to the
//                                         root translation: diag(M, M, M) with zeros left of the diagonal, so their rows of L
//                                         are known in closed form (1/sqrt(M) on the diagonal) and need no lane
//   lanes <-> ground-contact candidates   two passes of 32
//   lanes <-> constraint rows (<= 32)     assembly, Gram matrix on the matrix core (two MFMA chains), projected Gauss-Seidel
// Every VALU instruction therefore advances two characters; cross-lane traffic stays inside a 32-lane half (two v_readlane +
// select per broadcast).  A substep in which either character needs more than 32 rows runs the one-character-per-wave
// routine of dm_device.h on each record in turn (same LDS record layout), so the results do not depend on the pairing.
#pragma once
#include "dm_device.h"
// The borrowed-lane path's Y stash (duo_rows_xd): DM_XD_YSOPQ forms its address inside the path (opaque), DM_XD_YSLANE lays it out [lane][dof] (one address for all entries).
// Either removes the 13 hoisted row addresses (26 kernel-long VGPRs, spilled in the prologue: 17 MB of scratch stores per 4096-env launch) -- and makes the allocator reload
// 12-15 other values INSIDE the update loop (dynamics, factor, collision).  Shipped: both 0 (no scratch access on the hot path; tests/test_build_resources.py).
#ifndef DM_XD_YSOPQ
#define DM_XD_YSOPQ 0
#endif
#ifndef DM_XD_YSUNI
#define DM_XD_YSUNI 0
#endif
#ifndef DM_XD_YSLANE
#define DM_XD_YSLANE 0
#endif
// Wave priorities by phase (round 4; profiles/r04_ab_setprio.json).  Two waves share a SIMD; when both have an instruction ready the arbiter takes the
// higher priority, then the older.  A wave in a phase with much independent work per lane (dynamics, collision, the Gram MFMA chains) has an instruction
// ready almost every cycle and delays the ONE ready instruction of a wave that sits in a dependent chain (factorisation columns, substitutions,
// level-synchronous kinematics, Gauss-Seidel rows) by a few cycles each time.  Raising the chain phases costs the other wave next to nothing -- it fills
// the gaps -- and shortens the chains: sweep 1 / 2 / 3 by load +3.0 %, the other chain phases at 1 another +3.2 % (same-box A/Bs, outputs bit-identical).
#if DM_PRIO
#ifndef DM_PRIO_CHOL
#define DM_PRIO_CHOL 1      // factorisation + the two triangular solves
#define DM_PRIO_Y 1         // Y = L^-1 J^T (row lanes)
#define DM_PRIO_BACK 1      // L^-T (Y lambda), integration
#define DM_PRIO_KIN 1       // level-synchronous link kinematics
#endif
#else
#define DM_PRIO_CHOL 0
#define DM_PRIO_Y 0
#define DM_PRIO_BACK 0
#define DM_PRIO_KIN 0
#endif
#ifndef DM_PRIO_BASE
#define DM_PRIO_BASE 1      // the sweep: 1, 2 above DM_PRIO_LO rows, 3 above DM_PRIO_HI
#define DM_PRIO_MID 2
#endif
#ifndef DM_PRIO_LO
#define DM_PRIO_LO 16      // rows of the heavier character of a pair above which the sweep runs at priority 2 / 3 (median pair: 16)
#define DM_PRIO_HI 22
#endif

namespace dmk {

#ifdef DM_EMU
template <typename T> static inline T half_bcast(T v, int src, int half) { return wave_shfl(v, half * 32 + src); }
template <int SRC, typename T> static inline T half_bcast_c(T v, int half) { return half_bcast(v, SRC, half); }
template <int R, typename T> static inline T half_sel_c(T oldv, T newv, int hl, uint32_t) { return hl == R ? newv : oldv; }
template <typename T> static inline T half_sum(T v) {
    T* x = reinterpret_cast<T*>(emu::g_xchg);
    x[threadIdx.x] = v; __syncthreads();
    T s = 0; const int b = (threadIdx.x >> 5) * 32; for (int i = 0; i < 32; ++i) s += x[b + i];
    __syncthreads(); return s;
}
template <int NP2, typename R2, typename Real> static inline void duo_gram32(const R2* y2, Real (&out)[32]) {
    Real* x = reinterpret_cast<Real*>(emu::g_xchg);
    for (int p = 0; p < NP2; ++p) { x[threadIdx.x * 2 * NP2 + 2 * p] = y2[p][0]; x[threadIdx.x * 2 * NP2 + 2 * p + 1] = y2[p][1]; }
    __syncthreads();
    const int b = (threadIdx.x >> 5) * 32;
    for (int i = 0; i < 32; ++i) { Real a = 0; for (int k = 0; k < 2 * NP2; ++k) a += x[threadIdx.x * 2 * NP2 + k] * x[(b + i) * 2 * NP2 + k]; out[i] = a; }
    __syncthreads();
}
#else
__device__ __forceinline__ float half_bcast(float v, int src, int half) { const float a = lane_bcast(v, src), b = lane_bcast(v, 32 + src); return half ? b : a; }
__device__ __forceinline__ double half_bcast(double v, int src, int half) { const double a = lane_bcast(v, src), b = lane_bcast(v, 32 + src); return half ? b : a; }
__device__ __forceinline__ int half_bcast(int v, int src, int half) { const int a = lane_bcast(v, src), b = lane_bcast(v, 32 + src); return half ? b : a; }
// lane SRC of the own half, SRC static, without SGPR round trips or the LDS crossbar: `row_newbcast` spreads lane SRC & 15 of
// every 16-lane row over its row; then rows 1, 3 take rows 0, 2 (`row_bcast:15`, SRC < 16) or the gfx950 `v_permlane16_swap` of the
// value with itself hands rows 1, 3 down to rows 0, 2 (SRC >= 16).  2-3 VALU with a ~20-cycle dependent chain, against two
// v_readlane + two v_mov + a select (5 VALU) or one ds_swizzle (1 LDS op, ~50 cycles): the Gauss-Seidel row chain wants both few
// instructions and a short chain.
template <int SRC> __device__ __forceinline__ float half_bcast_c(float v, int half) {
    const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + (SRC & 15), 0xf, 0xf, true);      // row_newbcast
    if (SRC < 16) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false));          // row_bcast:15 into rows 1, 3
    return __int_as_float(__builtin_amdgcn_permlane16_swap(x, x, false, false)[1]);
}
template <int SRC> __device__ __forceinline__ double half_bcast_c(double v, int half) { return half_bcast(v, SRC, half); }
// lanes R and R + 32 take `newv`: the select mask is an immediate SGPR pair (two s_mov on the scalar unit) instead of a v_cmp per row
template <int R> __device__ __forceinline__ float half_sel_c(float oldv, float newv, int hl, uint32_t one) {
    const uint32_t m32 = one << R;           // `one` is re-made opaque per sweep, or the 32 masks are hoisted out of the loop and spilled
    const uint64_t mk = ((uint64_t)m32 << 32) | m32;
    float out;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(oldv), "v"(newv), "s"(mk));
    return out;
}
template <int R> __device__ __forceinline__ double half_sel_c(double oldv, double newv, int hl, uint32_t) { return hl == R ? newv : oldv; }
__device__ __forceinline__ float half_sum(float v) {
    v += DM_DPP_F(v, 0xB1, 0xf, true);     // quad_perm [1,0,3,2]
    v += DM_DPP_F(v, 0x4E, 0xf, true);     // quad_perm [2,3,0,1]
    v += DM_DPP_F(v, 0x141, 0xf, true);    // row_half_mirror
    v += DM_DPP_F(v, 0x140, 0xf, true);    // row_mirror: every lane holds the sum of its row of 16
    return v + __shfl_xor(v, 16, 64);
}
__device__ __forceinline__ double half_sum(double v) {
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Gram rows of both characters: lane l gets G[i] = sum_k y_l[k] y_{32*half + i}[k], i < 32.  Two MFMA chains share one
// cross-half shuffle per k-pair (the lower half sends its odd column up, the upper half its even column down).
template <int NP2> __device__ __forceinline__ void duo_gram32(const VecT<float>::v2* y2, float (&out)[32]) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
        // gfx950 v_permlane32_swap: (a0, a1) -> ([a0.lo, a1.lo], [a0.hi, a1.hi]): the lower character's column pair / the upper one's
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(y2[p][0]), __float_as_uint(y2[p][1]), false, false);
        const float opA = __uint_as_float(sw[0]), opB = __uint_as_float(sw[1]);
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(opA, opA, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_32x32x2f32(opB, opB, accB, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(accA[v]), __float_as_uint(accB[v]), false, false);
        out[8 * (v / 4) + (v % 4)] = __uint_as_float(sw[0]);        // lower: own accA, upper: the lower half's accB
        out[8 * (v / 4) + 4 + (v % 4)] = __uint_as_float(sw[1]);    // lower: the upper half's accA, upper: own accB
    }
}
template <int NP2> __device__ __forceinline__ void duo_gram32(const VecT<double>::v2* y2, double (&out)[32]) {
    const int half = threadIdx.x >> 5;
#pragma unroll 1
    for (int i = 0; i < 32; ++i) {
        double a = 0;
#pragma unroll
        for (int p = 0; p < NP2; ++p) a += y2[p][0] * half_bcast(y2[p][0], i, half) + y2[p][1] * half_bcast(y2[p][1], i, half);
        out[i] = a;
    }
}
#endif

// ------------------------------------------------------------------ lane borrowing (round 6)
// A pair in which ONE character has more than 32 constraint rows and the two together at most 64 -- under a random actor: two flat feet (8 contacts) plus two or
// three self contacts = 34 / 37 rows, beside a partner with 11 on average; 100 % of the pairs that used to fall back in the closed-loop spinkick run
// (profiles/r06_closed_loop_spinkick.json) -- stays in ONE instruction stream: row 32 + i of the heavy character is held by lane 31 - i of the OTHER half (the
// light character's rows end below those lanes: R_light <= 32 - borrowed), a position that depends on the heavy character's own row count only.  Each lane then
// works on the record of the character whose row it holds (contact slots, dof records, factor rows: per-lane LDS addresses); the Gram rows of all 64 lanes come off
// the matrix core (two half passes of wave_gram64_half) and every lane keeps the 64 entries of ITS character's row order; the sweep visits up to 64 rows, a visit
// hands each character its own delta (two v_readlane + one select); Y lambda is reduced per half as ever, the borrowed lanes' part in a second pass that crosses
// the halves once.  One two-per-wave pass with a longer sweep instead of the failed pass plus two 64-lane passes.  Same arithmetic as the 64-lane routine up to the
// summation order of Y lambda over the rows; deterministic and independent of the partner.
// Register budget: inlined into a kernel that has 252 of 256 VGPRs in use elsewhere, the routine must stay far below that or kernel-long values go to scratch with
// reloads inside the factorisation (first version: 64-entry row file + Y + two MFMA chains + outputs = 248 VGPRs on its own, 42 spilled in the kernel; as a real
// function -- s_swappc -- the values live across the call site were spilled instead, again with reloads on the hot path).  So: Y waits in the pair's overflow block
// (HBM / L2, 34 coalesced stores and loads) while the sweep runs, the Gram comes off the matrix core one 16-accumulator chain at a time, and the row file holds 48
// entries (a heavy character of up to 48 rows: 14 contacts; beyond that the pair falls back as before).
#ifndef DM_DUO_XD
#define DM_DUO_XD 1
#endif
#ifdef DM_EMU
#define DM_REGION_MARK(n) ((void)0)
#else
#define DM_REGION_MARK(n) asm volatile("s_nop " #n)
#endif
#ifndef DM_XD_ROWS
#define DM_XD_ROWS 48
#endif
template <int N, int MASK, int NP2, typename Real> DM_DEV void xd_tr_stage(Real (&w)[NP2], int hl) {
    const bool bit = (hl & MASK) != 0;
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; ++i) {
        Real a = w[2 * i], b = (2 * i + 1 < N) ? w[2 * i + 1] : (Real)0;
        Real keep = bit ? b : a, send = bit ? a : b;
        w[i] = keep + wave_shfl_xor_c<MASK>(send);
    }
}
// one 32 x 32 block of Y^T Y on the matrix core: rows = the rows held by the lanes of half X, columns = those of half Y; f(r, value) is called for r = 0..31 with
// G[row held by lane 32 X + r][row held by lane 32 Y + (lane & 31)] -- every lane gets column (lane & 31) of the block
#ifdef DM_EMU
template <int NP2, typename R2> static inline void xd_gram_operands(R2*) {}
template <int NP2, int X, int Y, typename R2, typename F> static inline void xd_gram_block(const R2* y2, F&& f) {
    typedef decltype(y2[0][0] + 0) Real;
    Real* x = reinterpret_cast<Real*>(emu::g_xchg);
    for (int p = 0; p < NP2; ++p) { x[threadIdx.x * 2 * NP2 + 2 * p] = y2[p][0]; x[threadIdx.x * 2 * NP2 + 2 * p + 1] = y2[p][1]; }
    __syncthreads();
    Real out[32];
    const int me = 32 * Y + (threadIdx.x & 31);
    for (int i = 0; i < 32; ++i) { Real a = 0; for (int k = 0; k < 2 * NP2; ++k) a += x[me * 2 * NP2 + k] * x[(32 * X + i) * 2 * NP2 + k]; out[i] = a; }
    __syncthreads();
    static_for<0, 32>([&](auto rc) { f(rc, out[decltype(rc)::value]); });
}
#else
// (float: `ops` = the column pairs after xd_gram_operands -- [0] = [Y[k0][0..31] | Y[k1][0..31]], [1] = [Y[k0][32..63] | Y[k1][32..63]] -- made ONCE for the four blocks and in
// place of Y, which waits in the overflow block meanwhile: Y and its swapped copy alive together were 34 registers too many)
template <int NP2> __device__ __forceinline__ void xd_gram_operands(VecT<float>::v2* y2) {
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(y2[p][0]), __float_as_uint(y2[p][1]), false, false);
        y2[p][0] = __uint_as_float(sw[0]); y2[p][1] = __uint_as_float(sw[1]);
    }
}
template <int NP2> __device__ __forceinline__ void xd_gram_operands(VecT<double>::v2*) {}
template <int NP2, int X, int Y, typename F> __device__ __forceinline__ void xd_gram_block(const VecT<float>::v2* ops, F&& f) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < NP2; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ops[p][X], ops[p][Y], acc, 0, 0, 0);
    static_for<0, 16>([&](auto vc) {
        constexpr int v = decltype(vc)::value;
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[v]), false, false);      // both halves: the lower lanes' rows, the upper lanes' rows of the column
        f(std::integral_constant<int, 8 * (v / 4) + (v % 4)>{}, __uint_as_float(sw[0]));
        f(std::integral_constant<int, 8 * (v / 4) + 4 + (v % 4)>{}, __uint_as_float(sw[1]));
    });
    DM_SCHED_FENCE();
}
template <int NP2, int X, int Y, typename F> __device__ __forceinline__ void xd_gram_block(const VecT<double>::v2* y2, F&& f) {
    const int me = 32 * Y + (threadIdx.x & 31);
    static_for<0, 32>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        double a = 0;
#pragma unroll
        for (int p = 0; p < NP2; ++p) a += wave_shfl(y2[p][0], me) * lane_bcast(y2[p][0], 32 * X + r) + wave_shfl(y2[p][1], me) * lane_bcast(y2[p][1], 32 * X + r);
        f(rc, a);
    });
}
#endif
template <typename Real, bool V2>
DM_DEV void duo_rows_xd(Lds<Real, ClsBiped>* rec, int wl, Real h, int nc, int R, int Ra, int Rb, int D, int NL, Real erp, Real friction, Real lim_max_impulse, int solver_iters, Real* ystash) {
    typedef ClsBiped C; typedef Lds<Real, C> L; typedef EnvSim<Real, C, false, 32> Base;
    typedef V3<Real> v3; typedef typename VecT<Real>::v2 R2; typedef typename VecT<Real>::v4 R4;
    constexpr int ND = C::ND, NP2 = ND / 2, NJ = C::NJ, HW = 32, XR = DM_XD_ROWS;
    static_assert(XR == 40 || XR == 48 || XR == 64, "RowFile sizes");
    const int half = wl >> 5, hl = wl & 31;
    L& s = rec[half];
    const int H = (Ra > HW) ? 0 : 1;                          // the heavy half (wave-uniform)
    const int Rv = H ? Rb : Ra;                               // rows of the heavy character = visits of the sweep
    const int nb = Rv - HW;                                   // borrowed lanes
    const bool xl = (half != H) && (HW - 1 - hl) < nb;        // this lane holds a row of the OTHER half's character
    const int ch = xl ? H : half;                             // the character whose row this lane holds
    const int row = xl ? HW + (HW - 1 - hl) : hl;             // ... and which row
    const int ncA = lane_bcast(nc, 0), ncB = lane_bcast(nc, 32);
    const int ncc = ch ? ncB : ncA, Rc = NL + 3 * ncc, RNc = NL + ncc;
    L& sc = rec[ch];
    if (hl == 0) { s.flg[FLG_NROWS] = R; s.flg[FLG_NCONT] = nc; s.clk[5] += 1.0; }      // (clk[5]: the pad word of the clock row counts this env's substeps on borrowed lanes, dm_get_debug "borrowed")
    __syncthreads();
    Real brow = 0;
    uint32_t ch_lo = 0, ch_hi = 0, ng_lo = 0, ng_hi = 0; v3 xd = mk3((Real)0, (Real)0, (Real)0), dd = xd;
    if (row < Rc) {
        if (row < NL) {                                       // (DuoSim::substep_post's limit row)
            const int lr = V2 ? (row >> 1) : row;
            int j = sc.mdl.lim_joint[lr]; int lj = sc.mdl.link_info[j]; int off = DM_LI_POFF(lj);
            const int limdof = DM_LI_DOFF(lj);
            Real th = sc.pose[off], pen_lo = th - sc.mdl.lim_lo[lr], pen_hi = sc.mdl.lim_hi[lr] - th;
            Real pen, sgn;
            if (V2 ? !(row & 1) : (pen_lo <= pen_hi)) { sgn = 1; pen = pen_lo; } else { sgn = -1; pen = pen_hi; }
            brow = (pen > 0) ? -pen / h : -erp * pen / h;
            xd = sgn * ld3(&sc.dofrec[limdof][0]);
            if (limdof < 32) ch_lo = 1u << limdof; else ch_hi = 1u << (limdof - 32);
        } else {                                              // (EnvSim::contact_row)
            int slot, kindr;
            if (row < NL + ncc) { slot = row - NL; kindr = 0; } else { int fi = row - NL - ncc; slot = fi >> 1; kindr = 1 + (fi & 1); }
            const Real* ct = sc.ct[slot];
            const int info = (int)ct[7];
            const int la = info & 0xff, lb = (info >> 8) & 0xff;
            const v3 n = ld3(ct + 3);
            v3 t1, t2; Base::plane_space(n, t1, t2);
            dd = (kindr == 0) ? n : ((kindr == 1) ? t1 : t2);
            xd = cross(ld3(ct) - ld3(sc.p[0]), dd);
            uint32_t a_lo = sc.mdl.chain_lo[la < NJ ? la : 0], a_hi = sc.mdl.chain_hi[la < NJ ? la : 0], b_lo = 0, b_hi = 0;
            if (lb != 255) { b_lo = sc.mdl.chain_lo[lb < NJ ? lb : 0]; b_hi = sc.mdl.chain_hi[lb < NJ ? lb : 0]; }
            ch_lo = a_lo ^ b_lo; ch_hi = a_hi ^ b_hi; ng_lo = b_lo & ch_lo; ng_hi = b_hi & ch_hi;
            if (kindr == 0) { const Real dc = ct[6]; brow = (dc > 0) ? -dc / h : -erp * dc / h; }
        }
    }
    // y := L^-1 J^T against this lane's character (EnvSim::substep_post's dense loop with per-lane record addresses)
    R2 y2[NP2]; Real cvec = 0;
#if DM_PRIO_Y
    dm_setprio<DM_PRIO_Y>();
#endif
#if DM_YPREF
    // software-pipelined like DuoSim::substep_post's loop (round 6, second pass): the factor row and the dof record of step k + 1 are requested before the
    // accumulation chain of step k (per-lane addresses here: a lane reads the record of the character whose row it holds); one dof per scheduling region
    R2 lr[2][NP2]; R4 rr[2][2];
#define DM_XD_YLOAD(k)                                                                                        \
    {                                                                                                         \
        rr[(k) & 1][0] = *reinterpret_cast<const R4*>(&sc.dofrec[(k)][0]);                                     \
        rr[(k) & 1][1] = *reinterpret_cast<const R4*>(&sc.dofrec[(k)][4]);                                     \
        const R2* lrow_ = reinterpret_cast<const R2*>(&sc.Lt[L::lrow(k)]);                                     \
        _Pragma("unroll") for (int p = 0; p <= ((k) >> 1); ++p) lr[(k) & 1][p] = lrow_[p];                    \
    }
    DM_XD_YLOAD(0)
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        Real yk = 0;
        if (k + 1 < ND) { if (DM_DUO_YFULL || k + 1 < D) DM_XD_YLOAD(k + 1) }
        if (DM_DUO_YFULL || k < D) {
            const R4 r0 = rr[k & 1][0], r1 = rr[k & 1][1];
            Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
            if (DM_DUO_YFULL) DM_OPAQUE_V(val);
            const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
            const Real raw = on ? (ng ? -val : val) : (Real)0;
            cvec += raw * r1[2];
            R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;
#pragma unroll
            for (int p = 0; p < (k >> 1); ++p) { if (p & 1) acc3 += lr[k & 1][p] * y2[p]; else acc2 += lr[k & 1][p] * y2[p]; }
            acc2 += acc3;
            Real acc = raw - (acc2[0] + acc2[1]);
            if (k & 1) acc -= lr[k & 1][k >> 1][0] * y2[k >> 1][0];
            yk = acc * lr[k & 1][k >> 1][k & 1];
            if (DM_DUO_YFULL) DM_OPAQUE_V(yk);
        }
        y2[k >> 1][k & 1] = yk;
        DM_SCHED_FENCE();
    }
#undef DM_XD_YLOAD
#else
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        Real yk = 0;
        if (k < D) {
            const R4 r0 = *reinterpret_cast<const R4*>(&sc.dofrec[k][0]), r1 = *reinterpret_cast<const R4*>(&sc.dofrec[k][4]);
            const Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
            const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
            const Real raw = on ? (ng ? -val : val) : (Real)0;
            cvec += raw * r1[2];
            R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;
            const R2* lrow = reinterpret_cast<const R2*>(&sc.Lt[L::lrow(k)]);
#pragma unroll
            for (int p = 0; p < (k >> 1); ++p) { if (p & 1) acc3 += lrow[p] * y2[p]; else acc2 += lrow[p] * y2[p]; }
            acc2 += acc3;
            Real acc = raw - (acc2[0] + acc2[1]);
            if (k & 1) acc -= sc.Lt[L::lrow(k) + k - 1] * y2[k >> 1][0];
            yk = acc * sc.Lt[L::lrow(k) + k];
        }
        y2[k >> 1][k & 1] = yk;
        DM_SCHED_FENCE();      // (one dof at a time: unfenced, the scheduler hoists the factor rows of many dofs -- up to 34 registers each -- and spills kernel-long values)
    }
#endif
#if DM_PRIO_Y
    dm_setprio<0>();
#endif
    const bool is_fric = row >= RNc && row < Rc;
    Real lam = 0;
    if (wave_ballot(row < RNc && (brow - cvec) > 0) != 0) {
        Real adiag;
        { R2 a2 = {(Real)0, (Real)0};
#pragma unroll
          for (int p = 0; p < NP2; ++p) a2 += y2[p] * y2[p];
          adiag = a2[0] + a2[1]; }
        const Real inv_adiag = (row < Rc) ? (Real)1 / adiag : (Real)0;
        // entry c of a lane's file = A[own row][row c of the same character]: rows 0..31 are the lanes of the character's half, row 32 + i lane 31 - i of the other one.
        // Four 32 x 32 blocks of Y^T Y, one MFMA chain (16 accumulators) at a time: after chain (rows of half X, columns of half Y) and one swap per accumulator, lanes j and
        // 32 + j both hold column j of block (X, Y), i.e. (symmetry) the entries of the row held by lane 32 Y + j against the rows of half X.
        RowFile<Real, XR> arow;
        // Y leaves for the pair's overflow block until the sweep is over (coalesced: [dof][lane]); its registers then carry the MFMA operands
        // (the lane's base address is formed HERE, opaque to the optimizer: hoisted out of the 20-update loop the 13 row addresses beyond the 12-bit offset range were 26 kernel-long
        // VGPRs of this rare path -- spilled in the prologue once the main path needed the registers: 17 MB of scratch stores per 4096-env launch)
        { Real* ys = (DM_XD_YSUNI ? dm_uniform_ptr(ystash) : ystash) + (DM_XD_YSLANE ? wl * (2 * NP2) : wl); if (DM_XD_YSOPQ) DM_OPAQUE_V(ys);        // DM_XD_YSLANE: [lane][dof] -- every entry within the 12-bit offset range of ONE address
#pragma unroll
          for (int p = 0; p < NP2; ++p) { ys[DM_XD_YSLANE ? 2 * p : (2 * p) * kWave] = y2[p][0]; ys[DM_XD_YSLANE ? 2 * p + 1 : (2 * p + 1) * kWave] = y2[p][1]; } }
        xd_gram_operands<NP2>(y2);
        DM_SCHED_FENCE();
        xd_gram_block<NP2, 0, 0>(y2, [&](auto rc, Real g) { constexpr int r = decltype(rc)::value; arow.set(r, g); if (XR - 32 > 31 - r) arow.set(32 + (31 - r), g); });       // half-0 lanes against half 0 (provisionally every lane's; the borrowed rows of a heavy half 1)
        xd_gram_block<NP2, 0, 1>(y2, [&](auto rc, Real g) { constexpr int r = decltype(rc)::value; if (half == 1) { if (ch == 0) arow.set(r, g); if (XR - 32 > 31 - r && H == 1) arow.set(32 + (31 - r), g); } });   // half-1 lanes against half 0: a borrowed lane of heavy 0 (its first 32) | the lanes of a heavy character 1 (its borrowed rows)
        xd_gram_block<NP2, 1, 0>(y2, [&](auto rc, Real g) { constexpr int r = decltype(rc)::value; if (half == 0) { if (ch == 1) arow.set(r, g); if (XR - 32 > 31 - r && H == 0) arow.set(32 + (31 - r), g); } });   // half-0 lanes against half 1: a borrowed lane of heavy 1 | the lanes of a heavy character 0
        xd_gram_block<NP2, 1, 1>(y2, [&](auto rc, Real g) { constexpr int r = decltype(rc)::value; if (half == 1) { if (ch == 1) arow.set(r, g); if (XR - 32 > 31 - r && H == 0) arow.set(32 + (31 - r), g); } });   // half-1 lanes against half 1
        // scaled by 1 / A_ll, zero diagonal (the sweep runs on t = lambda + q), zero beyond the character's rows: a visit past a character's last row, or to a
        // borrowed lane's position in the light half, must not move that character
#pragma unroll
        for (int c = 0; c < XR; ++c) arow.set(c, (c == row || c >= Rc) ? (Real)0 : arow.get(c) * inv_adiag);
        Real t = (brow - cvec) * inv_adiag;
        const int nrm_lane = is_fric ? ch * HW + NL + ((row - RNc) >> 1) : 0;      // the contact's normal row: always among the first 32, i.e. in its character's own half
        Real lo = 0, hi = is_fric ? (Real)0 : ((row < NL) ? lim_max_impulse : (Real)1e30);
        const uint32_t fmask = (1u << (NL + ncA)) | (1u << (NL + ncB));
        const int xsrc = H ? 0 : HW;                        // borrowed lanes sit in the half that is not H
#if DM_PRIO
        dm_setprio<3>();
#endif
        // (a variant with scalar-unit lane masks and one v_readlane on the visits past the light character's rows issued two VALU fewer per visit and measured 3 % slower
        // in the closed loop: profiles/r06_ab_xd_sweep.json)
        for (int it = 0; it < solver_iters; ++it) {
            int Rvo = Rv, rowo = row; uint32_t fm = fmask;
            DM_OPAQUE_S(Rvo); DM_OPAQUE_S(fm); DM_OPAQUE_V(rowo);
            static_for<0, XR / 4>([&](auto blkc) {
                constexpr int blk = decltype(blkc)::value;
                if (blk * 4 < Rvo) {
                    static_for<0, 4>([&](auto ic) {
                        constexpr int v = blk * 4 + decltype(ic)::value;
                        if (v < 32 && __builtin_expect((fm >> (v & 31)) & 1u, 0)) { const Real ln = wave_shfl(lam, nrm_lane); if (is_fric && v == RNc) { hi = friction * ln; lo = -hi; } }
                        const Real nl = dm_med3(lo, t, hi);
                        const Real d = nl - lam;
                        Real delta;
                        if (v < 32) { const Real d0 = lane_bcast(d, v & 31), d1 = lane_bcast(d, 32 + (v & 31)); delta = ch ? d1 : d0; }      // each character's own row v
                        else delta = lane_bcast(d, xsrc + (63 - v));                                                                    // the heavy character's borrowed row (the light one's entries are 0)
                        t -= arow.get(v) * delta;
                        lam = (rowo == v) ? nl : lam;
                    });
                }
            });
        }
#if DM_PRIO
        dm_setprio<DM_PRIO_BACK>();
#endif
        if (row >= Rc) lam = 0;
        { const Real* ys = (DM_XD_YSUNI ? dm_uniform_ptr(ystash) : ystash) + (DM_XD_YSLANE ? wl * (2 * NP2) : wl); if (DM_XD_YSOPQ) DM_OPAQUE_V(ys);
#pragma unroll
          for (int p = 0; p < NP2; ++p) { y2[p][0] = ys[DM_XD_YSLANE ? 2 * p : (2 * p) * kWave]; y2[p][1] = ys[DM_XD_YSLANE ? 2 * p + 1 : (2 * p + 1) * kWave]; } }
    } else {
#if DM_PRIO_BACK
        dm_setprio<DM_PRIO_BACK>();
#endif
    }
    // Y lambda per dof and character: the rows in their character's own half, then the borrowed lanes' rows (reduced in the half they sit in, handed across once)
    {
        Real w[NP2], wx[NP2];
        const bool bit = (hl & 1) != 0;
        const Real lam1 = xl ? (Real)0 : lam, lam2 = xl ? lam : (Real)0;
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            const Real a = y2[p][0] * lam1, bb = y2[p][1] * lam1;
            w[p] = (bit ? bb : a) + wave_shfl_xor_c<1>(bit ? a : bb);
            const Real ax = y2[p][0] * lam2, bx = y2[p][1] * lam2;
            wx[p] = (bit ? bx : ax) + wave_shfl_xor_c<1>(bit ? ax : bx);
        }
        xd_tr_stage<NP2, 2, NP2>(w, hl); xd_tr_stage<(NP2 + 1) / 2, 4, NP2>(w, hl); xd_tr_stage<(NP2 + 3) / 4, 8, NP2>(w, hl); xd_tr_stage<(NP2 + 7) / 8, 16, NP2>(w, hl);
        xd_tr_stage<NP2, 2, NP2>(wx, hl); xd_tr_stage<(NP2 + 1) / 2, 4, NP2>(wx, hl); xd_tr_stage<(NP2 + 3) / 4, 8, NP2>(wx, hl); xd_tr_stage<(NP2 + 7) / 8, 16, NP2>(wx, hl);
        const Real z0 = w[0] + wave_shfl(wx[0], wl ^ 32), z1 = w[1] + wave_shfl(wx[1], wl ^ 32);      // (the heavy half's own second pass is all zeros: nothing comes back to the light one)
        Real* xs = &s.Ic[0][0];
        xs[hl] = z0;
        if (hl + 32 < D) xs[hl + 32] = z1;
    }
}

#ifndef DM_DUO_YPMAX
#define DM_DUO_YPMAX 7
#endif
#ifndef DM_DUO_WIDE_FALLBACK
#define DM_DUO_WIDE_FALLBACK 0
#endif
// the fallback class of a pair that leaves the two-per-wave routine: the biped class's GRAM64 variant; for biped + free body the one-per-wave class itself
template <typename CC> struct DuoFallback {
#if DM_DUO_WIDE_FALLBACK
    typedef ClsBipedWide type;
#else
    typedef ClsBipedFb type;
#endif
};
template <> struct DuoFallback<ClsBipedObj> { typedef ClsBipedObj type; };
// CC: ClsBiped, or (round 6) ClsBipedObj -- biped + one free rigid sphere per character (dribble_amp's ball): three more register pairs of y per row lane, the ball's
// contacts in the half's own slots, its velocity change by six half-wave sums, its integration by lane 0 of the half
template <typename Real, bool TAPS, typename CC = ClsBiped>
struct DuoSim {
    typedef CC C;
    typedef Lds<Real, C> L;
    typedef EnvSim<Real, C, TAPS, 32> Base;
    // Fallback class for a pair with a heavily contacted character.  DM_DUO_WIDE_FALLBACK = 1 selects ClsBipedWide (all 64 rows of A
    // in VGPRs, 64-row Gram on the matrix core): +8..13 % closed-loop throughput under an untrained policy, but its 64-register row
    // file pushes 10 kernel-long-lived values of the two-per-wave kernel into scratch (40 B / lane, +7.5 MB of HBM traffic per launch),
    // so the default keeps the narrow class (rows 32..63 in the HBM / L2 overflow block, requested two rows ahead): no scratch at all.
    typedef typename DuoFallback<CC>::type FallbackCls;
    typedef EnvSim<Real, FallbackCls, TAPS, kWave> Single;
    typedef Lds<Real, FallbackCls> WideRec;
    static_assert(sizeof(WideRec) == sizeof(L), "the wide class must share the LDS record layout");
    static constexpr int ND = C::ND, NP2 = ND / 2, NP = C::NP, NJ = C::NJ, HW = 32, CP = 2 /* candidate passes */;
    static constexpr int NP2X = NP2 + (C::OBJ ? 3 : 0);      // + the free body's 3 linear + 3 angular velocities (rows of Y = M^-1/2 J^T)
    typedef V3<Real> v3; typedef M3<Real> m3; typedef typename VecT<Real>::v2 R2; typedef typename VecT<Real>::v4 R4;
    const ModelDev<Real>& m; L* rec;                    // the two records of this workgroup
    const int wl, half; int hl;                         // wave lane, which character, lane within the character
    Base b;                                             // per-character pieces (lane hl of record `half`)
    L& s;
    int cand_link[CP]; Real cand_loc[CP][3], cand_rad[CP];
    static constexpr int PP = C::NPAIRCAP / HW;         // self-collision pair passes
    int pair_code[PP];
    DM_DEV DuoSim(const ModelDev<Real>& m_, L* rec_, int wl_) : m(m_), rec(rec_), wl(wl_), half(wl_ >> 5), hl(wl_ & 31), b(m_, rec_[wl_ >> 5], wl_ & 31), s(rec_[wl_ >> 5]) {}
    DM_DEV void sync() const { __syncthreads(); }
    // the priority of the phases between the dependent chains.  DM_PRIO_LATE > 0: a wave that has left the 32-row path in this launch (borrowed lanes or the 64-lane routine) is one
    // the launch will wait for -- a one-round launch lasts as long as its slowest wave -- and keeps the raised level in the throughput phases too
#ifndef DM_PRIO_LATE
#define DM_PRIO_LATE 2      // same-box A/B, closed-loop spinkick: 1: -0.9 %, 2: -1.6 %, 3: -1.4 % step time; open loop +-0; the chain phases at 3 too: no further gain and +1.1 % open loop (profiles/r06_ab_lane_borrowing.json)
#endif
    int late = 0;
    DM_DEV void prio_low() const {
#if DM_PRIO_LATE
        if (late) dm_setprio<DM_PRIO_LATE>(); else dm_setprio<0>();
#else
        dm_setprio<0>();
#endif
    }
    DM_DEV Real& Lx(int r, int c) const { return s.Lt[L::lrow(r) + c]; }
    static DM_DEV v3 zero3() { return mk3((Real)0, (Real)0, (Real)0); }

    DM_DEV void load(const EnvState<Real>& st, int e) {
        b.load(st, e);
        load_cands();
    }
    // this lane's ground-contact candidates and self-collision pairs (kernel-long register values; read again behind the borrowed-lane path, which needs their registers)
    DM_DEV void load_cands() {
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            const int c = hl + HW * q;
            cand_link[q] = 0; cand_rad[q] = 0; cand_loc[q][0] = cand_loc[q][1] = cand_loc[q][2] = 0;
            if (c < m.NC) { cand_link[q] = m.cand_link[c]; cand_rad[q] = m.cand_rad[c]; for (int k = 0; k < 3; ++k) cand_loc[q][k] = m.cand_loc[c * 3 + k]; }
        }
#pragma unroll
        for (int q = 0; q < PP; ++q) { const int c = hl + HW * q; pair_code[q] = (c < m.NPAIR) ? m.pair_code[c] : -1; }
    }

    // ------------------------------------------------------------------ mass matrix rows 3..33 in lanes, rows 0..2 closed form
#ifndef DM_EMU
    // Subtree (composite-body) sums of both characters on the matrix core, fp32 build.  Every link's Newton-Euler terms are taken
    // about ONE origin o (the root joint), where they simply add over a subtree: X_k = (f, n + e x f, m, m e, I_w + m(|e|^2 1 - e e^T)),
    // e = com_k - o.  S = M X with the 16 x 16 ancestor mask M (M_jk = 1 when k is in the subtree of j) is four
    // v_mfma_f64_16x16x4_f64 per character; each link then moves its sums to its own joint origin (r = p_j - o):
    //   N_j = N0 - r x F,  h_j = H0 - m r,  I_j = I0 + (m r.r - 2 H0.r) 1 + H0 r^T + r H0^T - m r r^T.
    // The shift cancels two digits for distal links (I0 ~ m |r|^2 >> I_j), so X, the sums and the shift are carried in fp64: every
    // product of two fp32 inputs is exact there and the result is the exactly evaluated direct formula, rounded once to fp32.
    // Replaces a 2-lane-per-link gather loop (8 dependent LDS round trips for the root) by ~130 instructions without a loop.
    DM_DEV void subtree_mfma(int iset) {
        typedef double d4 __attribute__((ext_vector_type(4)));
        double* xb = reinterpret_cast<double*>(&s.Lt[0]);           // [16 links][16 quantities]; the factor storage is dead until dyn_row
        const v3 o = ld3(s.p[0]);
        if (hl < m.J) {
            const int k = hl;
            m3 Rb = ldm3(b.Rbp(k));
            const Real* Id = s.mdl.inertia[iset][k];
            const Real I0 = Id[0], I1 = Id[1], I2 = Id[2];
            Real Iw[6];
            Iw[0] = Rb.m[0] * Rb.m[0] * I0 + Rb.m[1] * Rb.m[1] * I1 + Rb.m[2] * Rb.m[2] * I2;
            Iw[1] = Rb.m[0] * Rb.m[3] * I0 + Rb.m[1] * Rb.m[4] * I1 + Rb.m[2] * Rb.m[5] * I2;
            Iw[2] = Rb.m[0] * Rb.m[6] * I0 + Rb.m[1] * Rb.m[7] * I1 + Rb.m[2] * Rb.m[8] * I2;
            Iw[3] = Rb.m[3] * Rb.m[3] * I0 + Rb.m[4] * Rb.m[4] * I1 + Rb.m[5] * Rb.m[5] * I2;
            Iw[4] = Rb.m[3] * Rb.m[6] * I0 + Rb.m[4] * Rb.m[7] * I1 + Rb.m[5] * Rb.m[8] * I2;
            Iw[5] = Rb.m[6] * Rb.m[6] * I0 + Rb.m[7] * Rb.m[7] * I1 + Rb.m[8] * Rb.m[8] * I2;
            const v3 w = ld3(s.w[k]), al = ld3(s.al[k]), com = ld3(s.com[k]);
            const v3 rc = com - ld3(s.p[k]);
            const v3 ac = cross_add(cross_add(ld3(s.aj[k]), al, rc), w, cross(w, rc));
            const Real mk = s.mdl.mass[k];
            const v3 f = mk * ac;
            const v3 Iwv = mk3(Iw[0] * w.x + Iw[1] * w.y + Iw[2] * w.z, Iw[1] * w.x + Iw[3] * w.y + Iw[4] * w.z, Iw[2] * w.x + Iw[4] * w.y + Iw[5] * w.z);
            const v3 Ial = mk3(Iw[0] * al.x + Iw[1] * al.y + Iw[2] * al.z, Iw[1] * al.x + Iw[3] * al.y + Iw[4] * al.z, Iw[2] * al.x + Iw[4] * al.y + Iw[5] * al.z);
            const v3 n = cross_add(Ial, w, Iwv);
            const double ex = (double)com.x - (double)o.x, ey = (double)com.y - (double)o.y, ez = (double)com.z - (double)o.z;
            const double fx = f.x, fy = f.y, fz = f.z, md = mk, ee = ex * ex + ey * ey + ez * ez;
            double* x = xb + 16 * k;
            x[0] = fx; x[1] = fy; x[2] = fz;
            x[3] = (double)n.x + (ey * fz - ez * fy); x[4] = (double)n.y + (ez * fx - ex * fz); x[5] = (double)n.z + (ex * fy - ey * fx);
            x[6] = md; x[7] = md * ex; x[8] = md * ey; x[9] = md * ez;
            x[10] = (double)Iw[0] + md * (ee - ex * ex); x[11] = (double)Iw[1] - md * ex * ey; x[12] = (double)Iw[2] - md * ex * ez;
            x[13] = (double)Iw[3] + md * (ee - ey * ey); x[14] = (double)Iw[4] - md * ey * ez; x[15] = (double)Iw[5] + md * (ee - ez * ez);
        } else if (hl == m.J) {
            double* x = xb + 16 * hl;
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = 0.0;                // link slot 15 does not exist: 0 x garbage must not reach the sums
        }
        sync();
        // S_c = M X_c for both characters: the operands are wave-wide (lane l: A[i = l & 15][k = l >> 4 + 4 t], B[k][q = l & 15])
        const int li_ = wl & 15, lk_ = wl >> 4;
        const uint32_t mrow = (li_ < m.J) ? s.mdl.subtree_mask[li_] : 0u;     // the model block is the same in both records
        const double* xa = reinterpret_cast<const double*>(&rec[0].Lt[0]);
        const double* xc = reinterpret_cast<const double*>(&rec[1].Lt[0]);
        d4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 4 * t + lk_;
            const double a = ((mrow >> k) & 1u) ? 1.0 : 0.0;
            s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xa[16 * k + li_], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xc[16 * k + li_], s1, 0, 0, 0);
        }
        sync();                                                     // every lane has read X before the sums overwrite it
        {
            double* sa = reinterpret_cast<double*>(&rec[0].Lt[0]);
            double* sc2 = reinterpret_cast<double*>(&rec[1].Lt[0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int j = lk_ + 4 * r; sa[16 * j + li_] = s0[r]; sc2[16 * j + li_] = s1[r]; }   // f64 C/D: row = (l >> 4) + 4 reg
        }
        sync();
        if (hl < m.J) {
            const int j = hl;
            const double* S = xb + 16 * j;
            const v3 pj = ld3(s.p[j]);
            const double rx = (double)pj.x - (double)o.x, ry = (double)pj.y - (double)o.y, rz = (double)pj.z - (double)o.z;
            const double Fx = S[0], Fy = S[1], Fz = S[2], mc = S[6], Hx = S[7], Hy = S[8], Hz = S[9];
            s.Fs[j][0] = (Real)Fx; s.Fs[j][1] = (Real)Fy; s.Fs[j][2] = (Real)Fz;
            s.Ns[j][0] = (Real)(S[3] - (ry * Fz - rz * Fy)); s.Ns[j][1] = (Real)(S[4] - (rz * Fx - rx * Fz)); s.Ns[j][2] = (Real)(S[5] - (rx * Fy - ry * Fx));
            const double dg = mc * (rx * rx + ry * ry + rz * rz) - 2.0 * (Hx * rx + Hy * ry + Hz * rz);
            Real* ic = s.Ic[j];
            ic[0] = (Real)mc; ic[1] = (Real)(Hx - mc * rx); ic[2] = (Real)(Hy - mc * ry); ic[3] = (Real)(Hz - mc * rz);
            ic[4] = (Real)(S[10] + dg + 2.0 * Hx * rx - mc * rx * rx);
            ic[5] = (Real)(S[11] + Hx * ry + rx * Hy - mc * rx * ry);
            ic[6] = (Real)(S[12] + Hx * rz + rx * Hz - mc * rx * rz);
            ic[7] = (Real)(S[13] + dg + 2.0 * Hy * ry - mc * ry * ry);
            ic[8] = (Real)(S[14] + Hy * rz + ry * Hz - mc * ry * rz);
            ic[9] = (Real)(S[15] + dg + 2.0 * Hz * rz - mc * rz * rz);
        }
        sync();
    }
#endif
    DM_DEV void dynamics(int iset, Real diag_scale) {
        const int D = m.D;
#ifndef DM_EMU
        if constexpr (sizeof(Real) == 4) {
            subtree_mfma(iset);
            for (int k = hl; k < D; k += HW) b.dyn_dofrec(k);
            sync();
        } else
#endif
        {
            b.dyn_links(iset);
            for (int k = hl; k < D; k += HW) b.dyn_dofrec(k);
            sync();
            b.dyn_subtree();
            sync();
        }
        if (hl + 3 < D) b.dyn_row(hl + 3, diag_scale);
        if (hl == HW - 1) {
            for (int k = 0; k < 3; ++k) s.dofrec[k][7] = s.Fs[0][k];       // root translation: C_k = total force
            s.sc[7] = s.Ic[0][0];                                           // H_kk, k < 3: total mass (Ic is recycled before the factorisation)
        }
        sync();
    }

    // ------------------------------------------------------------------ Cholesky + solve, 31 row lanes per character
    // x := H^-1 x for the LDS vector xvec.  Same factor layout in LDS as EnvSim::chol_solve (diagonal slot = 1/L_kk).
    DM_DEV void chol_solve(Real* xvec) {
        const int D = m.D;
        const int own = hl + 3;                          // the row this lane owns
        const bool valid = own < D;
        R2 h2[NP2];
        const int rr = valid ? own : 3;
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            R2 v = *reinterpret_cast<const R2*>(&s.Lt[L::lrow(rr) + 2 * p]);
            if (!valid) { v[0] = 0; v[1] = 0; }
            h2[p] = v;
        }
        // Two columns per block, software-pipelined: the block's two columns are published through LDS for the trailing rank-2
        // update, but the NEXT block's pivot pair is brought up to date from registers (four in-register broadcasts of the
        // entries L[k+2..k+3][k..k+1]), so the pivot chain (broadcast, rsq, Newton step) never waits on the LDS round trip; the
        // trailing update of block k runs behind the pivot computation of block k+2.
        Real* colbuf = &s.f[0][0];                       // 2 blocks x 2 columns x 40 words per character (aliases the dead Newton-Euler sums)
        const Real M = s.sc[7];                          // H_kk, k < 3: total mass (root Kd = 0)
        const Real dinv0 = dm_rsqrt(M);
        sync();                                          // every lane has read what it needs from the aliased region
        Real lik0, lik1;
        // columns k, k+1 (k even) of this lane from the up-to-date pair h2[k/2]; publishes them into buffer (k/2)&1
#define DM_DUO_COLS(k)                                                                                                  \
        {                                                                                                               \
            const int pk_ = (k) >> 1;                                                                                   \
            const Real inv0_ = ((k) < 3) ? dinv0 : dm_rsqrt(half_bcast_c<(((k) - 3) & 31)>(h2[pk_][0], half));                     \
            lik0 = h2[pk_][0] * inv0_; h2[pk_][0] = lik0;                                                               \
            const Real lk1k_ = ((k) + 1 < 3) ? (Real)0 : half_bcast_c<(((k) + 1 - 3) & 31)>(lik0, half);                           \
            h2[pk_][1] -= lik0 * lk1k_;                                                                                 \
            const Real inv1_ = ((k) + 1 < 3) ? dinv0 : dm_rsqrt(half_bcast_c<(((k) + 1 - 3) & 31)>(h2[pk_][1], half));             \
            lik1 = h2[pk_][1] * inv1_; h2[pk_][1] = lik1;                                                               \
            if ((k) + 2 < ND) {                                                                                         \
                Real* cb0_ = colbuf + (pk_ & 1) * 80; Real* cb1_ = cb0_ + 40;                                           \
                if (valid) { cb0_[own] = lik0; cb1_[own] = lik1; }                                                      \
                if ((k) < 3 && hl == HW - 1) { cb0_[0] = 0; cb0_[1] = 0; cb0_[2] = 0; cb1_[0] = 0; cb1_[1] = 0; cb1_[2] = 0; } \
            }                                                                                                           \
        }
#define DM_DUO_BLOCK(k)                                                                                                 \
        {                                                                                                               \
            const int pk = (k) >> 1; \
            const Real c0 = lik0, c1 = lik1; \
            if (k + 2 < ND) { \
                sync(); \
                const Real* cb0 = colbuf + (pk & 1) * 80; const Real* cb1 = cb0 + 40; \
                R2 t0[NP2], t1[NP2]; \
_Pragma("unroll") \
                for (int p = pk + 2; p < NP2; ++p) { t0[p] = *reinterpret_cast<const R2*>(&cb0[2 * p]); t1[p] = *reinterpret_cast<const R2*>(&cb1[2 * p]); } \
                const Real a = (k + 2 < 3) ? (Real)0 : half_bcast_c<(((k) + 2 - 3) & 31)>(c0, half), c = ((k) + 2 < 3) ? (Real)0 : half_bcast_c<(((k) + 2 - 3) & 31)>(c1, half); \
                const Real bq = half_bcast_c<(((k) + 3 - 3) & 31)>(c0, half), d = half_bcast_c<(((k) + 3 - 3) & 31)>(c1, half); \
                h2[pk + 1][0] = (h2[pk + 1][0] - c0 * a) - c1 * c; \
                h2[pk + 1][1] = (h2[pk + 1][1] - c0 * bq) - c1 * d; \
                DM_DUO_COLS(k + 2) \
                const R2 l20 = {c0, c0}, l21 = {c1, c1}; \
_Pragma("unroll") \
                for (int p = pk + 2; p < NP2; ++p) { \
                    h2[p] = h2[p] - l20 * t0[p]; h2[p] = h2[p] - l21 * t1[p]; \
                    DM_OPAQUE_V(h2[p]); \
                } \
            } \
        }
        DM_DUO_COLS(0)
        static_assert(ND == 34, "17 column blocks");
        DM_DUO_BLOCK(0) DM_DUO_BLOCK(2) DM_DUO_BLOCK(4) DM_DUO_BLOCK(6) DM_DUO_BLOCK(8) DM_DUO_BLOCK(10) DM_DUO_BLOCK(12) DM_DUO_BLOCK(14) DM_DUO_BLOCK(16)
        DM_DUO_BLOCK(18) DM_DUO_BLOCK(20) DM_DUO_BLOCK(22) DM_DUO_BLOCK(24) DM_DUO_BLOCK(26) DM_DUO_BLOCK(28) DM_DUO_BLOCK(30) DM_DUO_BLOCK(32)
#undef DM_DUO_BLOCK
#undef DM_DUO_COLS
        // the row goes to LDS as it is (L_kk on the diagonal); the lane then reads its own diagonal back, inverts it and puts 1/L_kk into
        // the slot -- one LDS round trip per factorisation instead of a lane compare + select per column (capture) and per pair (insert)
        Real dinv = 1;
        if (valid) {
            Real* row = &s.Lt[L::lrow(own)];
#pragma unroll
            for (int p = 0; p < NP2; ++p) if (2 * p <= own) *reinterpret_cast<R2*>(&row[2 * p]) = h2[p];
            dinv = dm_rcp(row[own]);
            row[own] = dinv;
        } else if (hl == HW - 1) {                       // rows 0..2 of the factor: 1/sqrt(M) on the diagonal, zeros left of it
            Lx(0, 0) = dinv0; Lx(1, 0) = 0; Lx(1, 1) = dinv0; Lx(2, 0) = 0; Lx(2, 1) = 0; Lx(2, 2) = dinv0;
        }
        // forward substitution; x0..x2 are uniform per character
        Real x = valid ? xvec[own] : (Real)0;
        Real xr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { xr[k] = xvec[k] * dinv0; x -= h2[k >> 1][k & 1] * xr[k]; }
        // rows 3..33 in order, statically expanded: the lane id of each broadcast is an immediate (DPP form of half_bcast_c).  The
        // entries of the own row on and right of the diagonal are zeroed once, so a step is mul, broadcast, FMA for every lane -- no
        // per-step lane compares -- and the lane's own scaling by 1/L_kk moves behind the loop (row k is final once step k is reached)
#pragma unroll
        for (int p = 0; p < NP2; ++p) { if (!(2 * p < own)) h2[p][0] = 0; if (!(2 * p + 1 < own)) h2[p][1] = 0; }
#define DM_DUO_FWD(k) { const Real xk = half_bcast_c<(k) - 3>(x * dinv, half); x -= h2[(k) >> 1][(k) & 1] * xk; }
        static_assert(ND == 34, "the expansion below covers rows 3..33");
        DM_DUO_FWD(3) DM_DUO_FWD(4) DM_DUO_FWD(5) DM_DUO_FWD(6) DM_DUO_FWD(7) DM_DUO_FWD(8) DM_DUO_FWD(9) DM_DUO_FWD(10)
        DM_DUO_FWD(11) DM_DUO_FWD(12) DM_DUO_FWD(13) DM_DUO_FWD(14) DM_DUO_FWD(15) DM_DUO_FWD(16) DM_DUO_FWD(17) DM_DUO_FWD(18)
        DM_DUO_FWD(19) DM_DUO_FWD(20) DM_DUO_FWD(21) DM_DUO_FWD(22) DM_DUO_FWD(23) DM_DUO_FWD(24) DM_DUO_FWD(25) DM_DUO_FWD(26)
        DM_DUO_FWD(27) DM_DUO_FWD(28) DM_DUO_FWD(29) DM_DUO_FWD(30) DM_DUO_FWD(31) DM_DUO_FWD(32) DM_DUO_FWD(33)
#undef DM_DUO_FWD
        x *= dinv;
        sync();
        back_substitute(x, xr, dinv, dinv0);
        if (valid) xvec[own] = x;
        if (hl == HW - 1) for (int k = 0; k < 3; ++k) xvec[k] = xr[k];
        sync();
    }
    // (x, xr) := L^-T (x, xr): rows 3..33 per lane with column reads from LDS, rows 0..2 by three half-wave sums
    DM_DEV void back_substitute(Real& x, Real (&xr)[3], Real dinv, Real dinv0) {
        const int own = hl + 3; const bool valid = own < m.D;
        Real c[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) c[k] = (valid && k > own) ? s.Lt[L::lrow(k) + own] : (Real)0;
#define DM_DUO_BWD(k) { const Real xk = half_bcast_c<(k) - 3>(x * dinv, half); x -= c[k] * xk; }      /* c[k] = 0 for k <= own */
        static_assert(ND == 34, "the expansion below covers rows 33..3");
        DM_DUO_BWD(33) DM_DUO_BWD(32) DM_DUO_BWD(31) DM_DUO_BWD(30) DM_DUO_BWD(29) DM_DUO_BWD(28) DM_DUO_BWD(27) DM_DUO_BWD(26)
        DM_DUO_BWD(25) DM_DUO_BWD(24) DM_DUO_BWD(23) DM_DUO_BWD(22) DM_DUO_BWD(21) DM_DUO_BWD(20) DM_DUO_BWD(19) DM_DUO_BWD(18)
        DM_DUO_BWD(17) DM_DUO_BWD(16) DM_DUO_BWD(15) DM_DUO_BWD(14) DM_DUO_BWD(13) DM_DUO_BWD(12) DM_DUO_BWD(11) DM_DUO_BWD(10)
        DM_DUO_BWD(9) DM_DUO_BWD(8) DM_DUO_BWD(7) DM_DUO_BWD(6) DM_DUO_BWD(5) DM_DUO_BWD(4) DM_DUO_BWD(3)
#undef DM_DUO_BWD
        x *= dinv;
        Real col[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = valid ? Lx(own, k) : (Real)0;
#pragma unroll
        for (int k = 0; k < 3; ++k) xr[k] = (xr[k] - half_sum(col[k] * x)) * dinv0;
    }

    DM_DEV Real clamp_vel(Real v, int dof) const { return b.clamp_vel(v, dof); }

    // ------------------------------------------------------------------ DM-physics v2, two characters per wavefront (round 4)
    // EnvSim::ground_manifolds with lane = link of the own half-wave: the persistent link-vs-ground manifold of each link (EnvState::manif) is
    // refreshed, given its one new support point, written back, and its points become the ground contact slots in (link, slot) order.  Ballots are
    // taken per half; the (rare) cap at max_contacts runs for both characters when either needs it (a no-op for the one that does not).
    DM_DEV int ground_manifolds(Real* manif) {
        const int J = m.J;
        int cnt = 0; Real lp[4][3], bxz[4][2], dist[4];
        Real* mfp = manif + (hl < J ? hl : 0) * MF_STRIDE;
        const bool has_body = hl < J && s.mdl.thresh[hl < J ? hl : 0] > (Real)0;
        // the candidates of a link are contiguous (build_host_model adds them link by link): every candidate lane reports its index to its link's
        // [first, count) cell in LDS, so the link lane scans its own <= 8 distances instead of all NC candidate records in global memory
        int* crange = &s.csel[0];                      // 2 x 16 ints (the cap path below reuses csel afterwards)
        static_assert(C::NCAP >= 64 && C::NJ <= 16, "csel holds the per-link candidate ranges");
        if (hl < 16) { crange[hl] = 0x7fffffff; crange[16 + hl] = 0; }
        sync();
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            const int c = hl + HW * q;
            if (c < m.NC) {
                const int link = cand_link[q];
                v3 x = ld3(s.com[link]) + ldm3(b.Rbp(link)) * mk3(cand_loc[q][0], cand_loc[q][1], cand_loc[q][2]);
                s.cdistc[c] = x.y - cand_rad[q];
                dm_atomic_min(&crange[link], c); dm_atomic_add(&crange[16 + link], 1);
            }
        }
        sync();
        const int cfirst = crange[hl & 15], ccount = crange[16 + (hl & 15)];
        sync();
        if (has_body) {
            const Real thr = s.mdl.thresh[hl];
            const v3 com = ld3(s.com[hl]); const m3 Rb = ldm3(b.Rbp(hl));
            cnt = (int)mfp[0];
            for (int i = 0; i < 4; ++i) { for (int k = 0; k < 3; ++k) lp[i][k] = mfp[1 + i * MF_PT + k]; bxz[i][0] = mfp[1 + i * MF_PT + 3]; bxz[i][1] = mfp[1 + i * MF_PT + 4]; dist[i] = mfp[1 + i * MF_PT + 5]; }
            for (int i = cnt - 1; i >= 0; --i) {         // refreshContactPoints, last to first
                const v3 xa = com + Rb * mk3(lp[i][0], lp[i][1], lp[i][2]);
                dist[i] = xa.y;
                const Real dx = bxz[i][0] - xa.x, dz = bxz[i][1] - xa.z;
                if (!(dist[i] <= thr) || dx * dx + dz * dz > thr * thr) {
                    for (int k = i; k + 1 < cnt; ++k) { for (int a = 0; a < 3; ++a) lp[k][a] = lp[k + 1][a]; bxz[k][0] = bxz[k + 1][0]; bxz[k][1] = bxz[k + 1][1]; dist[k] = dist[k + 1]; }
                    --cnt;
                }
            }
            int best = -1; Real bd = 0;                 // the new point: the deepest candidate of this link, first on ties
            for (int c = cfirst; c < cfirst + ccount; ++c) { const Real d = s.cdistc[c]; if (best < 0 || d < bd) { best = c; bd = d; } }
            if (best >= 0 && bd < thr) {
                v3 x = com + Rb * mk3(m.cand_loc[best * 3], m.cand_loc[best * 3 + 1], m.cand_loc[best * 3 + 2]);
                x.y -= m.cand_rad[best];
                const v3 d0 = x - com;
                const v3 nl = mk3(Rb.m[0] * d0.x + Rb.m[3] * d0.y + Rb.m[6] * d0.z, Rb.m[1] * d0.x + Rb.m[4] * d0.y + Rb.m[7] * d0.z, Rb.m[2] * d0.x + Rb.m[5] * d0.y + Rb.m[8] * d0.z);
                int slot = -1; Real shortest = thr * thr;
                for (int i = 0; i < cnt; ++i) { const Real ex = lp[i][0] - nl.x, ey = lp[i][1] - nl.y, ez = lp[i][2] - nl.z, d2 = ex * ex + ey * ey + ez * ez; if (d2 < shortest) { shortest = d2; slot = i; } }
                if (slot < 0) {
                    if (cnt < 4) slot = cnt++;
                    else {                              // sortCachedPoints: never the deepest; of the others the one that leaves the largest quadrilateral
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
            for (int i = 0; i < cnt; ++i) if (dist[i] <= m.report_dist) dm_atomic_or(&s.flg[FLG_CONTACT], 1 << hl);
        }
        sync();
        bool keep[4];
        for (int i = 0; i < 4; ++i) keep[i] = i < cnt;
        int total = 0; uint32_t km[4];
        for (int i = 0; i < 4; ++i) { km[i] = (uint32_t)(wave_ballot(keep[i]) >> (half * 32)); total += dm_popc64(km[i]); }
        if (wave_ballot(total > m.max_contacts) != 0) {
            for (int i = 0; i < 4; ++i) if (hl < J) { s.csel[hl * 4 + i] = keep[i] ? 1 : 0; s.cdistc[hl * 4 + i] = dist[i]; }
            sync();
            for (int i = 0; i < 4; ++i) {
                int rank = 0;
                if (keep[i]) for (int k = 0; k < 4 * J; ++k) if (s.csel[k] && (s.cdistc[k] < dist[i] || (s.cdistc[k] == dist[i] && k < hl * 4 + i))) ++rank;
                keep[i] = keep[i] && rank < m.max_contacts;
            }
            sync();
            total = 0;
            for (int i = 0; i < 4; ++i) { km[i] = (uint32_t)(wave_ballot(keep[i]) >> (half * 32)); total += dm_popc64(km[i]); }
        }
        const uint32_t ltm = (hl == 0) ? 0u : (~0u >> (32 - hl));
        int base = 0;
        for (int i = 0; i < 4; ++i) base += dm_popc64(km[i] & ltm);
        if (hl < J) {
            const v3 com = ld3(s.com[hl]); const m3 Rb = ldm3(b.Rbp(hl));
            int k = 0;
            for (int i = 0; i < 4; ++i) if (keep[i]) { b.store_contact(base + k, com + Rb * mk3(lp[i][0], lp[i][1], lp[i][2]), mk3((Real)0, (Real)1, (Real)0), dist[i], hl, 255); ++k; }
        }
        return total;
    }

    // the end of a substep for both characters: xs holds Y lambda per dof; delta v = L^-T xs, clamp, integrate
    DM_DEV void substep_tail(Real h) {
        const int D = m.D;
        sync();
        {
            const int own = hl + 3; const bool valid = own < D;
            Real x = valid ? b.xs()[own] : (Real)0;
            Real xr[3] = { b.xs()[0], b.xs()[1], b.xs()[2] };
            const Real dinv = valid ? Lx(valid ? own : 3, valid ? own : 3) : (Real)1, dinv0 = Lx(0, 0);
            sync();
            back_substitute(x, xr, dinv, dinv0);
            if (valid) b.xs()[own] = x;
            if (hl == HW - 1) for (int k = 0; k < 3; ++k) b.xs()[k] = xr[k];
        }
        sync();
        for (int k = hl; k < D; k += HW) { const int vidx = DM_DI_VIDX(s.mdl.dof_info[k]); s.vel[vidx] = clamp_vel(s.dofrec[k][6] + b.xs()[k], k); }
        sync();
        b.integrate(h);
        sync();
#if DM_PRIO_BACK
        prio_low();
#endif
    }

    // ------------------------------------------------------------------ rigid-body substep, constraint part
    // returns false (having done nothing that matters -- under V2: the manifolds are updated and the ground slots stored, FLG_NCONT tells the
    // one-per-wave routine how many) when either character needs more than 32 constraint rows
    template <bool V2 = false>
    DM_DEV bool substep_post(Real h, Real* manif = nullptr) {
        const int D = m.D;
        for (int k = hl; k < D; k += HW) { const int vidx = DM_DI_VIDX(s.mdl.dof_info[k]); s.dofrec[k][6] = clamp_vel(s.vel[vidx] + h * s.rhs[k], k); }
        if (hl == 0) s.flg[FLG_CONTACT] = 0;
        sync();
        b.mark(7);
        int nact = 0;
        const uint32_t lt = (hl == 0) ? 0u : (~0u >> (32 - hl));
        // (one scalar load per substep, then a plain register: left to the allocator these two kernel arguments are re-loaded from the kernarg segment at every use --
        // s_load + s_waitcnt lgkmcnt(0) inside the pair loop and the friction refresh -- once the borrowed-lane path shares the kernel's scalar registers)
        int maxc = m.max_contacts; DM_OPAQUE_S(maxc);
        // OBJ classes: the free body's unconstrained velocity (btRigidBody::applyDamping, then the gravity impulse), every lane of the half (EnvSim::substep_post)
        v3 bpos = zero3(), bvs = zero3(), bws = zero3();
        if constexpr (C::OBJ) {
            const Real* ob = s.obj;
            bpos = ld3(ob + OB_PX);
            bvs = (Real)exp((double)h * m.ball_ln_lin) * ld3(ob + OB_VX) + h * mk3(m.gravity[0], m.gravity[1], m.gravity[2]);
            bws = (Real)exp((double)h * m.ball_ln_ang) * ld3(ob + OB_WX);
        }
        if (V2) { nact = ground_manifolds(manif); if (hl == 0) s.flg[FLG_NCONT] = nact; }
        else {
        // ---- collision: lane = candidate, two passes
        bool active[CP]; Real dist[CP]; v3 cxp[CP]; uint32_t amask[CP];
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            const int c = hl + HW * q;
            active[q] = false; dist[q] = 0; cxp[q] = zero3();
            if (c < m.NC) {
                int link = cand_link[q];
                v3 x = ld3(s.com[link]) + ldm3(b.Rbp(link)) * mk3(cand_loc[q][0], cand_loc[q][1], cand_loc[q][2]);
                x.y -= cand_rad[q];
                dist[q] = x.y; cxp[q] = x;
                active[q] = x.y < s.mdl.thresh[link];
                if (x.y <= m.report_dist) dm_atomic_or(&s.flg[FLG_CONTACT], 1 << link);
            }
            amask[q] = (uint32_t)(wave_ballot(active[q]) >> (half * 32));
            nact += dm_popc64(amask[q]);
        }
        if (wave_ballot(nact > maxc) != 0) {
            // manifold reduction (rare): keep the max_contacts deepest, ties to the lower index
#pragma unroll
            for (int q = 0; q < CP; ++q) { const int c = hl + HW * q; s.csel[c] = active[q] ? 1 : 0; s.cdistc[c] = dist[q]; }
            sync();
#pragma unroll
            for (int q = 0; q < CP; ++q) {
                const int c = hl + HW * q; int rank = 0;
                if (active[q]) for (int k = 0; k < m.NC; ++k) if (s.csel[k] && (s.cdistc[k] < dist[q] || (s.cdistc[k] == dist[q] && k < c))) ++rank;
                active[q] = active[q] && rank < maxc;
            }
            sync();
            nact = 0;
#pragma unroll
            for (int q = 0; q < CP; ++q) { amask[q] = (uint32_t)(wave_ballot(active[q]) >> (half * 32)); nact += dm_popc64(amask[q]); }
        }
        {   // ground contacts -> slots, in candidate-index order
            int base = 0;
#pragma unroll
            for (int q = 0; q < CP; ++q) {
                if (active[q]) b.store_contact(base + dm_popc64(amask[q] & lt), cxp[q], mk3((Real)0, (Real)1, (Real)0), dist[q], cand_link[q], 255);
                base += dm_popc64(amask[q]);
            }
        }
        }
        int nc = V2 ? half_bcast(nact, 0, half) : nact;          // (uniform per character by construction; under V2 the broadcast tells the compiler)
        // ---- self collision: lane = link pair (three passes of 32); active pairs take the slots the ground left, in pair order
        const int npair_passes = (m.NPAIR + HW - 1) / HW;
#if DM_PAIRPREF
        // (round 6, second pass) the operands of pass q + 1 are requested before pass q is evaluated, by every lane (an idle lane reads pair 0 / 0): under the lanes'
        // `code >= 0` branch the reads of a pass went out one dependent round trip at a time
        typename Base::PairIn pin[2];
        pin[0] = b.pair_load(pair_code[0] >= 0 ? (pair_code[0] & 0xff) : 0, pair_code[0] >= 0 ? (pair_code[0] >> 8) : 0);
#endif
#pragma unroll
        for (int q = 0; q < PP; ++q) {
#if DM_PAIRPREF
            if (q + 1 < PP) { const int cn = pair_code[q + 1 < PP ? q + 1 : 0]; pin[(q + 1) & 1] = b.pair_load(cn >= 0 ? (cn & 0xff) : 0, cn >= 0 ? (cn >> 8) : 0); }
#endif
            if (q < npair_passes) {
                const int code = pair_code[q];
                v3 x = zero3(), n = zero3(); Real dsc = 0; bool act = false;
#if DM_PAIRPREF
                act = b.pair_eval(pin[q & 1], x, n, dsc) && code >= 0;
#else
                if (code >= 0) act = b.self_pair(code & 0xff, code >> 8, x, n, dsc);
#endif
                const uint64_t mk64 = wave_ballot(act);
                if (mk64 != 0) {
                    const uint32_t mk = (uint32_t)(mk64 >> (half * 32));
                    const int slot = nc + dm_popc64(mk & lt);
                    if (act && slot < maxc) b.store_contact(slot, x, n, dsc, code & 0xff, code >> 8);
                    nc = dm_min(maxc, nc + (int)dm_popc64(mk));
                }
            }
        }
        if constexpr (C::OBJ) {
            // the free body takes the slots that are left: lane 0 of the half tests it against the ground, lane 1 + j against link j (capsule models, as link against
            // link); slot order = lane order.  Link ids in the slot: 254 = the body, 255 = the ground.  (EnvSim::substep_post, per half)
            const Real rb = m.ball_radius, thr_b = m.ball_thresh;
            v3 x = zero3(), n = mk3((Real)0, (Real)1, (Real)0); Real dsc = 0; bool act = false; int la = 254, lb = 255;
            if (hl == 0) { dsc = bpos.y - rb; act = dsc < thr_b; x = bpos; x.y -= rb; }
            else if (hl <= m.J) {
                const int j = hl - 1;
                const Real* cj = s.mdl.cap[j];
                const v3 uj = ldm3(b.Rbp(j)) * ld3(cj), p1 = ld3(s.com[j]) + uj, d1 = (Real)-2 * uj, r = p1 - bpos;
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
            const uint64_t mk64 = wave_ballot(act);
            if (mk64 != 0) {
                const uint32_t mk = (uint32_t)(mk64 >> (half * 32));
                const int slot = nc + dm_popc64(mk & lt);
                if (act && slot < maxc) b.store_contact(slot, x, n, dsc, la, lb);
                nc = dm_min(maxc, nc + (int)dm_popc64(mk));
            }
        }
        const int NL = m.NL;
        const int R = NL + 3 * nc;
        if (wave_ballot(R > HW) != 0) {      // a heavily contacted character: the caller decides between borrowed lanes and the one-per-wave routine (FLG_NROWS is published for it; V2: FLG_NCONT holds the ground slots)
            if (hl == 0) s.flg[FLG_NROWS] = R;
            sync();
            return false;
        }
        if (hl == 0) { s.flg[FLG_NROWS] = R; s.flg[FLG_NCONT] = nc; }
        sync();
        b.mark(8);
        // ---- constraint rows: lane = row (see EnvSim::substep_post)
        Real brow = 0;
        uint32_t ch_lo = 0, ch_hi = 0, ng_lo = 0, ng_hi = 0; v3 xd = zero3(), dd = zero3();
        int ball_sg = 0; v3 ball_cx = zero3();
        if (hl < R) {
            if (hl < NL) {
                const int lr = V2 ? (hl >> 1) : hl;         // v2: both rows of a limit (q - lo, then hi - q), v1: the nearer bound
                int j = s.mdl.lim_joint[lr]; int lj = s.mdl.link_info[j]; int off = DM_LI_POFF(lj);
                const int limdof = DM_LI_DOFF(lj);
                Real th = s.pose[off], pen_lo = th - s.mdl.lim_lo[lr], pen_hi = s.mdl.lim_hi[lr] - th;
                Real pen, sgn;
                if (V2 ? !(hl & 1) : (pen_lo <= pen_hi)) { sgn = 1; pen = pen_lo; } else { sgn = -1; pen = pen_hi; }
                brow = (pen > 0) ? -pen / h : -m.erp * pen / h;
                xd = sgn * ld3(&s.dofrec[limdof][0]);
                if (limdof < 32) ch_lo = 1u << limdof; else ch_hi = 1u << (limdof - 32);
            } else if constexpr (C::OBJ) b.contact_row(hl, NL, nc, h, brow, ch_lo, ch_hi, ng_lo, ng_hi, xd, dd, &ball_sg, &ball_cx);
            else b.contact_row(hl, NL, nc, h, brow, ch_lo, ch_hi, ng_lo, ng_hi, xd, dd);
        }
        // friction coefficient of this row's contact; the free body's Jacobian columns: d . v_b + ((x - p_b) x d) . w_b, signed by its side
        Real mu_row = m.friction;
        v3 jbl = zero3(), jba = zero3();
        if constexpr (C::OBJ) { if (ball_sg != 0) { mu_row = m.ball_friction; jbl = (Real)ball_sg * dd; jba = (Real)ball_sg * cross(ball_cx - bpos, dd); } }
        // y := L^-1 J^T, software-pipelined: the factor row and the dof record of step k+1 are requested (LDS broadcasts) before
        // the dependent accumulation chain of step k runs, so their latency hides behind it (two register buffers, static parity)
        R2 y2[NP2X]; Real cvec = 0;
#if DM_DUO_YFULL
        // (round 6, second pass) No per-dof `k < D` tests: every launch of this kernel has D == ND (dm_host.cpp checks it; the 31 row lanes per character assume
        // it).  As wave-uniform branches around the requests they made the compiler wait for everything in flight (lgkmcnt(0)) at every dof -- the requests of
        // dof k + 1 included, i.e. the pipelining was undone -- and cost two v_readlane of a spilled mask pair per dof.  The look-ahead set holds the dof record
        // and the first YP pairs of row k + 1; the pairs beyond are requested at the top of their own step and consumed last, behind the chain over the first
        // YP (both sets whole: 68 registers, which the kernel does not have -- kernel-long values went to scratch with reloads inside the update loop).
        constexpr int YP = DM_DUO_YPMAX;
        R2 lr[2][YP], ltl[NP2 > YP ? NP2 - YP : 1]; R4 rr[2][2];
#if DM_PRIO_Y
        dm_setprio<DM_PRIO_Y>();
#endif
        auto yhead = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            rr[k & 1][0] = *reinterpret_cast<const R4*>(&s.dofrec[k][0]); rr[k & 1][1] = *reinterpret_cast<const R4*>(&s.dofrec[k][4]);
            const R2* lrow_ = reinterpret_cast<const R2*>(&s.Lt[L::lrow(k)]);
            static_for<0, ((k >> 1) + 1 < YP ? (k >> 1) + 1 : YP)>([&](auto pc) { constexpr int p = decltype(pc)::value; lr[k & 1][p] = lrow_[p]; });
        };
        yhead(std::integral_constant<int, 0>{});
        static_for<0, ND>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const R2* lrow_ = reinterpret_cast<const R2*>(&s.Lt[L::lrow(k)]);
            static_for<YP, (k >> 1) + 1>([&](auto pc) { constexpr int p = decltype(pc)::value; ltl[p - YP] = lrow_[p]; });      // the row's tail, if any
            if constexpr (k + 1 < ND) yhead(std::integral_constant<int, (k + 1 < ND ? k + 1 : 0)>{});
            if (DM_DUO_YFULL >= 2) DM_SCHED_FENCE();
            const R4 r0 = rr[k & 1][0], r1 = rr[k & 1][1];
            Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
            DM_OPAQUE_V(val);        // (evaluated where it stands, by every lane: see EnvSim::substep_post's tree loop)
            const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
            const Real raw = on ? (ng ? -val : val) : (Real)0;
            cvec += raw * r1[2];
            R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;      // two independent accumulation chains
            static_for<0, (k >> 1)>([&](auto pc) { constexpr int p = decltype(pc)::value;
                const R2 e = (p < YP) ? lr[k & 1][p < YP ? p : 0] : ltl[p >= YP ? p - YP : 0];
                if constexpr (p & 1) acc3 += e * y2[p]; else acc2 += e * y2[p]; });
            acc2 += acc3;
            Real acc = raw - (acc2[0] + acc2[1]);
            constexpr int pd = k >> 1;
            const R2 ed = (pd < YP) ? lr[k & 1][pd < YP ? pd : 0] : ltl[pd >= YP ? pd - YP : 0];
            if constexpr (k & 1) acc -= ed[0] * y2[k >> 1][0];
            Real yk = acc * ed[k & 1];
            DM_OPAQUE_V(yk);
            y2[k >> 1][k & 1] = yk;
            DM_SCHED_FENCE();       // (one dof per scheduling region)
        });
#else
        R2 lr[2][NP2]; R4 rr[2][2];
#if DM_PRIO_Y
        dm_setprio<DM_PRIO_Y>();
#endif
#define DM_DUO_YLOAD(k)                                                                                       \
        {                                                                                                     \
            rr[(k) & 1][0] = *reinterpret_cast<const R4*>(&s.dofrec[(k)][0]);                                  \
            rr[(k) & 1][1] = *reinterpret_cast<const R4*>(&s.dofrec[(k)][4]);                                  \
            const R2* lrow_ = reinterpret_cast<const R2*>(&s.Lt[L::lrow(k)]);                                  \
            _Pragma("unroll") for (int p = 0; p <= ((k) >> 1); ++p) lr[(k) & 1][p] = lrow_[p];                 \
        }
        DM_DUO_YLOAD(0)
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            Real yk = 0;
            if (k + 1 < ND) { if (k + 1 < D) DM_DUO_YLOAD(k + 1) }
            if (k < D) {
                const R4 r0 = rr[k & 1][0], r1 = rr[k & 1][1];
                const Real val = r0[0] * xd.x + r0[1] * xd.y + r0[2] * xd.z + r0[3] * dd.x + r1[0] * dd.y + r1[1] * dd.z;
                const bool on = (((k < 32) ? ch_lo : ch_hi) >> (k & 31)) & 1u, ng = (((k < 32) ? ng_lo : ng_hi) >> (k & 31)) & 1u;
                const Real raw = on ? (ng ? -val : val) : (Real)0;
                cvec += raw * r1[2];
                R2 acc2 = {(Real)0, (Real)0}, acc3 = acc2;      // two independent accumulation chains
#pragma unroll
                for (int p = 0; p < (k >> 1); ++p) { if (p & 1) acc3 += lr[k & 1][p] * y2[p]; else acc2 += lr[k & 1][p] * y2[p]; }
                acc2 += acc3;
                Real acc = raw - (acc2[0] + acc2[1]);
                if (k & 1) acc -= lr[k & 1][k >> 1][0] * y2[k >> 1][0];
                yk = acc * lr[k & 1][k >> 1][k & 1];
            }
            y2[k >> 1][k & 1] = yk;
        }
#undef DM_DUO_YLOAD
#endif
#if DM_PRIO_Y
        prio_low();
#endif
        if constexpr (C::OBJ) {
            // the free body's block of the mass matrix is diagonal: its rows of Y = M^-1/2 J^T are a scaling; J v* gains its share
            cvec += dot(jbl, bvs) + dot(jba, bws);
            const Real sm = dm_sqrt(m.ball_inv_mass), si = dm_sqrt(m.ball_inv_inertia);
            y2[NP2X - 3][0] = sm * jbl.x; y2[NP2X - 3][1] = sm * jbl.y; y2[NP2X - 2][0] = sm * jbl.z;
            y2[NP2X - 2][1] = si * jba.x; y2[NP2X - 1][0] = si * jba.y; y2[NP2X - 1][1] = si * jba.z;
        }
        b.mark(9);
        const int RN = NL + nc;
        const bool is_fric = hl >= RN && hl < R;
        Real lam = 0;
        if (wave_ballot(hl < RN && (brow - cvec) > 0) != 0) {
            RowFile<Real, 32> arow;
            Real adiag;
            { R2 a2 = {(Real)0, (Real)0};
#pragma unroll
              for (int p = 0; p < NP2X; ++p) a2 += y2[p] * y2[p];
              adiag = a2[0] + a2[1]; }
            const Real inv_adiag = (hl < R) ? (Real)1 / adiag : (Real)0;
            {
                Real g[32];
#pragma unroll
                for (int p = 0; p < NP2X; ++p) DM_OPAQUE_V(y2[p]);
                duo_gram32<NP2X>(y2, g);
#pragma unroll
                for (int r = 0; r < 32; ++r) arow.set(r, g[r] * inv_adiag);
            }
            b.mark(10);
            // Projected Gauss-Seidel on t = lambda + q (the pre-clamp target of a row): visiting row r leaves t_r unchanged
            // (A_rr = 1 after the row scaling) and moves every other t_c by -A_cr delta, so with a zeroed diagonal the column
            // update is one uniform FMA and the row's own `lambda + q` add disappears.
#pragma unroll
            for (int r = 0; r < 32; ++r) if (hl == r) arow.set(r, (Real)0);
            Real t = (brow - cvec) * inv_adiag;
            Real fric = m.friction; DM_OPAQUE_S(fric);
            const int nrm_lane = is_fric ? NL + ((hl - RN) >> 1) : 0;
            Real lo = 0, hi = is_fric ? (Real)0 : ((hl < NL) ? m.lim_max_impulse : (Real)1e30);      // (limit rows: maxAppliedImpulse)
            // sweep bounds of the pair: rows up to the larger R (rounded up to 4); a lane beyond its own R has t = 0, lambda = 0
            // and changes nothing.  fmask: the rows at which a character's friction bounds are refreshed from its normal impulses.
            const int Ra = lane_bcast(R, 0), Rb = lane_bcast(R, 32);
            int Rv = Ra > Rb ? Ra : Rb, lv = hl;
            uint32_t fmask = (1u << lane_bcast(RN, 0)) | (1u << lane_bcast(RN, 32));
#define DM_DUO_PGS_ROW(r)                                                                                              \
            {                                                                                                          \
                if (__builtin_expect((fmask >> (r)) & 1u, 0)) { const Real ln = wave_shfl(lam, half * 32 + nrm_lane); if (is_fric && (r) == RN) { if constexpr (C::OBJ) hi = mu_row * ln; else hi = fric * ln; lo = -hi; } } \
                const Real nl = dm_med3(lo, t, hi);                                                                    \
                const Real delta = half_bcast_c<(r)>(nl - lam, half);                                                  \
                t -= arow.get(r) * delta;                                                                              \
                lam = half_sel_c<(r)>(lam, nl, lv, one);                                                               \
            }
#define DM_DUO_PGS_BLK(b4) if ((b4) * 4 < Rv) { DM_DUO_PGS_ROW((b4) * 4) DM_DUO_PGS_ROW((b4) * 4 + 1) DM_DUO_PGS_ROW((b4) * 4 + 2) DM_DUO_PGS_ROW((b4) * 4 + 3) }
            // Wave priority by load while the sweep runs (round 4, profiles/r04_ab_setprio.json: +3.0 % on the headline, outputs bit-identical): the sweep
            // is one dependent chain per row that issues little; raised above the SIMD's other wave it stops waiting behind that wave's throughput
            // phases, and a pair with more rows than the median (16) -- the waves a one-round launch waits for -- outranks a lighter one.
            // The priority is per wave and lasts until it is set back after the sweep.  (Keeping it through all phases, thresholds 10..24,
            // or 3 for every wave measured the same or less; a never-raised control build measured +-0.)
#if DM_PRIO
            if (Rv > DM_PRIO_HI) dm_setprio<3>(); else if (Rv > DM_PRIO_LO) dm_setprio<DM_PRIO_MID>(); else dm_setprio<DM_PRIO_BASE>();
#endif
            for (int it = 0; it < m.solver_iters; ++it) {
                uint32_t one = 1u;
                DM_OPAQUE_S(Rv); DM_OPAQUE_S(fmask); DM_OPAQUE_V(lv); DM_OPAQUE_S(one);
                DM_DUO_PGS_BLK(0) DM_DUO_PGS_BLK(1) DM_DUO_PGS_BLK(2) DM_DUO_PGS_BLK(3)
                DM_DUO_PGS_BLK(4) DM_DUO_PGS_BLK(5) DM_DUO_PGS_BLK(6) DM_DUO_PGS_BLK(7)
            }
#undef DM_DUO_PGS_BLK
#undef DM_DUO_PGS_ROW
#if DM_PRIO
            dm_setprio<DM_PRIO_BACK>();
#endif
            if (hl >= R) lam = 0;
        } else { b.mark(10);
#if DM_PRIO_BACK
            dm_setprio<DM_PRIO_BACK>();
#endif
        }
        b.mark(11);
        if constexpr (C::OBJ) {
            // delta v of the free body = M^-1/2 (Y_b lambda): six sums over the half; then semi-implicit Euler with the exponential map, by lane 0 of the half
            Real dv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) dv[k] = half_sum(y2[NP2X - 3 + (k >> 1)][k & 1] * lam);
            if (hl == 0) {
                const Real sm = dm_sqrt(m.ball_inv_mass), si = dm_sqrt(m.ball_inv_inertia);
                const v3 bv = bvs + sm * mk3(dv[0], dv[1], dv[2]), bw = bws + si * mk3(dv[3], dv[4], dv[5]);
                Real* ob = s.obj;
                st3(ob + OB_VX, bv); st3(ob + OB_WX, bw);
                st3(ob + OB_PX, bpos + h * bv);
                stq(ob + OB_QW, qnormalize(qmul(quat_exp(h * bw), ldq(ob + OB_QW))));
            }
        }
        // delta v = L^-T (Y lambda): transposing reduction inside each half; dof k < 32 lands in lane k, dofs 32, 33 in lanes 0, 1
        {
            Real w[NP2];
            const bool bit = (hl & 1) != 0;
#pragma unroll
            for (int p = 0; p < NP2; ++p) {
                const Real a = y2[p][0] * lam, bb = y2[p][1] * lam;
                w[p] = (bit ? bb : a) + wave_shfl_xor_c<1>(bit ? a : bb);
            }
            b.template tr_stage<NP2, 2>(w);
            b.template tr_stage<(NP2 + 1) / 2, 4>(w);
            b.template tr_stage<(NP2 + 3) / 4, 8>(w);
            b.template tr_stage<(NP2 + 7) / 8, 16>(w);
            b.xs()[hl] = w[0];
            if (hl + 32 < D) b.xs()[hl + 32] = w[1];
        }
        return true;           // xs holds Y lambda: the caller runs substep_tail (shared with the borrowed-lane path)
    }

    // ------------------------------------------------------------------ one scene update for both characters
    template <bool PERT = false, bool V2 = false>
    DM_DEV void update(double dt, int e, Real* aovf_pair, double* pert = nullptr, bool kin_done = false, Real* manif_pair = nullptr) {
        if (hl == 0) { s.clk[CLK_TIMER] += dt; s.clk[CLK_CTRL] += dt; s.flg[FLG_NEED_ACTION] = 0; }
        if (PERT && pert) { if (hl == 0 && s.flg[FLG_PARKED] == 0) b.pert_tick(pert, e, dt); sync(); }      // enable_rand_perturbs (a parked character's row rests)
        b.template kin_update<PERT>(dt);
        const Real h = (Real)(dt / m.num_sim_substeps), rdt = (Real)dt;
        const int D = m.D;
        for (int ph = 0; ph <= m.num_sim_substeps; ++ph) {
            DM_OPAQUE_V(hl); DM_OPAQUE_V(b.l); DM_OPAQUE_V(b.li);
            b.mark(ph == 0 ? 0 : 4);
            if (ph == 1) {
                if (hl < m.J) { v3 da = b.gravity_a0() - b.spd_a0(); st3(s.aj[hl], ld3(s.aj[hl]) + da); }
                sync();
            } else if (!(kin_done && ph == 0)) {
#if DM_PRIO_KIN
                dm_setprio<DM_PRIO_KIN>();
#endif
                b.kinematics(s.pose, s.vel, ph == 0 ? b.spd_a0() : b.gravity_a0());      // (ph 0: EnvSim::kin_pre ran it)
#if DM_PRIO_KIN
                prio_low();
#endif
            }
            b.mark(ph == 0 ? 1 : 5);
            dynamics(ph == 0 ? 0 : 1, ph == 0 ? rdt : (Real)0);
            b.mark(ph == 0 ? 2 : 6);
            if (ph == 0) {
                b.spd_rhs_pre(rdt);
                for (int k = hl; k < D; k += HW) s.rhs[k] = b.xs()[k] - s.dofrec[k][7];
                sync();
            } else { for (int k = hl; k < D; k += HW) { Real r = s.tau[k] - s.dofrec[k][7]; if (PERT && pert) r += b.pert_gen_force(k); s.rhs[k] = r; } sync(); }
            DM_OPAQUE_V(hl); DM_OPAQUE_V(b.l);
#if DM_PRIO_CHOL
            dm_setprio<DM_PRIO_CHOL>();
#endif
            chol_solve(s.rhs);
#if DM_PRIO_CHOL
            prio_low();
#endif
            DM_OPAQUE_V(hl); DM_OPAQUE_V(b.l); DM_OPAQUE_V(b.li);
            if (ph == 0) {
                b.mark(3);
                for (int k = hl; k < D; k += HW) s.tau[k] = (k < 6) ? (Real)0 : b.xs()[k] - s.mdl.kd[DM_DI_JOINT(s.mdl.dof_info[k])] * rdt * s.rhs[k];
                sync();
                b.spd_clamp();
            } else {
                bool rows_done = substep_post<V2>(h, V2 ? manif_pair + (size_t)half * m.J * MF_STRIDE : nullptr);
                if (!rows_done) {
#if DM_PRIO_LATE
                    if (!PERT) late = 1;          // (plain instantiation only: in the AMP / v2 kernels the flag's scalar register cost 1 % of the open-loop rate of every AMP scene)
#endif
                    // more than 32 rows somewhere in the pair (the contact slots are stored, FLG_NROWS says how many).  One such character, at most DM_XD_ROWS rows, and at most 64
                    // rows together: the pair stays in this instruction stream on borrowed lanes (round 6)
                    const int R_ = s.flg[FLG_NROWS], nc_ = (R_ - m.NL) / 3;
                    const int Ra_ = lane_bcast(R_, 0), Rb_ = lane_bcast(R_, 32);
                    if constexpr (DM_DUO_XD && !C::OBJ) {      // (the borrowed-lane routine knows no free body: such a pair of biped + ball characters takes the 64-lane routine)
                        if (aovf_pair && Ra_ + Rb_ <= 2 * HW && Ra_ <= DM_XD_ROWS && Rb_ <= DM_XD_ROWS) {
                            b.mark(8);
                            DM_REGION_MARK(13);             // (s_nop 13 / s_nop 14 bracket the borrowed-lane region in the disassembly: tests/test_build_resources.py holds every scratch access of the update loop to it)
                            duo_rows_xd<Real, V2>(rec, wl, h, nc_, R_, Ra_, Rb_, D, m.NL, m.erp, m.friction, m.lim_max_impulse, m.solver_iters, aovf_pair);
                            load_cands();                   // (the candidate tables come back from L2 instead of living -- spilled in every wave's prologue -- across the region)
                            DM_REGION_MARK(14);
                            b.mark(11);                     // (profiling build: rows + Gram + sweep of such a substep count as "sub.PGS")
                            rows_done = true;
                        }
                    }
                }
                if (rows_done) { substep_tail(h); b.mark(12); continue; }
                // otherwise: one character at a time through the 64-lane routine
                if (FallbackCls::PRIO_FLOOR > 0) dm_setprio<FallbackCls::PRIO_FLOOR>();
                if (hl == 0) s.kin[7] += (Real)1;            // statistic (round 6): substeps this env spent on the fallback, in the pad word of its kin row (dm_get_debug "fallback")
                for (int x = 0; x < 2; ++x) {
                    int wlv = wl; DM_OPAQUE_V(wlv);          // keeps this rare path's address arithmetic out of the hot loop's live ranges
                    Single one(m, reinterpret_cast<WideRec&>(rec[x]), wlv);
                    one.li = (wlv < m.J) ? rec[x].mdl.link_info[wlv] : 0;
                    one.load_cands();
                    DebugTaps<Real> none = DebugTaps<Real>();
                    if (TAPS && b.prof) {                // profiling build: the 64-lane routine's marks count into this wave's phase totals (phases 7..12 of the fallback substep)
                        one.prof = b.prof; one.tprev = b.tprev;
#pragma unroll
                        for (int i = 0; i < 16; ++i) one.pacc[i] = 0;
                    }
                    // (V2: the two-per-wave pass above has already refreshed this character's manifolds and stored its ground slots -- the 64-lane routine takes them as they are)
                    one.template substep_post<V2, false>(h, none, e, (FallbackCls::RREG < kMaxRows && aovf_pair) ? aovf_pair + (size_t)x * (kMaxRows - FallbackCls::RREG) * kWave : nullptr,
                                                         V2 ? manif_pair + (size_t)x * m.J * MF_STRIDE : nullptr, V2 ? rec[x].flg[FLG_NCONT] : -1);
                    if (TAPS && b.prof) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) b.pacc[i] += one.pacc[i];
                        b.tprev = one.tprev;
                    }
                }
                if (FallbackCls::PRIO_FLOOR > 0) dm_setprio<0>();
            }
        }
        if (hl == 0) {
            double cur = s.clk[CLK_CTRL] + s.clk[CLK_INIT_OFF], pad = 0.001 * dt;
            int c1 = (int)floor((cur + pad) / m.query_period), c0 = (int)floor((cur + pad - dt) / m.query_period);
            s.flg[FLG_NEED_ACTION] = (c1 != c0) ? 1 : 0;
            bool over = b.episode_over_now();
            if (PERT) { if (m.enable_root_rot_fail && m.enable_fall_end && !over) over = b.root_rot_failed_now(); }
            s.flg[FLG_OVER] = (s.flg[FLG_OVER] & 2) | (over ? 1 : 0);
        }
        sync();
    }
};

// grid = N / 2 workgroups of one wavefront; character e = 2 * pair + (lane >> 5), pair = dm_wg_unit() (blockIdx, XCD-aware).  fp32: 2 waves / SIMD (20 KB LDS).
template <typename Real> struct DuoWaves { static constexpr int value = 1; };
template <> struct DuoWaves<float> { static constexpr int value = 2; };
template <typename Real, bool TAPS, bool AMP = false, bool V2 = false, typename CC = ClsBiped>
__global__ void __launch_bounds__(64) DM_WAVES_PER_EU((DuoWaves<Real>::value)) k_env_step_duo(ModelDev<Real> m, EnvState<Real> st, StepIO<Real> io, DebugTaps<Real> dbg) {
    constexpr bool HIST = TAPS || AMP;
    static_assert(!V2 || AMP, "the v2 instantiation carries the AMP code like k_env_step's");
    __shared__ Lds<Real, CC> lds[2];
    __shared__ ParkSnap<Real, CC> snap[2];     // 2 x 0.45 KB: the fp32 kernel stays inside 20 KB per wave (8 waves per CU)
    const int wl = threadIdx.x, half = wl >> 5;
    const int pair = dm_wg_unit();                   // (XCD-aware: dm_device.h)
    const int e = 2 * pair + half;
    DuoSim<Real, TAPS, CC> sim(m, lds, wl);
    if (TAPS && dbg.prof) sim.b.prof_begin(dbg.prof + (size_t)e * 16);
    sim.load(st, e);
    if (io.open_loop) sim.b.set_action_from_clip();
    else if (io.actions) sim.b.set_action(io.actions + (size_t)e * m.A);
    sim.b.mark(15);
    Real* aovf_pair = st.aovf ? st.aovf + (size_t)(2 * pair) * (kMaxRows - CC::RREG) * kWave : nullptr;
    Real* manif_pair = (V2 && st.manif) ? st.manif + (size_t)(2 * pair) * m.J * MF_STRIDE : nullptr;      // physics 2: the two characters' ground manifolds
    const bool goal = HIST && st.goal && m.scene_goal;
    if (HIST && st.goal) sim.b.clip = (int)st.goal[(size_t)e * GS_WIDTH + GS_CLIP];
    double* pert = (HIST && st.pert) ? st.pert + (size_t)e * PT_WIDTH : nullptr;
    if (goal) sim.b.goal_sync_flags(st, e);
    for (int u = 0; u < io.n_updates; ++u) {
        int eo = e; if (HIST) DM_OPAQUE_V(eo);          // see k_env_step: per-env addresses and draw keys of the rare paths are formed at their use
        double* po = (HIST && st.pert) ? st.pert + (size_t)eo * PT_WIDTH : nullptr;
        // link kinematics of the state + the validity test the driver runs after the previous update (EnvSim::kin_pre); a character found
        // invalid is over as of that update: both over -> done, one -> parked, exactly like an episode end
        sim.b.kin_pre(lds[half].flg[FLG_PARKED] == 0);
        if (io.end_early && ((lds[0].flg[FLG_OVER] | lds[1].flg[FLG_OVER]) & 2)) {
            const int o0 = lds[0].flg[FLG_OVER] | lds[0].flg[FLG_PARKED], o1 = lds[1].flg[FLG_OVER] | lds[1].flg[FLG_PARKED];
            if (o0 && o1) break;
            // bit 1 stays latched for the rest of the launch: park (three barriers) and the second kinematics pass only when a character is NEWLY invalid
            // (every later update used to pay them again)
            const bool newly = ((lds[0].flg[FLG_OVER] & 2) && !lds[0].flg[FLG_PARKED]) || ((lds[1].flg[FLG_OVER] & 2) && !lds[1].flg[FLG_PARKED]);
            if (newly) {
                const bool now = lds[half].flg[FLG_OVER] != 0 && lds[half].flg[FLG_PARKED] == 0;
                sim.b.park(snap[half], now);
                sim.b.kin_pre(false);                    // (the parked character's link state follows its new pose)
            }
        }
        if (HIST && st.hist) sim.b.latch_hist(st, eo, lds[half].flg[FLG_PARKED] == 0);
        if (goal) sim.b.goal_latch(st, eo, io.dt, lds[half].flg[FLG_PARKED] == 0);
        sim.template update<HIST, V2>(io.dt, eo, aovf_pair, po, true, V2 ? manif_pair : nullptr);
        if (goal) sim.b.goal_update(st, eo, io.dt, lds[half].flg[FLG_PARKED] == 0);      // the update that ends an episode includes its goal update
        if (io.end_early) {
            // DM_END_EPISODE_EARLY.  FLG_OVER is latched by each character's lane 0 at the end of update() (wave-uniform reads).
            // Both over: the wave is done.  One over (a few percent of the waves of a launch): what the outputs need of its record
            // is copied aside (LDS) and the character is parked; the copy comes back before the outputs are written.
            const bool v0 = lds[0].flg[FLG_OVER] != 0, v1 = lds[1].flg[FLG_OVER] != 0, p0 = lds[0].flg[FLG_PARKED] != 0, p1 = lds[1].flg[FLG_PARKED] != 0;
            if ((v0 || p0) && (v1 || p1)) break;
            if ((v0 && !p0) || (v1 && !p1)) {
                const bool now = lds[half].flg[FLG_OVER] != 0 && lds[half].flg[FLG_PARKED] == 0;
                sim.b.park(snap[half], now);
            }
        }
    }
    if (io.end_early) sim.b.unpark(snap[half]);
    if (io.emit) {
        DebugTaps<Real> tap = DebugTaps<Real>();
        sim.b.emit(io, tap, e, true);
        if (goal) sim.b.emit_goal(io, st, e, true);
        const bool ended = lds[half].sc[6] != (Real)0;
        if (HIST && io.amp_obs && st.hist) sim.b.emit_amp(io, st, e);
        if (io.auto_reset && ended) {                    // per character; no cross-half traffic inside
            uint64_t ep = (uint64_t)lds[half].flg[FLG_EPISODE];
            double mt = draw_time_limit<HIST>(m, e, ep, (HIST && st.goal) ? st.goal + (size_t)e * GS_WIDTH : nullptr);
            bool rec = false;
            if (HIST && st.goal) { rec = sim.b.try_recovery_reset(st, e, mt); if (!rec) reset_goal_env<Real, CC, TAPS, 32>(sim.b, m, lds[half], st, e, ep, nullptr, mt, true, pert); }
            else {
                double kt = m.duration * dm_rand01(m.seed, (uint64_t)(e + m.env_off), ep, 0);
                sim.b.reset_env(kt, mt);
                if (V2) sim.b.manif_clear(st, e);
                if (HIST && st.hist) sim.b.init_hist(st, e);
                if (HIST && pert && (wl & 31) == 0) sim.b.pert_reset(pert, e);
            }
            sim.b.emit(io, tap, e, false);
            if (goal) sim.b.emit_goal(io, st, e, false);
        }
        sim.b.mark(13);
    }
    sim.b.store(st, e);
    sim.b.mark(14);
    sim.b.prof_flush();
}

}  // namespace dmk
