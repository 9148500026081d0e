// On-device policy inference: the actor of learning/pg_agent.py:141-188 / ppo_agent.py --
//   a = unnormalize_a( W3 relu(W2 relu(W1 normalize_s(s) + b1) + b2) + b3  [+ exp(logstd) * N(0,1)] )
// with the net of learning/nets/fc_2layers_1024units.py (1024, 512, ReLU) and the normalizers of learning/normalizer.py:95-102 --
// so that the 30 Hz loop (observation -> action -> 20 scene updates) never leaves the GPU.
//
// Unlike the rigid-body path this IS matrix-core work (batch 4096 x 227 x 1024 ...): three dense layers on
// v_mfma_f32_16x16x32_bf16, bf16 operands, fp32 accumulation.  One wavefront per workgroup owns a (16*MT) x (16*NT) output
// tile; A fragments come straight from the row-major activations (16 B per lane), B fragments from weights the host packed
// in fragment order ([n-tile][k-step][lane][8], one coalesced 1 KB read per fragment, L2-resident: 1.5 MB in all); bias,
// ReLU, the observation normaliser (layer 1) and the Gaussian head + action un-normaliser (layer 3) are fused into the
// prologue / epilogue.  Operand mapping: lane l carries row / column l & 15 and the 8 k-values 8 * (l >> 4) .. + 7 of the
// 32-wide k-step for both A and B (any consistent assignment sums the same products); C/D: column l & 15, rows
// 4 * (l >> 4) + reg (cdna_hip_programming.md, "Fragment layout").
#pragma once
#include <stdint.h>

namespace dmp {

struct PolicyDev {
    int S, H1, H2, A;                 // true layer widths
    int K1, N3;                       // padded: K1 = S rounded up to 64, N3 = A rounded up to 32
    const uint16_t *w1p, *w2p, *w3p;  // packed bf16 fragments
    const float *b1, *b2, *b3;        // biases (b3 padded to N3 with zeros)
    const float *s_mean, *s_inv_std;  // observation normaliser (learning/normalizer.py:95-98), S entries
    float s_clip;
    const float *a_mean, *a_std;      // action un-normaliser (:100-102), A entries
    const float *logstd;              // A entries (TFDistributionGaussianDiag, StdType.Default: a bias vector)
    const uint16_t* wfs;              // k_policy_fused: the fragments of layers 1 and 2 in the order each of its four waves consumes them (null: widths it is not compiled for)
};

struct PolicyIO {
    const float* states;   // M x S fp32 (RecordState of every env)
    uint16_t* s16;         // M x K1 bf16: normalised, clipped, zero-padded observations (written by k_policy_prep)
    uint16_t* h1;          // M x H1 bf16
    uint16_t* h2;          // M x H2 bf16
    float* actions;        // M x A fp32
    float* logp;           // M or null
    int M;
    int sample;            // 0: mode of the Gaussian (test / deterministic), 1: mean + std * noise
    uint32_t seed_lo, seed_hi; uint32_t step; int env_off;   // Philox4x32-10 key = (seed_lo + global env id, seed_hi), counter = (step * A + j, 0, 0, 0)
    // dm_policy_forward_ex: the last G of the S input columns come from their own [M x G] block (RecordGoal; the net's input is the concatenation
    // [norm_s, norm_g], learning/pg_agent.py:170-187), states is then M x (S - G); exp_rate < 1: a row takes the sampled action with that
    // probability and the mode otherwise (pg_agent.py:214-216 _decide_action: flip_coin(exp_params_curr.rate)), exp_flags[row] says which
    const float* goals; int G;
    float exp_rate; int32_t* exp_flags;
    unsigned long long* prof;   // measurement only (DM_POLICY_PROBE=2): per workgroup 8 timestamps (s_memtime of wave 0 at the phase boundaries)
    int probe;             // measurement only (DM_POLICY_PROBE): 1 = k_policy_fused re-reads block 0 of its weight stream forever (the stream served by the vector L1: what the L2 path costs)
};

static inline uint16_t f32_to_bf16_host(float f) { uint32_t u; memcpy(&u, &f, 4); if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0; u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

#ifdef DM_EMU
#define DMP_SCHED_FENCE() ((void)0)
#define DMP_DEV inline
struct bf16x8 { uint16_t v[8]; };
struct f32x4 { float v[4]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };
static inline float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
// lane-exchange emulation of the matrix-core instruction; every lane calls it from uniform control flow
static inline f32x4 mfma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    static uint16_t xa[4][64][8], xb[4][64][8];          // up to 4 wavefronts per workgroup (k_policy_gemm)
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = 0; i < 8; ++i) { xa[w][l][i] = a.v[i]; xb[w][l][i] = b.v[i]; }
    __syncthreads();
    const int col = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r; float acc = c[r];
        for (int g2 = 0; g2 < 4; ++g2) for (int i = 0; i < 8; ++i) acc += bf16_to_f32(xa[w][row + 16 * g2][i]) * bf16_to_f32(xb[w][col + 16 * g2][i]);
        c[r] = acc;
    }
    __syncthreads();
    return c;
}
static inline float lane_xor_f(float v, int mask) { static float x[256]; x[threadIdx.x] = v; __syncthreads(); const float r = x[threadIdx.x ^ mask]; __syncthreads(); return r; }   // (workgroups of up to four wavefronts)
#else
#define DMP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DMP_DEV __device__ __forceinline__
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
DMP_DEV f32x4 mfma16(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
DMP_DEV float lane_xor_f(float v, int mask) { return __shfl_xor(v, mask, 64); }
#endif

DMP_DEV uint16_t f32_to_bf16(float f) {          // round to nearest even (inputs are finite)
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
DMP_DEV void set8(bf16x8& f, int i, uint16_t h) {
#ifdef DM_EMU
    f.v[i] = h;
#else
    f[i] = (short)h;
#endif
}

// two floats -> two bf16 in one dword, round to nearest even: v_cvt_pk_bf16_f32 on gfx950 (one instruction instead of ten)
DMP_DEV uint32_t pack2_bf16(float a, float b) {
#ifdef DM_EMU
    return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
#else
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
#endif
}
// four bf16 as ONE 8-byte store (dst is 8-byte aligned)
struct alignas(8) bf16x4_rec { uint32_t lo, hi; };
DMP_DEV void store4(uint16_t* dst, float a, float b, float c, float d) {
    bf16x4_rec v; v.lo = pack2_bf16(a, b); v.hi = pack2_bf16(c, d);
    *reinterpret_cast<bf16x4_rec*>(dst) = v;
}

// Philox4x32-10 (Salmon et al. 2011), the generator of deepmimic_amd/streams.py
DMP_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// the exploration coin of a row: one uniform per (env, step), counter word 1 = 1 keeps it apart from the action noise of the same step
DMP_DEV float philox_coin(uint32_t env, uint32_t step, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t r[4]; philox4x32_10(step, 1, 0, 0, seed_lo + env, seed_hi, r);
    return ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
DMP_DEV float philox_normal(uint32_t env, uint32_t ctr, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t r[4]; philox4x32_10(ctr, 0, 0, 0, seed_lo + env, seed_hi, r);
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// observation normaliser (learning/normalizer.py:95-98) fused with the bf16 conversion and the zero padding of K to a multiple of 32:
// one wavefront per row, coalesced reads and writes
__global__ void __launch_bounds__(64) k_policy_prep(PolicyDev p, PolicyIO io) {
    const int row = blockIdx.x, l = threadIdx.x;
    const int SS = p.S - io.G;                    // columns that come from the state block
    const float* srow = io.states + (size_t)row * SS;
    const float* grow = io.G ? io.goals + (size_t)row * io.G : nullptr;
    uint16_t* out = io.s16 + (size_t)row * p.K1;
    for (int k = l; k < p.K1; k += 64) {
        float x = 0.0f;
        if (k < p.S) { x = ((k < SS ? srow[k] : grow[k - SS]) - p.s_mean[k]) * p.s_inv_std[k]; x = fminf(fmaxf(x, -p.s_clip), p.s_clip); }
        out[k] = f32_to_bf16(x);
    }
}

// MODE 0: layer 1 (bf16 normalised observations; ReLU; bf16 out)             K = K1, N = H1
// MODE 1: layer 2 (bf16 in; ReLU; bf16 out)                                  K = H1, N = H2
// MODE 2: layer 3 (bf16 in; Gaussian head + un-normalise; fp32 actions)      K = H2, N = N3
// The k loop is software-pipelined by hand: the A / B fragments of step ks + 1 are requested before the MFMAs of step ks issue
// (two register sets), so one wave keeps its matrix core busy while the next 16-byte-per-lane reads are in flight.
template <int MODE, int MT, int NT>
__global__ void __launch_bounds__(64) k_policy_layer(PolicyDev p, PolicyIO io) {
    const int l = threadIdx.x, c = l & 15, g = l >> 4;
    const int K = (MODE == 0) ? p.K1 : (MODE == 1 ? p.H1 : p.H2);
    const int N = (MODE == 0) ? p.H1 : (MODE == 1 ? p.H2 : p.N3);
    const int n_col_tiles = N / (16 * NT);
    // consecutive workgroups walk down the rows of one column block: they read the same weight fragments back to back
    const int row_blocks = (io.M + 16 * MT - 1) / (16 * MT);
    // MODE 2 (Gaussian head): one workgroup owns ALL N3 columns of its rows -- the log-probability is a sum over every action
    // component, so the column blocks are walked inside the workgroup (A = 36 / 58 need two blocks of 32) and logp is written once.
    const int cb_first = (MODE == 2) ? 0 : (int)blockIdx.x / row_blocks, rb = (MODE == 2) ? (int)blockIdx.x : (int)blockIdx.x % row_blocks;
    if (cb_first >= n_col_tiles || rb >= row_blocks) return;
    const int cb_last = (MODE == 2) ? n_col_tiles : cb_first + 1;
    const int row0 = rb * 16 * MT, KS = K / 32;
    const uint16_t* wp = (MODE == 0) ? p.w1p : (MODE == 1 ? p.w2p : p.w3p);
    const uint16_t* ain = (MODE == 0) ? io.s16 : (MODE == 1 ? io.h1 : io.h2);
    float lps[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) lps[i][r] = 0.0f;

    for (int cb = cb_first; cb < cb_last; ++cb) {
    const int nt0 = cb * NT;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    const uint16_t* arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { const int row = row0 + 16 * i + c; arow[i] = ain + (size_t)(row < io.M ? row : 0) * K + 8 * g; }
    const uint16_t* bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bcol[j] = wp + ((size_t)(nt0 + j) * KS * 64 + l) * 8;

    bf16x8 a0[MT], b0[NT], a1[MT], b1[NT];
#define DMP_LOAD(A_, B_, ks_)                                                                                   \
    {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) A_[i] = *reinterpret_cast<const bf16x8*>(arow[i] + (size_t)(ks_) * 32);      \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) B_[j] = *reinterpret_cast<const bf16x8*>(bcol[j] + (size_t)(ks_) * 512);     \
    }
#define DMP_MMA(A_, B_)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(A_[i], B_[j], acc[i][j]);         \
    }
    if (MODE == 2) {
        // the head is tiny (N3 = 32): its time is all load latency, so the fragments of 8 k-steps are requested in one batch
        for (int kb = 0; kb < KS; kb += 8) {
            bf16x8 aa[8][MT], bb[8][NT];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (kb + u < KS) DMP_LOAD(aa[u], bb[u], kb + u)
#pragma unroll
            for (int u = 0; u < 8; ++u) if (kb + u < KS) DMP_MMA(aa[u], bb[u])
        }
    } else {
        DMP_LOAD(a0, b0, 0)
        for (int ks = 0; ks < KS; ks += 2) {          // KS is even: K1 is padded to 64, H1 and H2 are multiples of 64
            DMP_LOAD(a1, b1, ks + 1)
            DMP_MMA(a0, b0)
            if (ks + 2 < KS) DMP_LOAD(a0, b0, ks + 2)
            DMP_MMA(a1, b1)
        }
    }
#undef DMP_LOAD
#undef DMP_MMA

    if (MODE != 2) {
        const float* bias = (MODE == 0) ? p.b1 : p.b2;
        uint16_t* out = (MODE == 0) ? io.h1 : io.h2;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (nt0 + j) * 16 + c; const float bc = bias[col];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 16 * i + 4 * g + r;
                    if (row < io.M) out[(size_t)row * N + col] = f32_to_bf16(fmaxf(acc[i][j][r] + bc, 0.0f));
                }
        }
    } else {
        // Gaussian head: norm_a = mean + exp(logstd) z, a = norm_a * a_std + a_mean, logp = sum_j (-z^2/2 - logstd_j) - A/2 log 2 pi
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * i + 4 * g + r;
                float lp = 0.0f;
                bool explore = io.sample && row < io.M;
                if (explore && io.exp_rate < 1.0f) explore = philox_coin((uint32_t)(io.env_off + row), io.step, io.seed_lo, io.seed_hi) < io.exp_rate;
                if (io.exp_flags && c == 0 && nt0 == 0 && row < io.M) io.exp_flags[row] = explore ? 1 : 0;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = (nt0 + j) * 16 + c;
                    if (col < p.A) {
                        const float ls = p.logstd[col];
                        float z = 0.0f;
                        if (explore) z = philox_normal((uint32_t)(io.env_off + row), io.step * (uint32_t)p.A + (uint32_t)col, io.seed_lo, io.seed_hi);
                        const float na = acc[i][j][r] + p.b3[col] + expf(ls) * z;
                        if (row < io.M) io.actions[(size_t)row * p.A + col] = na * p.a_std[col] + p.a_mean[col];
                        lp += -0.5f * z * z - ls;
                    }
                }
                lps[i][r] += lp;
            }
    }
    }   // column blocks

    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * i + 4 * g + r;
                // the 16 lanes of a group hold 16 columns of one row (per column block); sum them, then the constant once
                float lp = lps[i][r];
                lp += lane_xor_f(lp, 1); lp += lane_xor_f(lp, 2); lp += lane_xor_f(lp, 4); lp += lane_xor_f(lp, 8);
                if (io.logp && c == 0 && row < io.M) io.logp[row] = lp - 0.5f * (float)p.A * 1.8378770664093453f;
            }
    }
}

// Layers 1 and 2 as an LDS-tiled GEMM: one workgroup of four wavefronts owns a 128 x 128 output tile (each wave 64 x 64 = 4 x 4 MFMA
// tiles, 16 v_mfma_f32_16x16x32_bf16 per k-step).  Per k-step the workgroup stages the 128 x 32 block of A and the 32 x 128 block of B
// (8 KB each, already in fragment order: [fragment][lane][8 bf16]) in LDS, double-buffered, every thread moving two 16-byte chunks of
// each; a wave then reads its 4 + 4 fragments back with conflict-free ds_read_b128 (lane-major 16-B records).  Against the one-wave
// kernel above that is 2 KB instead of 6 KB of L2 traffic per 8 MFMAs: the layer is no longer bound by the L2 -> CU path.
// MODE 0: layer 1 (A = normalised observations, K = K1, N = H1), MODE 1: layer 2 (A = h1, K = H1, N = H2).  N must be a multiple of 128.
// BM = 64 halves the tile height (each wave 32 x 64) for batches too small to fill the chip with 128 x 128 tiles.
// (Folding k_policy_prep into MODE 0 -- fp32 observations normalised on the way from the register stage to LDS -- was measured slower:
// 44.9 / 98.8 us against 41.2 / 83.4 us at 4096 / 16384 rows; eight scalar loads per chunk cost more than the 4 us kernel they replace.)
template <int MODE, int BM>
__global__ void __launch_bounds__(256) k_policy_gemm(PolicyDev p, PolicyIO io) {
    constexpr int BN = 128, MT = BM / 32, AF = BM / 16;      // MT: row fragments per wave, AF: row fragments per workgroup
    __shared__ bf16x8 sA[2][AF][64];
    __shared__ bf16x8 sB[2][8][64];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, g = l >> 4, wr = w >> 1, wc = w & 1;
    const int K = (MODE == 0) ? p.K1 : p.H1, N = (MODE == 0) ? p.H1 : p.H2, KS = K / 32;
    const int row_blocks = (io.M + BM - 1) / BM;
    // consecutive workgroups walk down the rows of one column block: they read the same weight fragments back to back
    const int cb = (int)blockIdx.x / row_blocks, rb = (int)blockIdx.x % row_blocks;
    if (cb >= N / BN) return;
    const int row0 = rb * BM, nt0 = cb * (BN / 16);
    const uint16_t* wp = (MODE == 0) ? p.w1p : p.w2p;
    const uint16_t* ain = (MODE == 0) ? io.s16 : io.h1;
    // staging: this thread moves fragments w and w + 4 (lane l) of both operands
    constexpr int AU = AF / 4;                               // A chunks per thread (2 for BM = 128, 1 for BM = 64)
    const uint16_t* ag[AU]; const uint16_t* bg[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = w + 4 * u, row = row0 + 16 * f + c;
        if (u < AU) ag[u < AU ? u : 0] = ain + (size_t)(row < io.M ? row : 0) * K + 8 * g;
        bg[u] = wp + ((size_t)(nt0 + f) * KS * 64 + l) * 8;
    }
    // two register stages: the chunks of k-step ks + 3 are requested while k-step ks is multiplied and reach LDS two iterations later,
    // so a request has two full iterations (MFMAs + barrier) to come back -- with one workgroup per CU nothing else hides the L2 latency
    bf16x8 ra[2][AU], rbv[2][2];
#define DMP_GLOAD(st_, ks_)                                                                                \
    { _Pragma("unroll") for (int u = 0; u < 2; ++u) { if (u < AU) ra[st_][u < AU ? u : 0] = *reinterpret_cast<const bf16x8*>(ag[u < AU ? u : 0] + (size_t)(ks_) * 32); \
                                                       rbv[st_][u] = *reinterpret_cast<const bf16x8*>(bg[u] + (size_t)(ks_) * 512); } }
#define DMP_LSTORE(buf_, st_)                                                                              \
    { _Pragma("unroll") for (int u = 0; u < 2; ++u) { if (u < AU) sA[buf_][w + 4 * u][l] = ra[st_][u < AU ? u : 0]; sB[buf_][w + 4 * u][l] = rbv[st_][u]; } }
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
    DMP_GLOAD(0, 0)
    DMP_LSTORE(0, 0)
    DMP_GLOAD(1, 1)                    // KS >= 2 (K is a multiple of 64)
    if (2 < KS) DMP_GLOAD(0, 2)
    __syncthreads();
#define DMP_ITER(ks, par)                                                                                  \
    {                                                                                                      \
        /* k-step ks + 1 (stage par ^ 1, requested two iterations ago) goes to the other buffer, its registers take k-step ks + 3 */ \
        if ((ks) + 1 < KS) DMP_LSTORE((par) ^ 1, (par) ^ 1)                                                \
        if ((ks) + 3 < KS) DMP_GLOAD((par) ^ 1, (ks) + 3)                                                  \
        DMP_MULT(par)                                                                                      \
        __syncthreads();                                                                                   \
    }
#define DMP_MULT(buf)                                                                                      \
    {                                                                                                      \
        bf16x8 a[MT], b[4];                                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { if (i < MT) a[i < MT ? i : 0] = sA[buf][MT * wr + i][l]; b[i] = sB[buf][4 * wc + i][l]; } \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);       \
    }
    for (int ks = 0; ks < KS; ks += 2) { DMP_ITER(ks, 0) DMP_ITER(ks + 1, 1) }
#undef DMP_ITER
#undef DMP_MULT
#undef DMP_GLOAD
#undef DMP_LSTORE
    const float* bias = (MODE == 0) ? p.b1 : p.b2;
    uint16_t* out = (MODE == 0) ? io.h1 : io.h2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = (nt0 + 4 * wc + j) * 16 + c; const float bc = bias[col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * MT * wr + 16 * i + 4 * g + r;
                if (row < io.M) out[(size_t)row * N + col] = f32_to_bf16(fmaxf(acc[i][j][r] + bc, 0.0f));
            }
    }
}


// ---- One launch for the whole actor (round 5) ---------------------------------------------------------------------------------------------
// A workgroup of four wavefronts takes a tile of 32 batch rows through prep + the three layers; an MLP is row-independent, so nothing is
// synchronised beyond the workgroup.  The products are computed TRANSPOSED, out^T = W^T x in^T: the weight fragment is the A operand (16
// output features x 32 k), the activations the B operand (32 k x 16 batch rows), so a lane of the result holds FOUR CONSECUTIVE FEATURES of
// one batch row (rows 4 (l >> 4) + r of the C tile) -- after bias + ReLU + bf16 they are 8 contiguous bytes of the next layer's B fragment:
// one ds_write_b64 per tile, no transposition.  (The host's fragment packing serves both roles: lane <-> (feature l & 15, k 8 (l >> 4) + i).)
//
// Layer 1 is produced in four chunks of 256 features (each wave 64); behind every chunk all waves add its contribution to their 128 layer-2
// outputs (64 accumulator VGPRs per wave, alive across the chunks).  LDS then holds the observations (KS1 x 2 KB), TWO chunk buffers
// (2 x 16 KB: one barrier per chunk) and, overlaid on the first two once they are dead, h2 (32 KB): 48 KB at K1 = 256, so the kernel fits
// beside resident waves of the step kernel of another env group (20 KB of LDS each).
//
// What bounds it: every workgroup streams ALL 1.5 MB of weights through its CU's 64 B / clk path from L2 (~ 10 us at 2.4 GHz; 128 workgroups
// at 4096 rows = 19 TB/s across the eight L2s).  The stream is laid out by the host in consumption order per wave -- blocks of 8 fragments
// (8 KB): layer-1 blocks = 2 k-steps x the wave's 4 feature tiles, layer-2 blocks = 1 k-step x its 8 feature tiles -- and walked with a
// three-deep register ring (block t + 2 is requested before block t is multiplied), the whole schedule unrolled at compile time so that the
// ring slots are plain registers.  KS1 = K1 / 32 (8: humanoid, 12: dog3d); H1 = 1024, H2 = 512 (learning/nets/fc_2layers_1024units.py);
// other widths take the per-layer kernels above.
template <int KS1, int N3T>
__global__ void __launch_bounds__(256, 2) k_policy_fused(PolicyDev p, PolicyIO io) {
    constexpr int R = 32, NQ = 4, NB1 = KS1 / 2, NB2 = 8, NBQ = NB1 + NB2, NBLK = NQ * NBQ;
    constexpr int LDS_S16 = KS1 * 2 * 64, LDS_H1C = 8 * 2 * 64, LDS_H2 = 16 * 2 * 64;            // in 16-byte records
    constexpr int LDS_TOTAL = (LDS_S16 + 2 * LDS_H1C > LDS_H2) ? LDS_S16 + 2 * LDS_H1C : LDS_H2;
    __shared__ bf16x8 lds[LDS_TOTAL];
    __shared__ float lp_part[4][16];
    __shared__ float sbias[1024 + 512];              // b1 | b2: a global read in an epilogue would have to drain the weight requests in flight behind it (vmcnt is in order)
    bf16x8* const s16 = lds;                         // [ks][bt][lane]
    bf16x8* const h1c = lds + LDS_S16;               // [2][ksl][bt][lane]
    bf16x8* const h2 = lds;                          // [ks][bt][lane]: overlays s16 and h1c[0] (dead from the barrier of the last chunk on)
    const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, g = l >> 4;
    const int row0 = (int)blockIdx.x * R;
#ifndef DM_EMU
#define DMF_STAMP(i_) { if (io.prof && t == 0) io.prof[(size_t)blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime(); }
#else
#define DMF_STAMP(i_) {}
#endif
    DMF_STAMP(0)
    const uint16_t* const ws = p.wfs + ((size_t)w * NBLK * 8 * 64 + l) * 8;      // this wave's stream; fragment f of block b at ws + (b * 8 + f) * 512
    bf16x8 ring[3][8];
#define DMF_LOAD(slot_, blk_)                                                                                              \
    { DMP_SCHED_FENCE(); _Pragma("unroll") for (int f = 0; f < 8; ++f) ring[slot_][f] = *reinterpret_cast<const bf16x8*>(ws + ((size_t)(io.probe == 1 ? 0 : ((blk_) < NBLK ? (blk_) : NBLK - 1)) * 8 + f) * 512); DMP_SCHED_FENCE(); }   \
    /* (fenced: left alone, the machine scheduler sinks every request to just in front of its MFMA to save registers, and the stream runs at one L2 latency per fragment) */
    DMF_LOAD(0, 0)                                   // the first two blocks are requested before anything else: they fly while the observations are prepared
    DMF_LOAD(1, 1)
    // ---- prep: normalise, clip, bf16, into B-fragment order (learning/normalizer.py:95-98).  A thread owns input column t (and t + 256): consecutive lanes read
    // consecutive floats of a row, mean and 1 / std are read once per thread, and all row reads are independent and requested up front (one latency, not 32)
    constexpr int NCOL = (KS1 * 32 + 255) / 256;
    float px[NCOL][R], pmean[NCOL], pinv[NCOL];
    {
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int k = t + 256 * i; sbias[k] = (k < 1024) ? p.b1[k] : p.b2[k - 1024]; }
        const int SS = p.S - io.G;
#pragma unroll
        for (int kk = 0; kk < NCOL; ++kk) {
            const int k = t + 256 * kk, kc = k < p.S ? k : p.S - 1;
            pmean[kk] = p.s_mean[kc]; pinv[kk] = p.s_inv_std[kc];
            const bool from_goal = kc >= SS;
            const float* const base = from_goal ? io.goals + (kc - SS) : io.states + kc;
            const size_t stride = from_goal ? (size_t)io.G : (size_t)SS;
#pragma unroll
            for (int m = 0; m < R; ++m) { int row = row0 + m; if (row >= io.M) row = io.M - 1; px[kk][m] = base[(size_t)row * stride]; }
        }
        DMP_SCHED_FENCE();
    }
    // the head's noise does not depend on the net: drawn now, while the weight and observation requests are in flight (4 Philox + Box-Muller per owned action tile and lane)
    constexpr int NT3W = (N3T + 1) / 2;
    float hz[NT3W][4]; bool explore;
    {
        const int row = row0 + 16 * (w & 1) + c;
        explore = io.sample && row < io.M;
        if (explore && io.exp_rate < 1.0f) explore = philox_coin((uint32_t)(io.env_off + row), io.step, io.seed_lo, io.seed_hi) < io.exp_rate;
#pragma unroll
        for (int it = 0; it < NT3W; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 16 * ((w >> 1) + 2 * it) + 4 * g + r;
                hz[it][r] = (explore && col < p.A) ? philox_normal((uint32_t)(io.env_off + row), io.step * (uint32_t)p.A + (uint32_t)col, io.seed_lo, io.seed_hi) : 0.0f;
            }
    }

    {
        uint16_t* const s16h = reinterpret_cast<uint16_t*>(s16);
#pragma unroll
        for (int kk = 0; kk < NCOL; ++kk) {
            const int k = t + 256 * kk;
            if (k < KS1 * 32) {
                const int rec0 = (k >> 5) * 128 + 16 * ((k & 31) >> 3), e = k & 7;          // record of batch row m: rec0 + 64 (m >> 4) + (m & 15)
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    float v = fminf(fmaxf((px[kk][m] - pmean[kk]) * pinv[kk], -p.s_clip), p.s_clip);
                    if (k >= p.S) v = 0.0f;
                    s16h[(size_t)(rec0 + 64 * (m >> 4) + (m & 15)) * 8 + e] = f32_to_bf16(v);
                }
            }
        }
    }
    __syncthreads();

    DMF_STAMP(1)
    f32x4 acc2[8][2];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[n][b][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        f32x4 acc1[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[j][b][r] = 0.0f;
        // layer 1, features [256 q + 64 w, + 64): NB1 blocks of 2 k-steps x 4 feature tiles
#pragma unroll
        for (int b1 = 0; b1 < NB1; ++b1) {
            const int blk = q * NBQ + b1;
            DMF_LOAD((blk + 2) % 3, blk + 2)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ks = 2 * b1 + kk;
                const bf16x8 x0 = s16[(ks * 2 + 0) * 64 + l], x1 = s16[(ks * 2 + 1) * 64 + l];
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc1[j][0] = mfma16(ring[blk % 3][kk * 4 + j], x0, acc1[j][0]); acc1[j][1] = mfma16(ring[blk % 3][kk * 4 + j], x1, acc1[j][1]); }
            }
        }
        // bias + ReLU + bf16 -> this chunk's buffer, already in the B-fragment order of layer 2 (local k = 64 w + 16 j + 4 g + r)
        {
            uint16_t* const dst = reinterpret_cast<uint16_t*>(h1c + (q & 1) * LDS_H1C);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f0 = 256 * q + 64 * w + 16 * j + 4 * g;
                const int ksl = 2 * w + (j >> 1), ll = c + 16 * (2 * (j & 1) + (g >> 1));
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    store4(dst + ((size_t)((ksl * 2 + b) * 64 + ll)) * 8 + 4 * (g & 1), fmaxf(acc1[j][b][0] + sbias[f0], 0.0f), fmaxf(acc1[j][b][1] + sbias[f0 + 1], 0.0f),
                           fmaxf(acc1[j][b][2] + sbias[f0 + 2], 0.0f), fmaxf(acc1[j][b][3] + sbias[f0 + 3], 0.0f));
                }
            }
        }
        __syncthreads();
        if (q == 0) DMF_STAMP(2)
        // layer 2, outputs [128 w, + 128), k in [256 q, + 256): NB2 blocks of 1 k-step x 8 feature tiles
        {
            const bf16x8* const src = h1c + (q & 1) * LDS_H1C;
#pragma unroll
            for (int ksl = 0; ksl < NB2; ++ksl) {
                const int blk = q * NBQ + NB1 + ksl;
                DMF_LOAD((blk + 2) % 3, blk + 2)
                const bf16x8 x0 = src[(ksl * 2 + 0) * 64 + l], x1 = src[(ksl * 2 + 1) * 64 + l];
#pragma unroll
                for (int n = 0; n < 8; ++n) { acc2[n][0] = mfma16(ring[blk % 3][n], x0, acc2[n][0]); acc2[n][1] = mfma16(ring[blk % 3][n], x1, acc2[n][1]); }
            }
        }
    }
#undef DMF_LOAD
    DMF_STAMP(3)
    // the first action tile's 16 weight fragments of layer 3 are requested now (the ring is dead): they arrive during the epilogue and the barrier
    constexpr int KS3 = 16;
    bf16x8 wf[KS3];
    {
        const int ntc0 = (w >> 1) < N3T ? (w >> 1) : N3T - 1;
        DMP_SCHED_FENCE();
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(p.w3p + (((size_t)ntc0 * KS3 + ks) * 64 + l) * 8);
        DMP_SCHED_FENCE();
    }
    // ... and so are the head's per-column constants of every tile this wave owns (branch-free, clamped: one latency for all of them)
    float hc_ls[NT3W][4], hc_b3[NT3W][4], hc_as[NT3W][4], hc_am[NT3W][4];
#pragma unroll
    for (int it = 0; it < NT3W; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * ((w >> 1) + 2 * it) + 4 * g + r, cc = col < p.A ? col : p.A - 1;
            hc_ls[it][r] = p.logstd[cc]; hc_b3[it][r] = p.b3[cc]; hc_as[it][r] = p.a_std[cc]; hc_am[it][r] = p.a_mean[cc];
        }
    DMP_SCHED_FENCE();
    // layer-2 epilogue -> h2 (B-fragment order of layer 3: k = 128 w + 16 n + 4 g + r).  Every wave is past the barrier of the last chunk, so s16 and
    // h1c[0], which h2 overlays, are dead; h1c[1] (still being read by slower waves) lies behind them.
    {
        uint16_t* const dst = reinterpret_cast<uint16_t*>(h2);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int f0 = 128 * w + 16 * n + 4 * g;
            const int ks = 4 * w + (n >> 1), ll = c + 16 * (2 * (n & 1) + (g >> 1));
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                store4(dst + ((size_t)((ks * 2 + b) * 64 + ll)) * 8 + 4 * (g & 1), fmaxf(acc2[n][b][0] + sbias[1024 + f0], 0.0f), fmaxf(acc2[n][b][1] + sbias[1024 + f0 + 1], 0.0f),
                       fmaxf(acc2[n][b][2] + sbias[1024 + f0 + 2], 0.0f), fmaxf(acc2[n][b][3] + sbias[1024 + f0 + 3], 0.0f));
            }
        }
    }
    __syncthreads();
    DMF_STAMP(4)
    // layer 3 + Gaussian head: wave w owns batch tile w & 1 and the action tiles (w >> 1), (w >> 1) + 2, ...; K = 512
    {
        const int bt = w & 1, row = row0 + 16 * bt + c;
        float lp = 0.0f;
#pragma unroll
        for (int nt = (w >> 1), it = 0; it < (N3T + 1) / 2; ++it, nt += 2) {
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) acc = mfma16(wf[ks], h2[(ks * 2 + bt) * 64 + l], acc);
            if (it + 1 < (N3T + 1) / 2) {              // the next tile's fragments, in flight while this tile's head is computed
                const int ntn = nt + 2 < N3T ? nt + 2 : N3T - 1;
#pragma unroll
                for (int ks = 0; ks < KS3; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(p.w3p + (((size_t)ntn * KS3 + ks) * 64 + l) * 8);
            }
            if (nt < N3T) {
                if (io.exp_flags && nt == 0 && g == 0 && row < io.M) io.exp_flags[row] = explore ? 1 : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * nt + 4 * g + r;
                    if (col < p.A) {
                        const float ls = hc_ls[it][r];
                        const float z = hz[it][r];
                        const float na = acc[r] + hc_b3[it][r] + expf(ls) * z;
                        if (row < io.M) io.actions[(size_t)row * p.A + col] = na * hc_as[it][r] + hc_am[it][r];
                        lp += -0.5f * z * z - ls;
                    }
                }
            }
        }
        DMF_STAMP(5)
        // log-probability: sum over the four feature groups of a lane column, then over the two waves that share the batch tile
        lp += lane_xor_f(lp, 16); lp += lane_xor_f(lp, 32);
        if (g == 0) lp_part[w][c] = lp;
        __syncthreads();
        if (io.logp && w < 2 && g == 0 && row < io.M) io.logp[row] = lp_part[w][c] + lp_part[w + 2][c] - 0.5f * (float)p.A * 1.8378770664093453f;
        DMF_STAMP(6)
    }
#undef DMF_STAMP
}

}  // namespace dmp
