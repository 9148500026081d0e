// Host side of the on-device policy (dm_policy.h): packs the fp32 weights of the reference's actor into bf16 MFMA fragments
// and launches the three layer kernels.  Included at the end of dm_host.cpp (shares its runtime shim).
#include "dm_policy.h"

struct dm_policy {
    dmp::PolicyDev pd; int device_id = 0; int cap = 0;
    uint16_t *h1 = nullptr, *h2 = nullptr, *s16 = nullptr;
    std::vector<void*> allocs;
    ~dm_policy() { for (void* p : allocs) rt_free(p); if (h1) rt_free(h1); if (h2) rt_free(h2); if (s16) rt_free(s16); }
    void* up(const void* host, size_t bytes) { void* d = nullptr; if (rt_malloc(&d, bytes)) return nullptr; allocs.push_back(d); if (rt_h2d(d, host, bytes, 0)) return nullptr; return d; }
};

// fragment order [n-tile][k-step][lane][8]: lane l <-> column 16 nt + (l & 15), k = 32 ks + 8 (l >> 4) + i; W is [K x N] row-major
// (tf.layers.dense kernel layout: input index first)
static std::vector<uint16_t> pack_weights(const float* W, int K, int N, int Kp, int Np) {
    std::vector<uint16_t> out((size_t)Kp * Np, 0);
    const int KS = Kp / 32;
    for (int nt = 0; nt < Np / 16; ++nt) for (int ks = 0; ks < KS; ++ks) for (int l = 0; l < 64; ++l) for (int i = 0; i < 8; ++i) {
        const int n = 16 * nt + (l & 15), k = 32 * ks + 8 * (l >> 4) + i;
        out[(((size_t)nt * KS + ks) * 64 + l) * 8 + i] = (k < K && n < N) ? dmp::f32_to_bf16_host(W[(size_t)k * N + n]) : (uint16_t)0;
    }
    return out;
}

// the weight stream of k_policy_fused (dm_policy.h): per wave w, per layer-1 chunk q: K1 / 64 blocks {2 k-steps x feature tiles 16 q + 4 w + j}, then 8 blocks
// {k-step 8 q + ksl of layer 2 x feature tiles 8 w + n}; a block is 8 fragments of [lane][8] bf16 (1 KB each), fragments as pack_weights lays them out
static std::vector<uint16_t> pack_fused_stream(const std::vector<uint16_t>& w1p, const std::vector<uint16_t>& w2p, int K1) {
    const int KS1 = K1 / 32, NB1 = KS1 / 2, NBQ = NB1 + 8, NBLK = 4 * NBQ, KS2 = 1024 / 32;
    std::vector<uint16_t> out((size_t)4 * NBLK * 8 * 512, 0);
    for (int w = 0; w < 4; ++w) for (int q = 0; q < 4; ++q) {
        for (int b1 = 0; b1 < NB1; ++b1) for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 4; ++j) {
            const size_t dst = (((size_t)w * NBLK + q * NBQ + b1) * 8 + kk * 4 + j) * 512;
            const size_t src = ((size_t)(16 * q + 4 * w + j) * KS1 + 2 * b1 + kk) * 512;
            std::copy(w1p.begin() + src, w1p.begin() + src + 512, out.begin() + dst);
        }
        for (int ksl = 0; ksl < 8; ++ksl) for (int n = 0; n < 8; ++n) {
            const size_t dst = (((size_t)w * NBLK + q * NBQ + NB1 + ksl) * 8 + n) * 512;
            const size_t src = ((size_t)(8 * w + n) * KS2 + 8 * q + ksl) * 512;
            std::copy(w2p.begin() + src, w2p.begin() + src + 512, out.begin() + dst);
        }
    }
    return out;
}

extern "C" {

int dm_policy_create(int device_id, const dm_policy_params* pp, dm_policy** out) {
    if (!pp || !out) return fail("null argument");
    if (pp->state_dim < 1 || pp->action_dim < 1 || pp->hidden1 < 64 || pp->hidden2 < 64) return fail("dm_policy_create: bad layer widths");
    if (pp->hidden1 % 64 || pp->hidden2 % 64) return fail("dm_policy_create: hidden widths must be multiples of 64 (reference: 1024, 512)");
    if (!pp->w1 || !pp->b1 || !pp->w2 || !pp->b2 || !pp->w3 || !pp->b3) return fail("dm_policy_create: null weights");
#ifndef DM_EMU
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device available: libdm_hip.so has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("invalid device_id");
#endif
    DevGuard guard(device_id);
    dm_policy* p = new dm_policy(); p->device_id = device_id;
    dmp::PolicyDev& d = p->pd; memset(&d, 0, sizeof(d));
    d.S = pp->state_dim; d.H1 = pp->hidden1; d.H2 = pp->hidden2; d.A = pp->action_dim;
    d.K1 = (d.S + 63) / 64 * 64; d.N3 = (d.A + 31) / 32 * 32;
    std::vector<uint16_t> w1 = pack_weights(pp->w1, d.S, d.H1, d.K1, d.H1), w2 = pack_weights(pp->w2, d.H1, d.H2, d.H1, d.H2), w3 = pack_weights(pp->w3, d.H2, d.A, d.H2, d.N3);
    std::vector<float> b3(d.N3, 0.0f), sm(d.S, 0.0f), si(d.S, 1.0f), am(d.A, 0.0f), as(d.A, 1.0f), ls(d.A, 0.0f);
    for (int i = 0; i < d.A; ++i) { b3[i] = pp->b3[i]; if (pp->a_mean) am[i] = pp->a_mean[i]; if (pp->a_std) as[i] = pp->a_std[i]; if (pp->logstd) ls[i] = pp->logstd[i]; }
    for (int i = 0; i < d.S; ++i) { if (pp->s_mean) sm[i] = pp->s_mean[i]; if (pp->s_std) si[i] = 1.0f / pp->s_std[i]; }
    // one-launch actor: compiled for the reference's widths (1024, 512), K1 = 256 / 384 and up to 64 action slots; a K1 of 320 is padded up to 384
    d.wfs = nullptr;
    if (d.H1 == 1024 && d.H2 == 512 && d.K1 <= 384 && d.N3 <= 64) {
        if (d.K1 == 320) { d.K1 = 384; w1 = pack_weights(pp->w1, d.S, d.H1, d.K1, d.H1); }
        if (d.K1 < 256) { d.K1 = 256; w1 = pack_weights(pp->w1, d.S, d.H1, d.K1, d.H1); }
        std::vector<uint16_t> fs = pack_fused_stream(w1, w2, d.K1);
        d.wfs = (const uint16_t*)p->up(fs.data(), fs.size() * 2);
        if (!d.wfs) { delete p; return fail("device allocation failed"); }
    }
    d.w1p = (const uint16_t*)p->up(w1.data(), w1.size() * 2); d.w2p = (const uint16_t*)p->up(w2.data(), w2.size() * 2); d.w3p = (const uint16_t*)p->up(w3.data(), w3.size() * 2);
    d.b1 = (const float*)p->up(pp->b1, sizeof(float) * d.H1); d.b2 = (const float*)p->up(pp->b2, sizeof(float) * d.H2); d.b3 = (const float*)p->up(b3.data(), sizeof(float) * d.N3);
    d.s_mean = (const float*)p->up(sm.data(), sizeof(float) * d.S); d.s_inv_std = (const float*)p->up(si.data(), sizeof(float) * d.S);
    d.a_mean = (const float*)p->up(am.data(), sizeof(float) * d.A); d.a_std = (const float*)p->up(as.data(), sizeof(float) * d.A);
    d.logstd = (const float*)p->up(ls.data(), sizeof(float) * d.A);
    d.s_clip = (pp->s_clip > 0) ? (float)pp->s_clip : std::numeric_limits<float>::infinity();
    if (!d.w1p || !d.w2p || !d.w3p || !d.b1 || !d.b2 || !d.b3 || !d.s_mean || !d.s_inv_std || !d.a_mean || !d.a_std || !d.logstd) { delete p; return fail("device allocation failed"); }
    *out = p;
    return 0;
}

int dm_policy_destroy(dm_policy* p) { if (!p) return 0; DevGuard guard(p->device_id); delete p; return 0; }

int dm_policy_forward_ex(dm_policy* p, const float* states_dev, const float* goals_dev, int goal_dim, int n, float* actions_dev, float* logp_dev,
                         int32_t* exp_flags_dev, double exp_rate, int sample, uint64_t seed, uint32_t step, int env_id_offset, void* hip_stream);

int dm_policy_forward(dm_policy* p, const float* states_dev, int n, float* actions_dev, float* logp_dev, int sample,
                      uint64_t seed, uint32_t step, int env_id_offset, void* hip_stream) {
    return dm_policy_forward_ex(p, states_dev, nullptr, 0, n, actions_dev, logp_dev, nullptr, 1.0, sample, seed, step, env_id_offset, hip_stream);
}

int dm_policy_forward_ex(dm_policy* p, const float* states_dev, const float* goals_dev, int goal_dim, int n, float* actions_dev, float* logp_dev,
                         int32_t* exp_flags_dev, double exp_rate, int sample, uint64_t seed, uint32_t step, int env_id_offset, void* hip_stream) {
    if (!p || !states_dev || !actions_dev) return fail("null argument");
    if (goal_dim < 0 || goal_dim >= p->pd.S || (goal_dim > 0 && !goals_dev)) return fail("dm_policy_forward_ex: goal_dim must be in [0, state_dim) with a goal block when positive (state_dim counts the goal columns)");
    if (!(exp_rate >= 0.0 && exp_rate <= 1.0)) return fail("dm_policy_forward_ex: exp_rate must be in [0, 1]");
    if (n <= 0) return 0;
    DevGuard guard(p->device_id);
    rt_stream stream = (rt_stream)hip_stream;
    if (n > p->cap) {                       // hidden activations: n x (H1 + H2) bf16, grown on demand
        rt_sync(stream);
        if (p->h1) rt_free(p->h1); if (p->h2) rt_free(p->h2); if (p->s16) rt_free(p->s16); p->h1 = p->h2 = p->s16 = nullptr; p->cap = 0;
        void *a = nullptr, *b = nullptr, *c = nullptr;
        if (rt_malloc(&a, (size_t)n * p->pd.H1 * 2) || rt_malloc(&b, (size_t)n * p->pd.H2 * 2) || rt_malloc(&c, (size_t)n * p->pd.K1 * 2)) { if (a) rt_free(a); if (b) rt_free(b); return fail("device allocation failed"); }
        p->h1 = (uint16_t*)a; p->h2 = (uint16_t*)b; p->s16 = (uint16_t*)c; p->cap = n;
    }
    dmp::PolicyIO io; memset(&io, 0, sizeof(io));
    io.states = states_dev; io.s16 = p->s16; io.h1 = p->h1; io.h2 = p->h2; io.actions = actions_dev; io.logp = logp_dev; io.M = n; io.sample = sample ? 1 : 0;
    io.seed_lo = (uint32_t)seed; io.seed_hi = (uint32_t)(seed >> 32); io.step = step; io.env_off = env_id_offset;
    if (const char* pr = getenv("DM_POLICY_PROBE")) io.probe = atoi(pr);
    io.goals = goal_dim ? goals_dev : nullptr; io.G = goal_dim; io.exp_rate = (float)exp_rate; io.exp_flags = exp_flags_dev;
    const dmp::PolicyDev& d = p->pd;
    // one launch for the whole actor (k_policy_fused) where it is compiled for the widths; DM_POLICY_LAYERED=1 keeps the per-layer kernels (A/B, tests)
    if (d.wfs && getenv("DM_POLICY_LAYERED") == nullptr) {
        const unsigned grid = (unsigned)((n + 31) / 32);
#ifndef DM_EMU
        static unsigned long long* prof_buf = nullptr; static int prof_calls = 0;
        if (io.probe == 2) { if (!prof_buf && hipMalloc((void**)&prof_buf, (size_t)8192 * 8 * 8) != hipSuccess) prof_buf = nullptr; io.prof = grid <= 8192 ? prof_buf : nullptr; }
#endif
        if (d.K1 == 256 && d.N3 == 32) RT_LAUNCH4((dmp::k_policy_fused<8, 2>), grid, stream, d, io);
        else if (d.K1 == 256) RT_LAUNCH4((dmp::k_policy_fused<8, 4>), grid, stream, d, io);
        else if (d.N3 == 32) RT_LAUNCH4((dmp::k_policy_fused<12, 2>), grid, stream, d, io);
        else RT_LAUNCH4((dmp::k_policy_fused<12, 4>), grid, stream, d, io);
#ifndef DM_EMU
        hipError_t le0 = hipGetLastError(); if (le0 != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le0));
        if (io.prof && ++prof_calls == 100) {          // DM_POLICY_PROBE=2: phase times of the 100th launch (100 MHz constant clock -> ns), mean over the workgroups
            (void)hipStreamSynchronize(stream);
            std::vector<unsigned long long> h((size_t)grid * 8); (void)hipMemcpy(h.data(), prof_buf, h.size() * 8, hipMemcpyDeviceToHost);
            double acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long t0 = ~0ull, t1 = 0;
            for (unsigned b = 0; b < grid; ++b) { for (int i = 0; i < 6; ++i) acc[i] += (double)(h[b * 8 + i + 1] - h[b * 8 + i]); t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 6]); }
            fprintf(stderr, "k_policy_fused phases (s_memtime ticks, mean of %u workgroups): weights + observations requested, noise drawn, observations to LDS %.0f | chunk 0 layer 1 %.0f | rest of the chunks %.0f | layer-2 epilogue + barrier %.0f | layer 3 + head %.0f | logp %.0f || first start -> last end %.0f\n",
                    grid, acc[0] / grid, acc[1] / grid, acc[2] / grid, acc[3] / grid, acc[4] / grid, acc[5] / grid, (double)(t1 - t0));
        }
#endif
        return 0;
    }
    // tiles sized so that every launch has at least ~1 wave per SIMD at 4096 rows: 64 x 64 (layer 1), 32 x 64 (layer 2), 16 x 32 (layer 3)
    RT_LAUNCH(dmp::k_policy_prep, n, stream, d, io);
    // layers 1 and 2: the LDS-tiled four-wave GEMM when the width allows it, the one-wave kernel otherwise
    const bool tiled = (getenv("DM_POLICY_ONE_WAVE") == nullptr);
    // 64-row tiles by default (measured: 45 / 82 us at 4096 / 16384 rows against 56 / 85 us with 128-row tiles: more workgroups per CU hide more
    // of the L2 latency than the bigger tile saves in traffic); DM_POLICY_TILE=128 selects the tall tile
    bool big1 = false, big2 = false;
    if (const char* tl = getenv("DM_POLICY_TILE")) big1 = big2 = (atoi(tl) == 128);
    if (tiled && d.H1 % 128 == 0) {
        if (big1) RT_LAUNCH4((dmp::k_policy_gemm<0, 128>), ((n + 127) / 128) * (d.H1 / 128), stream, d, io);
        else RT_LAUNCH4((dmp::k_policy_gemm<0, 64>), ((n + 63) / 64) * (d.H1 / 128), stream, d, io);
    } else RT_LAUNCH((dmp::k_policy_layer<0, 4, 4>), ((n + 63) / 64) * (d.H1 / 64), stream, d, io);
    if (tiled && d.H2 % 128 == 0) {
        if (big2) RT_LAUNCH4((dmp::k_policy_gemm<1, 128>), ((n + 127) / 128) * (d.H2 / 128), stream, d, io);
        else RT_LAUNCH4((dmp::k_policy_gemm<1, 64>), ((n + 63) / 64) * (d.H2 / 128), stream, d, io);
    } else RT_LAUNCH((dmp::k_policy_layer<1, 2, 4>), ((n + 31) / 32) * (d.H2 / 64), stream, d, io);
    RT_LAUNCH((dmp::k_policy_layer<2, 1, 2>), (n + 15) / 16, stream, d, io);   // one workgroup per 16 rows owns all N3 columns (logp is a row sum)
#ifndef DM_EMU
    hipError_t le = hipGetLastError(); if (le != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
#endif
    return 0;
}

}  // extern "C"
