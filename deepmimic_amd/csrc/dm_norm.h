// Running observation statistics on the device: the Normalizer of the reference's learner (learning/normalizer.py:6-152)
// for records that never leave HBM -- record() is a column sum / sum of squares over the [n x size] fp32 record block of a control step,
// update() folds the pending sums into mean / mean_sq / std with the reference's group rules, and the result is written where the policy
// kernels read it (dm_policy.h: s_mean, s_inv_std).  Included at the end of dm_host.cpp (shares its runtime shim).
//
// HBM-bound integer-light work: record() reads every element once, coalesced along the columns (thread <-> column, a workgroup owns
// a slab of consecutive rows), accumulates in fp64 and leaves one partial per workgroup; a second pass (one wavefront per column) adds the
// partials in a fixed order, so the sums are deterministic (no atomics).  4096 x 227 floats = 3.7 MB per control step.
#pragma once

namespace dmn {

constexpr int kRows = 16;            // fewest rows per workgroup of the record pass
constexpr int kMaxGroups = 1024;     // ... and the most workgroups (4 per CU): a workgroup owns a contiguous slab of ceil(n / groups) rows
constexpr int kThreads = 256;
struct alignas(16) F4 { float x, y, z, w; };      // one 16-byte global load / store

// pass 1: partial[b][0 .. size) = sum over the rows of workgroup b of x[r][c]; partial[b][size .. 2 size) = sum of squares.
// thread <-> column: the 64 lanes of a wave read 64 consecutive floats of a row (coalesced); the row loop is unrolled so that 8 rows are in flight.
__global__ void __launch_bounds__(256) k_norm_partial(const float* __restrict__ x, int n, int size, int rows_per_group, double* __restrict__ partial) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int r0 = b * rows_per_group, r1 = (r0 + rows_per_group < n) ? r0 + rows_per_group : n;
    for (int c = t; c < size; c += kThreads) {
        double s = 0, q = 0;
        const float* col = x + c;
#pragma unroll 8
        for (int r = r0; r < r1; ++r) { const double v = (double)col[(size_t)r * size]; s += v; q += v * v; }
        partial[(size_t)b * 2 * size + c] = s; partial[(size_t)b * 2 * size + size + c] = q;
    }
}

// pass 2: one wavefront per column c < 2 size: pending[1 + c] += sum_b partial[b][c].  Lane l adds the partials b = l, l + 64, ... in that order, then the
// 64 lane sums are added by a fixed tree through LDS: the order never depends on timing, so the statistics are reproducible bit for bit.
__global__ void __launch_bounds__(64) k_norm_fold(const double* __restrict__ partial, int ngroups, int n, int size, double* __restrict__ pending) {
    __shared__ double red[64];
    const int c = blockIdx.x, l = threadIdx.x;
    double s = 0;
    for (int b = l; b < ngroups; b += 64) s += partial[(size_t)b * 2 * size + c];
    red[l] = s;
    __syncthreads();
    for (int w = 32; w >= 1; w >>= 1) {
        if (l < w) red[l] += red[l + w];
        __syncthreads();
    }
    if (l == 0) { pending[1 + c] += red[0]; if (c == 0) pending[0] += (double)n; }
}

// Normalizer.update (learning/normalizer.py:47-73) with _process_group_data (141-149) and calc_std (109-115), one workgroup.
// state = {count, mean[size], mean_sq[size], std[size]}; group_id[c]: -1 keep the old value, 0 per element, > 0 the average of the NEW data over
// the group's members (member lists: grp_start[c] .. grp_start[c] + grp_len[c] into grp_idx, in index order like np.mean over group.indices).
__global__ void __launch_bounds__(256) k_norm_update(double* __restrict__ state, double* __restrict__ pending, int size, const int* __restrict__ group_id,
                                                      const int* __restrict__ grp_start, const int* __restrict__ grp_len, const int* __restrict__ grp_idx, double eps,
                                                      float* __restrict__ mean_f, float* __restrict__ inv_std_f) {
    const int t = threadIdx.x;
    const double new_count = pending[0], count = state[0];
    if (new_count > 0) {
        const double new_total = count + new_count, w_old = count / new_total, w_new = new_count / new_total;
        double* mean = state + 1; double* mean_sq = state + 1 + size; double* sd = state + 1 + 2 * size;
        for (int c = t; c < size; c += kThreads) {
            const int g = group_id[c];
            double nm, nq;
            if (g == -1) { nm = mean[c]; nq = mean_sq[c]; }
            else if (g == 0) { nm = pending[1 + c] / new_count; nq = pending[1 + size + c] / new_count; }
            else {
                double a = 0, b = 0;
                for (int k = 0; k < grp_len[c]; ++k) { const int j = grp_idx[grp_start[c] + k]; a += pending[1 + j] / new_count; b += pending[1 + size + j] / new_count; }
                nm = a / grp_len[c]; nq = b / grp_len[c];
            }
            const double m = w_old * mean[c] + w_new * nm, q = w_old * mean_sq[c] + w_new * nq;
            double var = q - m * m; if (var < 0) var = 0;
            double s = sqrt(var); if (s < eps) s = eps;
            mean[c] = m; mean_sq[c] = q; sd[c] = s;
            mean_f[c] = (float)m; inv_std_f[c] = (float)(1.0 / s);
        }
    }
    __syncthreads();                 // every thread has read pending[0] / its columns before they are cleared
    if (new_count > 0) {
        for (int c = t; c < 2 * size; c += kThreads) pending[1 + c] = 0;
        if (t == 0) { state[0] = count + new_count; pending[0] = 0; }
    }
}

// Normalizer.normalize (learning/normalizer.py:95-98): out = clip((x - mean) / std, -clip, clip), fp32.  A thread owns kQuads quads of four consecutive
// elements of the flat [n x size] array, a workgroup a contiguous run of kQuads x 1024 floats (quad u of thread t at 1024 u + 4 t: every load / store
// instruction of a wave is one contiguous KB, kQuads of them in flight); the column index wraps inside a quad, size need not divide by 4.
constexpr int kQuads = 4;
constexpr int kLdsCols = 2048;      // widest row whose statistics are staged in LDS (16 KB)
__global__ void __launch_bounds__(256) k_norm_apply(const float* __restrict__ x, int total, int size, const float* __restrict__ mean_f, const float* __restrict__ inv_std_f,
                                                     float clip, float* __restrict__ out, int vec_ok) {
    // mean / 1/std staged in LDS (8 gathers per quad: as global loads they out-numbered the payload's own load and store 8 : 2 and held the kernel at 3.5 TB/s)
    __shared__ float sm[2 * kLdsCols];
    const bool staged = size <= kLdsCols;
    if (staged) {
        for (int c = threadIdx.x; c < size; c += kThreads) { sm[c] = mean_f[c]; sm[kLdsCols + c] = inv_std_f[c]; }
        __syncthreads();
    }
    const long long base = (long long)blockIdx.x * (kQuads * 4 * kThreads) + 4 * threadIdx.x;
    float v[kQuads][4]; int cnt[kQuads];
#pragma unroll
    for (int u = 0; u < kQuads; ++u) {
        const long long i0 = base + (long long)u * 4 * kThreads;
        cnt[u] = (i0 >= total) ? 0 : ((total - i0 < 4) ? (int)(total - i0) : 4);
        if (vec_ok && cnt[u] == 4) { const F4 q = *reinterpret_cast<const F4*>(x + i0); v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w; }
        else for (int k = 0; k < cnt[u]; ++k) v[u][k] = x[i0 + k];
    }
#pragma unroll
    for (int u = 0; u < kQuads; ++u) {
        const long long i0 = base + (long long)u * 4 * kThreads;
        if (cnt[u] == 0) continue;
        int c = (int)(i0 % size);
        for (int k = 0; k < cnt[u]; ++k) {
            const float y = staged ? (v[u][k] - sm[c]) * sm[kLdsCols + c] : (v[u][k] - mean_f[c]) * inv_std_f[c];
            v[u][k] = y < -clip ? -clip : (y > clip ? clip : y);
            if (++c == size) c = 0;
        }
        if (vec_ok && cnt[u] == 4) { F4 q; q.x = v[u][0]; q.y = v[u][1]; q.z = v[u][2]; q.w = v[u][3]; *reinterpret_cast<F4*>(out + i0) = q; }
        else for (int k = 0; k < cnt[u]; ++k) out[i0 + k] = v[u][k];
    }
}

}  // namespace dmn

struct dm_normalizer {
    int device_id = 0, size = 0; double eps = 0.02, clip = 0;
    double *state = nullptr, *pending = nullptr, *partial = nullptr; int partial_blocks = 0;
    float *mean_f = nullptr, *inv_std_f = nullptr, *stage = nullptr; int stage_rows = 0;
    int *group_id = nullptr, *grp_start = nullptr, *grp_len = nullptr, *grp_idx = nullptr;
    // ONE stream per normaliser: `stage`, `partial` and `pending` are single buffers that every call reuses, ordered only by the stream the calls are issued
    // on.  The stream of the last record is remembered; a call that arrives on another one first waits for the old stream's work.
    rt_stream last_stream = 0; bool used = false;
    void order_on(rt_stream s) { if (used && s != last_stream) rt_sync(last_stream); last_stream = s; used = true; }
    ~dm_normalizer() { for (void* p : {(void*)state, (void*)pending, (void*)partial, (void*)mean_f, (void*)inv_std_f, (void*)stage, (void*)group_id, (void*)grp_start, (void*)grp_len, (void*)grp_idx}) if (p) rt_free(p); }
};

extern "C" {

int dm_norm_create(int device_id, int size, const int32_t* group_ids, double eps, double clip, dm_normalizer** out) {
    if (!out) return fail("null argument");
    if (size < 1 || size > (1 << 20)) return fail("dm_norm_create: size out of range");
    if (!(eps > 0)) return fail("dm_norm_create: eps must be positive (learning/normalizer.py: 0.02)");
#ifndef DM_EMU
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device available: libdm_hip.so has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail("invalid device_id");
#endif
    DevGuard guard(device_id);
    // groups (learning/normalizer.py:124-139): NULL ids = one NORM_GROUP_SINGLE group; members of a group id > 0 in index order
    std::vector<int> gid(size, 0), gs(size, 0), gl(size, 0), gi;
    if (group_ids) for (int c = 0; c < size; ++c) { if (group_ids[c] < -1) return fail("dm_norm_create: group ids are -1 (none), 0 (single) or positive"); gid[c] = group_ids[c]; }
    {
        std::map<int, std::vector<int>> members;
        for (int c = 0; c < size; ++c) if (gid[c] > 0) members[gid[c]].push_back(c);
        std::map<int, int> start;
        for (auto& kv : members) { start[kv.first] = (int)gi.size(); gi.insert(gi.end(), kv.second.begin(), kv.second.end()); }
        for (int c = 0; c < size; ++c) if (gid[c] > 0) { gs[c] = start[gid[c]]; gl[c] = (int)members[gid[c]].size(); }
        if (gi.empty()) gi.push_back(0);
    }
    dm_normalizer* h = new dm_normalizer(); h->device_id = device_id; h->size = size; h->eps = eps; h->clip = clip;
    void* p = nullptr;
    auto alloc = [&](size_t bytes) -> void* { p = nullptr; return rt_malloc(&p, bytes) ? nullptr : p; };
    h->state = (double*)alloc(sizeof(double) * (1 + 3 * (size_t)size)); h->pending = (double*)alloc(sizeof(double) * (1 + 2 * (size_t)size));
    h->mean_f = (float*)alloc(sizeof(float) * size); h->inv_std_f = (float*)alloc(sizeof(float) * size);
    h->group_id = (int*)alloc(sizeof(int) * size); h->grp_start = (int*)alloc(sizeof(int) * size); h->grp_len = (int*)alloc(sizeof(int) * size); h->grp_idx = (int*)alloc(sizeof(int) * gi.size());
    if (!h->state || !h->pending || !h->mean_f || !h->inv_std_f || !h->group_id || !h->grp_start || !h->grp_len || !h->grp_idx) { delete h; return fail("device allocation failed"); }
    // Normalizer.__init__: mean 0, mean_sq 0, std 1, count 0
    std::vector<double> st(1 + 3 * (size_t)size, 0.0); for (int c = 0; c < size; ++c) st[1 + 2 * (size_t)size + c] = 1.0;
    std::vector<float> mf(size, 0.0f), sf(size, 1.0f);
    if (rt_h2d(h->state, st.data(), sizeof(double) * st.size(), 0) || rt_h2d(h->mean_f, mf.data(), sizeof(float) * size, 0) || rt_h2d(h->inv_std_f, sf.data(), sizeof(float) * size, 0) ||
        rt_h2d(h->group_id, gid.data(), sizeof(int) * size, 0) || rt_h2d(h->grp_start, gs.data(), sizeof(int) * size, 0) || rt_h2d(h->grp_len, gl.data(), sizeof(int) * size, 0) ||
        rt_h2d(h->grp_idx, gi.data(), sizeof(int) * gi.size(), 0)) { delete h; return fail("copy failed"); }
    *out = h;
    return 0;
}

int dm_norm_destroy(dm_normalizer* h) { if (!h) return 0; DevGuard guard(h->device_id); delete h; return 0; }

int dm_norm_record(dm_normalizer* h, const float* x, int n, int flags, void* hip_stream) {
    if (!h || !x) return fail("null argument");
    if (n <= 0) return 0;
    DevGuard guard(h->device_id);
    rt_stream stream = (rt_stream)hip_stream;
    h->order_on(stream);
    const float* xd = x;
    if (!(flags & DM_DEVICE_PTRS)) {                    // host rows (tests, small callers): staged through a device buffer
        if (n > h->stage_rows) { rt_sync(stream); if (h->stage) rt_free(h->stage); h->stage = nullptr; h->stage_rows = 0; void* p = nullptr; if (rt_malloc(&p, sizeof(float) * (size_t)n * h->size)) return fail("device allocation failed"); h->stage = (float*)p; h->stage_rows = n; }
        if (rt_h2d(h->stage, x, sizeof(float) * (size_t)n * h->size, stream)) return fail("copy failed");
        xd = h->stage;
    }
    int nb = (n + dmn::kRows - 1) / dmn::kRows; if (nb > dmn::kMaxGroups) nb = dmn::kMaxGroups;
    const int rows_per_group = (n + nb - 1) / nb;
    nb = (n + rows_per_group - 1) / rows_per_group;                 // (no empty workgroup)
    if (nb > h->partial_blocks) { rt_sync(stream); if (h->partial) rt_free(h->partial); h->partial = nullptr; h->partial_blocks = 0; void* p = nullptr; if (rt_malloc(&p, sizeof(double) * (size_t)nb * 2 * h->size)) return fail("device allocation failed"); h->partial = (double*)p; h->partial_blocks = nb; }
    RT_LAUNCH4(dmn::k_norm_partial, nb, stream, xd, n, h->size, rows_per_group, h->partial);
    RT_LAUNCH(dmn::k_norm_fold, 2 * h->size, stream, (const double*)h->partial, nb, n, h->size, h->pending);
#ifndef DM_EMU
    hipError_t le = hipGetLastError(); if (le != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
#endif
    return 0;
}

int dm_norm_pending(dm_normalizer* h, double** dev_ptr, int* len) {
    if (!h || !dev_ptr) return fail("null argument");
    *dev_ptr = h->pending; if (len) *len = 1 + 2 * h->size;
    return 0;
}

int dm_norm_update(dm_normalizer* h, void* hip_stream) {
    if (!h) return fail("null argument");
    DevGuard guard(h->device_id);
    rt_stream stream = (rt_stream)hip_stream;
    h->order_on(stream);
    RT_LAUNCH4(dmn::k_norm_update, 1, stream, h->state, h->pending, h->size, (const int*)h->group_id, (const int*)h->grp_start, (const int*)h->grp_len, (const int*)h->grp_idx, h->eps, h->mean_f, h->inv_std_f);
#ifndef DM_EMU
    hipError_t le = hipGetLastError(); if (le != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
#endif
    return 0;
}

int dm_norm_set(dm_normalizer* h, const double* mean, const double* std_, int64_t count, void* hip_stream) {
    if (!h || !mean || !std_) return fail("null argument");
    DevGuard guard(h->device_id);
    rt_stream stream = (rt_stream)hip_stream;
    const int size = h->size;
    // set_mean_std (learning/normalizer.py:79-93): mean_sq = std^2 + mean^2; count < 0 keeps the current count (TFNormalizer.load sets it)
    std::vector<double> st(1 + 3 * (size_t)size); std::vector<float> mf(size), sf(size);
    if (rt_d2h(st.data(), h->state, sizeof(double), stream)) return fail("copy failed");
    if (count >= 0) st[0] = (double)count;
    for (int c = 0; c < size; ++c) {
        if (!(std_[c] > 0)) return fail("dm_norm_set: std must be positive");
        st[1 + c] = mean[c]; st[1 + size + c] = std_[c] * std_[c] + mean[c] * mean[c]; st[1 + 2 * (size_t)size + c] = std_[c];
        mf[c] = (float)mean[c]; sf[c] = (float)(1.0 / std_[c]);
    }
    if (rt_h2d(h->state, st.data(), sizeof(double) * st.size(), stream) || rt_h2d(h->mean_f, mf.data(), sizeof(float) * size, stream) || rt_h2d(h->inv_std_f, sf.data(), sizeof(float) * size, stream)) return fail("copy failed");
    return 0;
}

int dm_norm_get(dm_normalizer* h, double* mean, double* std_, double* mean_sq, int64_t* count, void* hip_stream) {
    if (!h) return fail("null argument");
    DevGuard guard(h->device_id);
    const int size = h->size;
    std::vector<double> st(1 + 3 * (size_t)size);
    if (rt_d2h(st.data(), h->state, sizeof(double) * st.size(), (rt_stream)hip_stream)) return fail("copy failed");
    if (count) *count = (int64_t)st[0];
    if (mean) memcpy(mean, &st[1], sizeof(double) * size);
    if (mean_sq) memcpy(mean_sq, &st[1 + size], sizeof(double) * size);
    if (std_) memcpy(std_, &st[1 + 2 * (size_t)size], sizeof(double) * size);
    return 0;
}

int dm_norm_normalize(dm_normalizer* h, const float* x_dev, int n, float* out_dev, void* hip_stream) {
    if (!h || !x_dev || !out_dev) return fail("null argument");
    if (n <= 0) return 0;
    DevGuard guard(h->device_id);
    const long long total = (long long)n * h->size;
    if (total > 0x7fffffffLL) return fail("dm_norm_normalize: too many elements for one call");
    const float clip = (h->clip > 0 && std::isfinite(h->clip)) ? (float)h->clip : std::numeric_limits<float>::infinity();
    const int vec_ok = (((uintptr_t)x_dev | (uintptr_t)out_dev) & 15) == 0 ? 1 : 0;
    RT_LAUNCH4(dmn::k_norm_apply, (int)((total + dmn::kQuads * 4 * dmn::kThreads - 1) / (dmn::kQuads * 4 * dmn::kThreads)), (rt_stream)hip_stream, x_dev, (int)total, h->size, (const float*)h->mean_f, (const float*)h->inv_std_f, clip, out_dev, vec_ok);
#ifndef DM_EMU
    hipError_t le = hipGetLastError(); if (le != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
#endif
    return 0;
}

// columns [first_column, first_column + size) of the policy's observation normaliser := this one (device-to-device, ordered on hip_stream): the actor then
// sees the statistics of the last update().  The reference keeps s_norm and g_norm apart (learning/rl_agent.py:212-222): bind the first at column 0 and
// the second at column state_size of a policy whose state_dim counts both blocks.
int dm_policy_bind_obs_normalizer(dm_policy* p, dm_normalizer* h, int first_column, void* hip_stream) {
    if (!p || !h) return fail("null argument");
    if (first_column < 0 || first_column + h->size > p->pd.S) return fail("dm_policy_bind_obs_normalizer: the normaliser's columns do not fit the policy's state_dim");
    if (p->device_id != h->device_id) return fail("dm_policy_bind_obs_normalizer: policy and normaliser live on different devices");
    DevGuard guard(p->device_id);
    float* dm = const_cast<float*>(p->pd.s_mean) + first_column; float* ds = const_cast<float*>(p->pd.s_inv_std) + first_column;
#ifdef DM_EMU
    memcpy(dm, h->mean_f, sizeof(float) * h->size); memcpy(ds, h->inv_std_f, sizeof(float) * h->size);
#else
    if (hipMemcpyAsync(dm, h->mean_f, sizeof(float) * h->size, hipMemcpyDeviceToDevice, (rt_stream)hip_stream) != hipSuccess ||
        hipMemcpyAsync(ds, h->inv_std_f, sizeof(float) * h->size, hipMemcpyDeviceToDevice, (rt_stream)hip_stream) != hipSuccess) return fail("copy failed");
#endif
    return 0;
}

}  // extern "C"
