// Host side of libdm_hip.so: builds the device tables from the raw reference-layout scene tables, owns the
// per-env HBM state and launches the kernels of dm_device.h behind the C-ABI of include/dm_hip.h.
//
// Init-time reference functions realised here (DeepMimicCore/...):
//   cKinTree::PostProcessJointMat anim/KinTree.cpp:1005-1020 | body / joint frames anim/KinTree.cpp:1022-1032,1108-1118
//   cRBDUtil::BuildMomentInertia* sim/RBDUtil.cpp:644-740 (+ Bullet 2.88 capsule inertia [EXT-BULLET])
//   cMotion::PostProcessFrames / BuildFrameVel anim/Motion.cpp:170-191,403-430; cKinTree::CalcVel anim/KinTree.cpp:1470-1508
//   cKinController::PostProcessMotion / CalcCycleRootDelta anim/KinController.cpp:131-161
//   cCtController::BuildCtrlParamOffset sim/CtController.cpp:183-195; cCtCtrlUtil offsets/scales/bounds sim/CtCtrlUtil.cpp
//   cSceneImitate::CalcJointWeights scenes/SceneImitate.cpp:236-248
//
// The same file is compiled twice: by hipcc for the product, and by g++ with -DDM_EMU (tests/emu) where
// "device memory" is host memory and a launch runs the unmodified kernels on the fiber emulator.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <random>
#include <sstream>
#include <limits>
#include <vector>

#include "../../include/dm_hip.h"
#include "dm_launch.h"
#include "dm_math.h"
#ifdef DM_EMU
#include "dm_device.h"      // the emulator's wave intrinsics, for the policy kernels of dm_policy.h (templates: nothing is instantiated here)
#endif

using namespace dmk;

static thread_local std::string g_err;
static int fail(const std::string& msg) { g_err = msg; return -1; }

// ---------------------------------------------------------------- runtime shim
#ifdef DM_EMU
static int rt_malloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : -1; }
static void rt_free(void* p) { free(p); }
static int rt_h2d(void* d, const void* h, size_t n, rt_stream) { memcpy(d, h, n); return 0; }
static int rt_d2h(void* h, const void* d, size_t n, rt_stream) { memcpy(h, d, n); return 0; }
static int rt_memset(void* d, int v, size_t n, rt_stream) { memset(d, v, n); return 0; }
static int rt_sync(rt_stream) { return 0; }
static int rt_h2d_async(void* d, const void* h, size_t n, rt_stream) { memcpy(d, h, n); return 0; }
static int rt_d2h_async(void* h, const void* d, size_t n, rt_stream) { memcpy(h, d, n); return 0; }
static int rt_d2h_rows(void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t rows, rt_stream) {
    for (size_t r = 0; r < rows; ++r) memcpy((char*)h + r * hpitch, (const char*)d + r * dpitch, width);
    return 0;
}
#else
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
// zero-fill synchronously: a null-stream hipMemset is not ordered against the ctx's non-blocking stream
static int rt_malloc(void** p, size_t n) { hipError_t e = hipMalloc(p, n ? n : 1); if (e != hipSuccess) return -1; if (hipMemset(*p, 0, n ? n : 1) != hipSuccess) return -1; return hipDeviceSynchronize() == hipSuccess ? 0 : -1; }
static void rt_free(void* p) { (void)hipFree(p); }
static int rt_h2d(void* d, const void* h, size_t n, rt_stream s) { if (hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) != hipSuccess) return -1; return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
static int rt_d2h(void* h, const void* d, size_t n, rt_stream s) { if (hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) != hipSuccess) return -1; return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
static int rt_memset(void* d, int v, size_t n, rt_stream s) { return hipMemsetAsync(d, v, n, s) == hipSuccess ? 0 : -1; }
static int rt_sync(rt_stream s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
static int rt_h2d_async(void* d, const void* h, size_t n, rt_stream s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -1; }
static int rt_d2h_async(void* h, const void* d, size_t n, rt_stream s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) == hipSuccess ? 0 : -1; }
static int rt_d2h_rows(void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t rows, rt_stream s) {      // `rows` strided pieces in one copy
    if (hipMemcpy2DAsync(h, hpitch, d, dpitch, width, rows, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
#endif

// Every C-ABI entry point runs with the ctx's device current (and restores the caller's): a process may hold contexts on
// several GPUs, or torch may have another device current in this thread.
#ifdef DM_EMU
struct DevGuard { explicit DevGuard(int) {} };
static int launch_status(int rc) { return rc; }
#else
struct DevGuard {
    int prev = -1; bool switched = false;
    explicit DevGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess); }
    ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
};
// a failed launch must not return 0 with stale outputs
static int launch_status(int rc) {
    if (rc) return rc;
    hipError_t le = hipGetLastError();
    return le == hipSuccess ? 0 : fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
}
#endif

// ---------------------------------------------------------------- host-side (double) model description
enum { JD_TYPE = 0, JD_PARENT, JD_AX, JD_AY, JD_AZ, JD_ATX, JD_ATY, JD_ATZ, JD_LL0, JD_LL1, JD_LL2, JD_LH0, JD_LH1, JD_LH2,
       JD_TORQUE_LIM, JD_FORCE_LIM, JD_IS_EE, JD_DIFF_W, JD_PARAM_OFFSET, JD_MAX };
enum { BD_SHAPE = 0, BD_MASS, BD_COLGROUP, BD_FALL, BD_AX, BD_AY, BD_AZ, BD_ATX, BD_ATY, BD_ATZ, BD_P0, BD_P1, BD_P2, BD_MAX = 17 };

struct HostModel {
    int J = 0, P = 0, D = 0, A = 0, S = 0, F = 0, NC = 0, NL = 0, max_depth = 0;
    std::vector<int> parent, jtype, pose_off, dof_off, ndof, depth, act_off, is_ee, fall, brot_ident, arot_ident;
    std::vector<uint32_t> subtree_mask;
    std::vector<double> attach, attach_rot, battach, brot, mass, inertia, torque_lim, lim_lo, lim_hi, diffw, thresh, aabb_he;
    std::vector<int> dof_joint, dof_kind, dof_axis, dof_vidx; std::vector<uint64_t> dof_anc; std::vector<double> kp, kd;
    std::vector<int> cand_link; std::vector<double> cand_loc, cand_rad;
    std::vector<double> cap; std::vector<int> pair_code;
    std::vector<int> lim_joint;
    std::vector<double> frame_time, frames, frame_vel; double duration = 0; int loop = 0; double cycle_delta[3] = {0, 0, 0};
    int num_clips = 1; std::vector<int> clip_start, clip_loop; std::vector<double> clip_dur, clip_delta, clip_cdf;   // multi-clip datasets
    std::vector<double> s_off, s_scale, a_off, a_scale, a_min, a_max; std::vector<int> s_groups;
    dm_scene_tables cfg;
};

static M3<double> rot_euler(double x, double y, double z) {   // cMathUtil::RotateMat(euler) = Rz*Ry*Rx (MathUtil.cpp:159-186)
    double xs = sin(x), xc = cos(x), ys = sin(y), yc = cos(y), zs = sin(z), zc = cos(z);
    M3<double> r;
    r.m[0] = yc * zc; r.m[3] = yc * zs; r.m[6] = -ys;
    r.m[1] = xs * ys * zc - xc * zs; r.m[4] = xs * ys * zs + xc * zc; r.m[7] = xs * yc;
    r.m[2] = xc * ys * zc + xs * zs; r.m[5] = xc * ys * zs - xs * zc; r.m[8] = xc * yc;
    return r;
}
static bool is_identity(const M3<double>& r) { for (int i = 0; i < 9; ++i) if (fabs(r.m[i] - ((i % 4 == 0) ? 1.0 : 0.0)) > 1e-12) return false; return true; }

static int build_host_model(const dm_scene_tables& t, int max_contacts, HostModel& hm) {
    hm.cfg = t;
    const int J = t.num_joints;
    if (J < 1 || J > 32) return fail("num_joints must be in [1,32]");
    hm.J = J;
    auto jd = [&](int j, int c) { return t.joint_mat[j * JD_MAX + c]; };
    auto bd = [&](int j, int c) { return t.body_defs[j * BD_MAX + c]; };
    hm.parent.resize(J); hm.jtype.resize(J); hm.pose_off.resize(J); hm.dof_off.resize(J); hm.ndof.resize(J); hm.depth.resize(J);
    hm.act_off.resize(J); hm.is_ee.resize(J); hm.fall.resize(J); hm.brot_ident.resize(J); hm.arot_ident.resize(J); hm.subtree_mask.assign(J, 0);
    hm.attach.assign(J * 3, 0); hm.attach_rot.assign(J * 9, 0); hm.battach.assign(J * 3, 0); hm.brot.assign(J * 9, 0);
    hm.mass.resize(J); hm.inertia.assign(2 * J * 3, 0); hm.torque_lim.resize(J); hm.lim_lo.resize(J); hm.lim_hi.resize(J);
    hm.diffw.resize(J); hm.thresh.resize(J); hm.aabb_he.assign(J * 4, 0);
    int poff = 0, doff = 0, aoff = 0; double wsum = 0;
    for (int j = 0; j < J; ++j) {
        int type = (int)jd(j, JD_TYPE), par = (int)jd(j, JD_PARENT);
        if (j == 0 && par != -1) return fail("joint 0 must be the root");
        if (j > 0 && (par < 0 || par >= j)) return fail("parent id must be < child id");
        if (j > 0 && type != JT_SPHERICAL && type != JT_REVOLUTE && type != JT_FIXED) return fail("only spherical / revolute / fixed joints are supported on the imitate path");
        hm.parent[j] = par; hm.jtype[j] = type;
        int psz = (j == 0) ? 7 : (type == JT_SPHERICAL ? 4 : (type == JT_REVOLUTE ? 1 : 0));
        int dsz = (j == 0) ? 6 : (type == JT_SPHERICAL ? 3 : (type == JT_REVOLUTE ? 1 : 0));
        int asz = (j == 0) ? 0 : dsz;
        hm.pose_off[j] = poff; poff += psz;
        hm.dof_off[j] = doff; hm.ndof[j] = dsz; doff += dsz;
        hm.act_off[j] = aoff; aoff += asz;
        hm.depth[j] = (j == 0) ? 0 : hm.depth[par] + 1; hm.max_depth = std::max(hm.max_depth, hm.depth[j]);
        hm.is_ee[j] = jd(j, JD_IS_EE) != 0; hm.fall[j] = t.fall_mask ? (t.fall_mask[j] != 0) : (bd(j, BD_FALL) != 0);
        if (j > 0) for (int k = 0; k < 3; ++k) hm.attach[j * 3 + k] = jd(j, JD_AX + k);   // root attach is zeroed at load
        M3<double> ar = rot_euler(jd(j, JD_ATX), jd(j, JD_ATY), jd(j, JD_ATZ));
        if (j == 0 && !is_identity(ar)) return fail("root joint AttachTheta must be zero");
        for (int k = 0; k < 9; ++k) hm.attach_rot[j * 9 + k] = ar.m[k];
        hm.arot_ident[j] = is_identity(ar);
        for (int k = 0; k < 3; ++k) hm.battach[j * 3 + k] = bd(j, BD_AX + k);
        M3<double> br = rot_euler(bd(j, BD_ATX), bd(j, BD_ATY), bd(j, BD_ATZ));
        for (int k = 0; k < 9; ++k) hm.brot[j * 9 + k] = br.m[k];
        hm.brot_ident[j] = is_identity(br);
        double mass = bd(j, BD_MASS); hm.mass[j] = mass;
        int shape = (int)bd(j, BD_SHAPE); double p0 = bd(j, BD_P0), p1 = bd(j, BD_P1), p2 = bd(j, BD_P2);
        double* I0 = &hm.inertia[(0 * J + j) * 3]; double* I1 = &hm.inertia[(1 * J + j) * 3];
        double* he = &hm.aabb_he[j * 4];
        if (shape == SH_BOX) {
            I0[0] = mass / 12 * (p1 * p1 + p2 * p2); I0[1] = mass / 12 * (p0 * p0 + p2 * p2); I0[2] = mass / 12 * (p0 * p0 + p1 * p1);
            I1[0] = I0[0]; I1[1] = I0[1]; I1[2] = I0[2];
            hm.thresh[j] = 0.02 * 0.5 * sqrt(p0 * p0 + p1 * p1 + p2 * p2);
            he[0] = 0.5 * p0; he[1] = 0.5 * p1; he[2] = 0.5 * p2;
        } else if (shape == SH_CAPSULE) {
            double r = 0.5 * p0, h = p1;
            double c_vol = DM_PI * r * r * h, hs_vol = DM_PI * 2.0 / 3.0 * r * r * r, density = mass / (c_vol + 2 * hs_vol);
            double cm = c_vol * density, hsm = hs_vol * density;
            I0[0] = cm * (0.25 * r * r + (1.0 / 12.0) * h * h) + 2 * hsm * (0.4 * r * r + (3.0 / 8) * r * h + 0.25 * h * h);
            I0[1] = (0.5 * cm + 0.8 * hsm) * r * r; I0[2] = I0[0];
            double mg = 0.04 / t.world_scale;   // [EXT-BULLET] btCapsuleShape::calculateLocalInertia
            double lx = 2 * (r + mg), ly = 2 * (r + 0.5 * h + mg), lz = 2 * (r + mg);
            I1[0] = mass / 12 * (ly * ly + lz * lz); I1[1] = mass / 12 * (lx * lx + lz * lz); I1[2] = mass / 12 * (lx * lx + ly * ly);
            hm.thresh[j] = 0.02 * (r + 0.5 * h);
            he[0] = r; he[1] = r + 0.5 * h; he[2] = r;
        } else if (shape == SH_SPHERE) {
            double r = 0.5 * p0;
            I0[0] = I0[1] = I0[2] = 0.4 * mass * r * r; I1[0] = I1[1] = I1[2] = I0[0];
            hm.thresh[j] = 0.02 * r; he[0] = r; he[3] = 1;
        } else return fail("unsupported body shape (box / capsule / sphere only)");
        hm.torque_lim[j] = jd(j, JD_TORQUE_LIM); hm.lim_lo[j] = jd(j, JD_LL0); hm.lim_hi[j] = jd(j, JD_LH0);
        hm.diffw[j] = jd(j, JD_DIFF_W); wsum += fabs(hm.diffw[j]);
        if (type == JT_REVOLUTE && hm.lim_lo[j] <= hm.lim_hi[j]) hm.lim_joint.push_back(j);
    }
    for (int j = 0; j < J; ++j) hm.diffw[j] /= wsum;
    hm.P = poff; hm.D = doff; hm.A = aoff; hm.NL = (int)hm.lim_joint.size();
    hm.S = (t.enable_phase_input ? 1 : 0) + (J * 9 + 1) + J * 6;
    if (t.scene_goal == 5) hm.S += 15;      // cSceneDribbleAMP::GetTaskStateSize (SceneDribbleAMP.cpp:541-545): offsets 0, scales 1, group single
    if (hm.D > 64) return fail("more than 64 degrees of freedom");
    for (int j = 0; j < J; ++j) for (int k = j; k != -1; k = hm.parent[k]) hm.subtree_mask[k] |= (1u << j);
    // generalized velocities
    hm.dof_joint.resize(hm.D); hm.dof_kind.resize(hm.D); hm.dof_axis.resize(hm.D); hm.dof_vidx.resize(hm.D); hm.dof_anc.assign(hm.D, 0);
    hm.kp.assign(hm.D, 0); hm.kd.assign(hm.D, 0);
    for (int j = 0; j < J; ++j) for (int k = 0; k < hm.ndof[j]; ++k) {
        int i = hm.dof_off[j] + k;
        hm.dof_joint[i] = j;
        if (j == 0) { hm.dof_kind[i] = (k < 3) ? DK_ROOT_LIN : DK_ROOT_ANG; hm.dof_axis[i] = k % 3; hm.dof_vidx[i] = k; }
        else {
            hm.dof_kind[i] = (hm.jtype[j] == JT_SPHERICAL) ? DK_SPH : DK_REV; hm.dof_axis[i] = (hm.jtype[j] == JT_SPHERICAL) ? k : 2;
            hm.dof_vidx[i] = hm.pose_off[j] + k;
            hm.kp[i] = t.pd_params[j * 2]; hm.kd[i] = t.pd_params[j * 2 + 1];   // root gains are never used (ExpPDController.cpp:22-32)
        }
    }
    for (int i = 0; i < hm.D; ++i) for (int a = hm.dof_joint[i]; a != -1; a = hm.parent[a])
        for (int k = 0; k < hm.ndof[a]; ++k) hm.dof_anc[i] |= (1ull << (hm.dof_off[a] + k));
    // ground contact candidates: sphere centre, capsule end centres (radius offset along -y), box corners
    for (int j = 0; j < J; ++j) {
        int shape = (int)bd(j, BD_SHAPE); double p0 = bd(j, BD_P0), p1 = bd(j, BD_P1), p2 = bd(j, BD_P2);
        auto add = [&](double x, double y, double z, double r) { hm.cand_link.push_back(j); hm.cand_loc.push_back(x); hm.cand_loc.push_back(y); hm.cand_loc.push_back(z); hm.cand_rad.push_back(r); };
        if (shape == SH_SPHERE) add(0, 0, 0, 0.5 * p0);
        else if (shape == SH_CAPSULE) { add(0, 0.5 * p1, 0, 0.5 * p0); add(0, -0.5 * p1, 0, 0.5 * p0); }
        else if (shape == SH_BOX) for (int s = 0; s < 8; ++s) add((s & 1) ? -0.5 * p0 : 0.5 * p0, (s & 2) ? -0.5 * p1 : 0.5 * p1, (s & 4) ? -0.5 * p2 : 0.5 * p2, 0);
    }
    // self collision: capsule model of every link (sphere: point, capsule: own axis, box: inscribed capsule along the longest
    // extent) and the non-adjacent pairs in (i < j) order
    hm.cap.assign(J * 4, 0);
    for (int j = 0; j < J; ++j) {
        int shape = (int)bd(j, BD_SHAPE); double p0 = bd(j, BD_P0), p1 = bd(j, BD_P1), p2 = bd(j, BD_P2);
        double* c = &hm.cap[j * 4];
        if (shape == SH_SPHERE) c[3] = 0.5 * p0;
        else if (shape == SH_CAPSULE) { c[1] = 0.5 * p1; c[3] = 0.5 * p0; }
        else if (shape == SH_BOX) {
            double e[3] = {p0, p1, p2}; int a = 0; if (e[1] > e[a]) a = 1; if (e[2] > e[a]) a = 2;
            double r = 0.5 * std::min(e[(a + 1) % 3], e[(a + 2) % 3]);
            c[a] = std::max(0.0, 0.5 * e[a] - r); c[3] = r;
        }
    }
    if (!t.disable_self_collision)
        for (int i = 0; i < J; ++i) for (int j = i + 1; j < J; ++j) if (hm.parent[j] != i && hm.parent[i] != j) hm.pair_code.push_back(i | (j << 8));
    hm.NC = (int)hm.cand_link.size();
    if (hm.NC > 128) return fail("more than 128 ground-contact candidate points");
    if (hm.NL + 3 * max_contacts > kMaxRows) return fail("limit rows + 3*max_contacts exceeds 64 constraint rows");

    // ---- motion clip(s): a dataset (`--kin_ctrl clips`) is the concatenation of its clips, each processed like a single clip
    const int FT = t.num_frames, P = hm.P;
    const int NCL = (t.num_clips > 1) ? t.num_clips : 1;
    if (t.num_clips > 1 && (!t.clip_starts || !t.clip_weights || !t.clip_loops)) return fail("num_clips > 1 needs clip_starts / clip_weights / clip_loops");
    if (NCL > 1 && (t.clip_starts[0] != 0 || t.clip_starts[NCL] != FT)) return fail("clip_starts must span [0, num_frames]");
    hm.num_clips = NCL;
    hm.clip_start.assign(NCL + 1, 0); hm.clip_loop.assign(NCL, 0); hm.clip_dur.assign(NCL, 0); hm.clip_delta.assign((size_t)NCL * 3, 0); hm.clip_cdf.assign(NCL, 1);
    hm.frame_time.assign(FT, 0); hm.frames.assign((size_t)FT * P, 0); hm.frame_vel.assign((size_t)FT * P, 0);
    double wsum_clips = 0;
    for (int c = 0; c < NCL; ++c) {
        const int r0 = (NCL > 1) ? t.clip_starts[c] : 0, r1 = (NCL > 1) ? t.clip_starts[c + 1] : FT, F = r1 - r0;
        if (F < 2) return fail("motion needs at least 2 frames");
        hm.clip_start[c] = r0; hm.clip_start[c + 1] = r1; hm.clip_loop[c] = (NCL > 1) ? t.clip_loops[c] : t.loop;
        wsum_clips += (NCL > 1) ? t.clip_weights[c] : 1.0; hm.clip_cdf[c] = wsum_clips;
        double* ftime = &hm.frame_time[r0]; double* frames = &hm.frames[(size_t)r0 * P]; double* fvel = &hm.frame_vel[(size_t)r0 * P];
        double tcur = 0; const double* raw = t.frames + (size_t)r0 * (P + 1);
        double ox = raw[1], oz = raw[3];
        for (int f = 0; f < F; ++f) {
            ftime[f] = tcur; tcur += raw[(size_t)f * (P + 1)];
            double* fr = &frames[(size_t)f * P];
            for (int k = 0; k < P; ++k) fr[k] = raw[(size_t)f * (P + 1) + 1 + k];
            fr[0] -= ox; fr[2] -= oz;
            stq(fr + 3, qnormalize(ldq(fr + 3)));
            for (int j = 1; j < J; ++j) if (hm.jtype[j] == JT_SPHERICAL) stq(fr + hm.pose_off[j], qnormalize(ldq(fr + hm.pose_off[j])));
        }
        hm.clip_dur[c] = ftime[F - 1];
        for (int f = 0; f < F - 1; ++f) {
            double dt = ftime[f + 1] - ftime[f];
            const double* a = &frames[(size_t)f * P]; const double* b = a + P; double* v = &fvel[(size_t)f * P];
            for (int k = 0; k < 3; ++k) v[k] = (b[k] - a[k]) / dt;
            V3<double> wr = quat_to_rotvec(qmul(ldq(b + 3), qconj(ldq(a + 3))), 0.000001);
            v[3] = wr.x / dt; v[4] = wr.y / dt; v[5] = wr.z / dt;
            for (int j = 1; j < J; ++j) {
                int off = hm.pose_off[j];
                if (hm.jtype[j] == JT_SPHERICAL) { V3<double> w = quat_to_rotvec(qmul(qconj(ldq(a + off)), ldq(b + off)), 0.000001); v[off] = w.x / dt; v[off + 1] = w.y / dt; v[off + 2] = w.z / dt; }
                else if (hm.jtype[j] == JT_REVOLUTE) v[off] = (b[off] - a[off]) / dt;
            }
        }
        for (int k = 0; k < P; ++k) fvel[(size_t)(F - 1) * P + k] = fvel[(size_t)(F - 2) * P + k];
        {   // PostProcessMotion + CalcCycleRootDelta
            double bx = frames[0], bz = frames[2];
            for (int f = 0; f < F; ++f) { frames[(size_t)f * P] -= bx; frames[(size_t)f * P + 2] -= bz; }
            hm.clip_delta[(size_t)c * 3] = frames[(size_t)(F - 1) * P] - frames[0]; hm.clip_delta[(size_t)c * 3 + 1] = 0; hm.clip_delta[(size_t)c * 3 + 2] = frames[(size_t)(F - 1) * P + 2] - frames[2];
        }
    }
    for (int c = 0; c < NCL; ++c) hm.clip_cdf[c] /= wsum_clips;      // cClipsController::BuildClipsCDF (:196-212)
    // the model's own clip members describe clip 0
    hm.F = hm.clip_start[1]; hm.loop = hm.clip_loop[0]; hm.duration = hm.clip_dur[0];
    for (int k = 0; k < 3; ++k) hm.cycle_delta[k] = hm.clip_delta[k];
    // ---- offsets / scales / bounds handed to the learner
    hm.s_off.assign(hm.S, 0); hm.s_scale.assign(hm.S, 1); hm.s_groups.assign(hm.S, 0);
    if (t.enable_phase_input) { hm.s_off[0] = -0.5; hm.s_scale[0] = 2; hm.s_groups[0] = -1; }   // CtController.cpp:268-279,364-371
    hm.a_off.assign(hm.A, 0); hm.a_scale.assign(hm.A, 1); hm.a_min.assign(hm.A, 0); hm.a_max.assign(hm.A, 0);
    for (int j = 1; j < J; ++j) {
        int ao = hm.act_off[j];
        if (hm.jtype[j] == JT_SPHERICAL) for (int k = 0; k < 3; ++k) { hm.a_scale[ao + k] = 2.0 / (2 * DM_PI); hm.a_min[ao + k] = -2 * DM_PI; hm.a_max[ao + k] = 2 * DM_PI; }
        else if (hm.jtype[j] == JT_REVOLUTE) {
            double lo = hm.lim_lo[j], hi = hm.lim_hi[j];
            if (!(hi >= lo)) { lo = -DM_PI; hi = DM_PI; }
            hm.a_off[ao] = -0.5 * (hi + lo); hm.a_scale[ao] = 0.5 / (hi - lo);
            double mean = 0.5 * (hi + lo), delta = hi - lo; hm.a_min[ao] = mean - 2 * delta; hm.a_max[ao] = mean + 2 * delta;
        }
    }
    return 0;
}

// ---------------------------------------------------------------- device-side context, typed on the kernel precision
struct CtxBase {
    HostModel hm; int N = 0; int env_off = 0; uint64_t seed = 0; int precision = 32; int max_contacts = 20; int device_id = 0;
    rt_stream own_stream = 0, stream = 0;
    std::vector<void*> allocs;
    float *d_actions = nullptr, *d_states = nullptr, *d_rewards = nullptr; int *d_term = nullptr, *d_valid = nullptr, *d_end = nullptr;
    bool duo = false, duo_obj = true, upload_failed = false; int physics = 1;
    int* d_ids = nullptr; const int* step_ids = nullptr; int step_n_ids = 0;      // dm_step_envs: the subset the next step() call runs (device ids), cleared after it
    virtual ~CtxBase() { for (void* p : allocs) rt_free(p); }
    void* dalloc(size_t n) { void* p = nullptr; if (rt_malloc(&p, n) != 0) return nullptr; allocs.push_back(p); return p; }
    virtual int setup() = 0;
    virtual int reset(const int* ids_dev, int n, const double* kt_dev, const double* mt_dev) = 0;
    virtual int step(const float* actions_dev, double dt, int n_updates, float* states, float* rewards, int* term, int* valid, int* end, int flags, float* amp = nullptr) = 0;
    virtual int query(float* states, float* rewards, int* term, int* valid, int* end, float* amp = nullptr) = 0;
    virtual int amp_expert(int n, const double* times_dev, const double* gh_dev, float* out_dev) = 0;
    int amp_size = 0; float* d_amp = nullptr; uint64_t expert_calls = 0;
    int goal_size = 0; float* d_goals = nullptr;            // RecordGoal of the last emit (goal scenes)
    virtual int get_goal(double* out) = 0; virtual int set_goal(const double* in) = 0; virtual int get_clips(int* out) = 0; virtual int set_clips(const int* in) = 0;
    virtual int goal_aux(double* out, const double* in) = 0; virtual void set_mode(int test) = 0; virtual int pert_state(double* out, const double* in) = 0;
    virtual int obj_state(double* out, const double* in) = 0;
    virtual int manifolds(double* out, const double* in) = 0;
    virtual int amp_expert_clips(int n, const int* clips_dev, const double* times_dev, const double* gh_dev, float* out_dev) = 0;
    virtual int probe(int what, double dt) = 0;
    virtual int get_state(double* pose, double* vel, double* tar, double* kin, double* clk, int* flg) = 0;
    virtual int set_state(const double* pose, const double* vel, const double* tar, const double* kin, const double* clk, const int* flg) = 0;
    virtual int get_debug(const char* name, double* out) = 0;
    virtual int set_tau(const double* tau) = 0;
    virtual void set_time_limits(double lo, double hi) = 0;
    virtual void set_timer_exp(double ex) = 0;
    virtual int set_env_keys(const int* ids, int n, const uint64_t* seeds) = 0;
    virtual int draw_tape(const double* in, int unbind, double* hdr_out) = 0;
    virtual int draw_tape_envs(const int* ids, int n, const double* rows, double* hdr_out) = 0;
};

template <typename Real>
struct CtxT : CtxBase {
    ModelDev<Real> md; EnvState<Real> st; DebugTaps<Real> dbg; int cls = 0; long long* d_prof = nullptr; double* d_tape = nullptr;

    template <typename T, typename U> const T* up(const std::vector<U>& v) {
        std::vector<T> tmp(v.size()); for (size_t i = 0; i < v.size(); ++i) tmp[i] = (T)v[i];
        void* p = dalloc(sizeof(T) * std::max<size_t>(1, v.size()));
        if (!p || (!v.empty() && rt_h2d(p, tmp.data(), sizeof(T) * v.size(), stream) != 0)) upload_failed = true;
        return (const T*)p;
    }
    template <typename C> const uint32_t* build_mdl(int* words) {
        const HostModel& h = hm;
        MdlLds<Real, C>* b = new MdlLds<Real, C>();
        memset(b, 0, sizeof(*b));
        for (int j = 0; j < h.J; ++j) {
            b->link_info[j] = (h.parent[j] + 1) | (h.jtype[j] << 5) | (h.depth[j] << 8) | (h.pose_off[j] << 12) | (h.dof_off[j] << 19) |
                              (h.arot_ident[j] << 26) | (h.brot_ident[j] << 27) | ((h.is_ee[j] ? 1 : 0) << 28) | ((h.fall[j] ? 1 : 0) << 29);
            b->subtree_mask[j] = h.subtree_mask[j];
            uint64_t chain = 0;
            for (int a = j; a != -1; a = h.parent[a]) for (int k = 0; k < h.ndof[a]; ++k) chain |= (1ull << (h.dof_off[a] + k));
            b->chain_lo[j] = (uint32_t)(chain & 0xffffffffull); b->chain_hi[j] = (uint32_t)(chain >> 32);
            for (int k = 0; k < 3; ++k) { b->attach[j][k] = (Real)h.attach[j * 3 + k]; b->battach[j][k] = (Real)h.battach[j * 3 + k];
                                          b->inertia[0][j][k] = (Real)h.inertia[(0 * h.J + j) * 3 + k]; b->inertia[1][j][k] = (Real)h.inertia[(1 * h.J + j) * 3 + k]; }
            b->mass[j] = (Real)h.mass[j]; b->thresh[j] = (Real)h.thresh[j]; b->torque_lim[j] = (Real)h.torque_lim[j];
            for (int k = 0; k < 4; ++k) b->cap[j][k] = (Real)h.cap[j * 4 + k];
            b->kp[j] = (j > 0 && h.ndof[j] > 0) ? (Real)h.kp[h.dof_off[j]] : (Real)0; b->kd[j] = (j > 0 && h.ndof[j] > 0) ? (Real)h.kd[h.dof_off[j]] : (Real)0;
        }
        if (C::ROT) {
            int nb = 0, na = 0;
            for (int j = 0; j < h.J; ++j) {
                int idx = 0;
                if (!h.brot_ident[j]) { if (nb >= kMaxRotLinks) { delete b; *words = 0; fail("more than 4 links with a body attach rotation"); return nullptr; }
                                        for (int k = 0; k < 9; ++k) b->brot[C::ROT ? nb : 0][k] = (Real)h.brot[j * 9 + k]; idx |= ++nb; }
                if (!h.arot_ident[j]) { if (na >= kMaxRotLinks) { delete b; *words = 0; fail("more than 4 joints with an attach rotation"); return nullptr; }
                                        for (int k = 0; k < 9; ++k) b->attach_rot[C::ROT ? na : 0][k] = (Real)h.attach_rot[j * 9 + k]; idx |= (++na) << 4; }
                b->rot_idx[C::ROT ? j : 0] = idx;
            }
        }
        for (int i = 0; i < h.D; ++i) {
            b->dof_info[i] = h.dof_joint[i] | (h.dof_kind[i] << 8) | (h.dof_axis[i] << 10) | (h.dof_vidx[i] << 12);
        }
        for (int r = 0; r < h.NL; ++r) { int j = h.lim_joint[r]; b->lim_joint[r] = j; b->lim_lo[r] = (Real)h.lim_lo[j]; b->lim_hi[r] = (Real)h.lim_hi[j]; }
        const size_t bytes = (sizeof(*b) + 3) / 4 * 4;
        void* p = dalloc(bytes);
        if (!p || rt_h2d(p, b, sizeof(*b), stream) != 0) upload_failed = true;
        delete b;
        *words = (int)(bytes / 4);
        return (const uint32_t*)p;
    }
    int setup() override {
        const HostModel& h = hm; const dm_scene_tables& c = h.cfg;
        memset(&md, 0, sizeof(md)); memset(&dbg, 0, sizeof(dbg));
        md.J = h.J; md.P = h.P; md.D = h.D; md.A = h.A; md.S = h.S; md.F = h.F; md.NC = h.NC; md.NL = h.NL; md.max_depth = h.max_depth;
        bool any_rot = false;
        for (int j = 0; j < h.J; ++j) if (!h.arot_ident[j] || !h.brot_ident[j]) any_rot = true;
        if (h.NL > kMaxLim) return fail("more than 4 revolute joints with limits");
        md.physics = physics; md.NLJ = h.NL;
        if (physics == 2) {
            // both unilateral rows of every limit; the row budget (kMaxRows = 64 = limits + 3 per contact) then caps the contacts lower
            md.NL = 2 * h.NL;
            if (max_contacts > (kMaxRows - md.NL) / 3) max_contacts = (kMaxRows - md.NL) / 3;
        }
        if (h.J <= ClsBiped::NJ && h.D <= ClsBiped::ND && h.P <= ClsBiped::NP && h.NC <= ClsBiped::NCAP && !any_rot) cls = 0;
        else if (h.J <= ClsLarge::NJ && h.D <= ClsLarge::ND && h.P <= ClsLarge::NP && h.NC <= ClsLarge::NCAP) cls = 1;
        else return fail("character too large for the compiled kernel classes (J<=23, D<=64, P<=83, <=128 contact candidates)");
        if (c.scene_goal == 5) { if (cls != 0) return fail("dribble_amp is compiled for the biped class only"); cls = 2; }      // biped + one free body
        // a skeleton whose dof tree is one of the compiled topologies runs the branch-sparse, level-scheduled factor (dm_types.h
        // TopoTables); anything else of that size the dense class.  DM_TREE=0 keeps the dense class (A/B runs).
        if (cls == 1 && h.D == TopoDog3d::N) {
            std::vector<int> lam(h.D, -1);
            for (int j = 0; j < h.J; ++j) {
                int a = h.parent[j];
                while (a >= 0 && h.ndof[a] == 0) a = h.parent[a];
                for (int k = 0; k < h.ndof[j]; ++k) lam[h.dof_off[j] + k] = (k == 0) ? (a < 0 ? -1 : h.dof_off[a] + h.ndof[a] - 1) : h.dof_off[j] + k - 1;
            }
            bool same = true;
            for (int k = 0; k < h.D; ++k) if (lam[k] != TopoDog3d::PAR[k]) same = false;
            const char* tv = getenv("DM_TREE");
            if (same && !(tv && tv[0] == '0')) cls = 3;
        }
        // humanoid3d one character per wavefront (wave_packing 1): the same factor on its compiled topology.  The two-per-wave kernel (the
        // default for this class) keeps the dense factor.
        if (cls == 0 && (!duo || physics == 2 || (N % 2) != 0) && h.D == TopoHumanoid3d::N) {
            std::vector<int> lam(h.D, -1);
            for (int j = 0; j < h.J; ++j) {
                int a = h.parent[j];
                while (a >= 0 && h.ndof[a] == 0) a = h.parent[a];
                for (int k = 0; k < h.ndof[j]; ++k) lam[h.dof_off[j] + k] = (k == 0) ? (a < 0 ? -1 : h.dof_off[a] + h.ndof[a] - 1) : h.dof_off[j] + k - 1;
            }
            bool same = true;
            for (int k = 0; k < h.D; ++k) if (lam[k] != TopoHumanoid3d::PAR[k]) same = false;
            // opt-in (DM_TREE_BIPED=1): on humanoid3d the sparse factor saves too little (154 of 289 packed FMAs per factorisation) to pay for
            // the column build and the level bookkeeping inside a 128-VGPR budget -- measured 1.00 M env-steps/s at 4 waves / SIMD, 1.31 M
            // at 2, against 1.66 M for the dense one-per-wave kernel and 2.06 M two-per-wave (same box, profiles/r03_ab_biped_tree.json)
            const char* tv = getenv("DM_TREE_BIPED");
            if (same && tv && tv[0] == '1') cls = 4;
        }
        md.mdl_blob = (cls == 0 || cls == 2 || cls == 4) ? build_mdl<ClsBiped>(&md.mdl_words) : build_mdl<ClsLarge>(&md.mdl_words);
        md.act_off = up<int>(h.act_off); md.diffw = up<Real>(h.diffw); md.aabb_he = up<Real>(h.aabb_he);
        md.cand_link = up<int>(h.cand_link); md.cand_loc = up<Real>(h.cand_loc); md.cand_rad = up<Real>(h.cand_rad);
        md.pair_code = up<int>(h.pair_code); md.NPAIR = (int)h.pair_code.size();
        if (md.NPAIR > ((cls == 0 || cls == 2 || cls == 4) ? ClsBiped::NPAIRCAP : ClsLarge::NPAIRCAP)) return fail("too many self-collision pairs for the compiled kernel classes");
        md.frame_time = up<double>(h.frame_time); md.frames = up<Real>(h.frames); md.frame_vel = up<Real>(h.frame_vel);
        md.duration = h.duration; md.loop = h.loop; for (int k = 0; k < 3; ++k) { md.cycle_delta[k] = (Real)h.cycle_delta[k]; md.gravity[k] = (Real)c.gravity[k]; }
        md.num_sim_substeps = c.num_sim_substeps; md.solver_iters = c.solver_iters > 0 ? c.solver_iters : 10; md.max_contacts = max_contacts;
        md.friction = (Real)(c.friction > 0 ? c.friction : 0.9 * 0.9); md.erp = (Real)(c.erp > 0 ? c.erp : 0.2);
        md.report_dist = (Real)0.001; md.max_lin_vel = (Real)(100.0 / c.world_scale); md.max_ang_vel = (Real)100.0; md.lim_max_impulse = (Real)(100.0 / (c.world_scale * c.world_scale));
        md.slerp_one = (sizeof(Real) == 8) ? (Real)(1.0 - std::numeric_limits<double>::epsilon()) : (Real)(1.0 - 1e-6);
        md.sync_root_pos = c.sync_char_root_pos; md.sync_root_rot = c.sync_char_root_rot; md.enable_fall_end = c.enable_fall_end;
        md.enable_contact_fall = c.enable_char_contact_fall; md.enable_root_rot_fail = c.enable_root_rot_fail; md.enable_rand_placement = c.enable_rand_char_placement;
        md.enable_phase_input = c.enable_phase_input; md.record_world_root_pos = c.record_world_root_pos; md.record_world_root_rot = c.record_world_root_rot;
        md.query_period = 1.0 / (c.query_rate > 0 ? c.query_rate : 30.0);
        md.time_lim_min = c.time_lim_min; md.time_lim_max = c.time_lim_max; md.timer_exp = 0; md.seed = seed; md.draw_tape = nullptr;
        st.N = N;
        st.pose = (Real*)dalloc(sizeof(Real) * N * h.P); st.vel = (Real*)dalloc(sizeof(Real) * N * h.P); st.tar = (Real*)dalloc(sizeof(Real) * N * h.P);
        st.tau = (Real*)dalloc(sizeof(Real) * N * h.D); st.kin = (Real*)dalloc(sizeof(Real) * N * 8);
        st.clock = (double*)dalloc(sizeof(double) * N * 6); st.flag = (int*)dalloc(sizeof(int) * N * 4);
        if (st.clock) rt_memset(st.clock, 0, sizeof(double) * (size_t)N * 6, stream);   // word 5 of a clock row counts the env's substeps on borrowed lanes (two-per-wave kernel, round 6)
        if (st.kin) rt_memset(st.kin, 0, sizeof(Real) * (size_t)N * 8, stream);      // word 7 of a kin row is the env's fallback-substep counter (two-per-wave kernel), never reset by the device
        // overflow rows of the constraint-space matrix (rows RREG..63 of a character with more than RREG rows in a substep)
        { const int ovf = kMaxRows - ((cls == 0 || cls == 2 || cls == 4) ? ClsBiped::RREG : ClsLarge::RREG); st.aovf = ovf > 0 ? (Real*)dalloc(sizeof(Real) * (size_t)N * ovf * kWave) : nullptr; }
        st.obj = nullptr;
        if (cls == 2) {
            st.obj = (Real*)dalloc(sizeof(Real) * (size_t)N * OB_WIDTH);
            if (!st.obj) return fail("device allocation failed");
            rt_memset(st.obj, 0, sizeof(Real) * (size_t)N * OB_WIDTH, stream);
            const double r = c.ball_radius, mass = c.ball_mass;
            md.ball_radius = (Real)r; md.ball_inv_mass = (Real)(1.0 / mass); md.ball_inv_inertia = (Real)(1.0 / (0.4 * mass * r * r));
            md.ball_friction = (Real)c.ball_friction; md.ball_thresh = (Real)(0.02 * r);
            md.ball_ln_lin = log(1.0 - c.ball_lin_damping); md.ball_ln_ang = log(1.0 - c.ball_ang_damping);
            md.obj_time_min = c.rand_tar_obj_time_min; md.obj_time_max = c.rand_tar_obj_time_max; md.min_tar_obj_dist = c.min_tar_obj_dist; md.max_tar_obj_dist = c.max_tar_obj_dist;
        }
        st.manif = nullptr;
        if (physics == 2) {
            st.manif = (Real*)dalloc(sizeof(Real) * (size_t)N * h.J * MF_STRIDE);
            if (!st.manif) return fail("device allocation failed");
            rt_memset(st.manif, 0, sizeof(Real) * (size_t)N * h.J * MF_STRIDE, stream);
        }
        st.hist = nullptr;
        md.scene_amp = c.scene_amp ? 1 : 0; md.amp_local_root = c.enable_amp_obs_local_root ? 1 : 0;
        if (c.scene_amp) {
            // feature layout of one pose block (cSceneImitateAMP::RecordAMPObsPose, SceneImitateAMP.cpp:279-338): root height,
            // root normal / tangent, joint rotations in joint order (spherical 6, revolute 1), end-effector positions
            std::vector<int> off(h.J, 0), ee(h.J, -1);
            int o = 7, ne = 0;
            for (int j = 1; j < h.J; ++j) { off[j] = o; o += (h.jtype[j] == JT_SPHERICAL) ? 6 : ((h.jtype[j] == JT_REVOLUTE) ? 1 : 0); }
            for (int j = 0; j < h.J; ++j) if (h.is_ee[j]) ee[j] = ne++;
            md.amp_ee_base = o; md.amp_pose_size = o + 3 * ne; md.amp_vel_size = h.P - 7 + 6;
            md.amp_off = up<int>(off); md.amp_ee = up<int>(ee);
            amp_size = 2 * (md.amp_pose_size + md.amp_vel_size);
            st.hist = (Real*)dalloc(sizeof(Real) * (size_t)N * 2 * h.P);
            d_amp = (float*)dalloc(sizeof(float) * (size_t)N * amp_size);
            if (!st.hist || !d_amp) return fail("device allocation failed");
        }
        // goal-conditioned task scenes / multi-clip datasets
        st.goal = nullptr;
        md.scene_goal = c.scene_goal; md.enable_min_tar_vel = c.enable_min_tar_vel; md.enable_rand_rot_reset = c.enable_rand_rot_reset;
        md.goal_time_min = c.rand_target_time_min; md.goal_time_max = c.rand_target_time_max;
        md.max_target_dist = (Real)c.max_target_dist; md.target_succ_dist = (Real)c.target_succ_dist; md.tar_fail_dist = (Real)c.tar_fail_dist;
        md.tar_speed = (Real)c.tar_speed; md.pos_reward_scale = (Real)c.pos_reward_scale; md.max_heading_turn_rate = (Real)c.max_heading_turn_rate;
        md.sharp_turn_prob = (Real)c.sharp_turn_prob; md.speed_change_prob = (Real)c.speed_change_prob;
        md.tar_speed_min = (Real)c.tar_speed_min; md.tar_speed_max = (Real)c.tar_speed_max; md.vel_reward_scale = (Real)c.vel_reward_scale;
        md.goal_dim = (c.scene_goal == 3 || c.scene_goal == 4) ? 4 : (c.scene_goal ? 3 : 0); md.mode_test = c.mode_test;
        md.getup_time = c.getup_time; md.recover_prob = c.recover_episode_prob; md.getup_height_root = (Real)c.getup_height_root; md.getup_height_head = (Real)c.getup_height_head;
        md.head_id = c.head_id; md.getup_clip_mask = c.getup_clip_mask;
        md.tar_far_prob = c.tar_far_prob; md.init_hit_prob = c.init_hit_prob; md.hit_reset_time = c.target_hit_reset_time;
        for (int k = 0; k < 3; ++k) { md.target_min[k] = c.target_min[k]; md.target_max[k] = c.target_max[k]; }
        md.tar_near_dist = (Real)c.tar_near_dist; md.target_radius = (Real)c.target_radius; md.hit_tar_speed = (Real)c.hit_tar_speed; md.tar_reward_scale = (Real)c.tar_reward_scale;
        md.strike_mask = c.strike_mask; md.fail_tar_mask = c.fail_tar_mask;
        md.num_clips = h.num_clips;
        md.clip_start = up<int>(h.clip_start); md.clip_dur = up<double>(h.clip_dur); md.clip_loop = up<int>(h.clip_loop);
        md.clip_delta = up<Real>(h.clip_delta); md.clip_cdf = up<double>(h.clip_cdf);
        if (c.scene_goal || h.num_clips > 1 || c.enable_rand_rot_reset) {
            if (!c.scene_amp) return fail("goal scenes, multi-clip datasets and enable_rand_rot_reset ride on the AMP instantiation of the kernels: scene_amp must be set");
            st.goal = (double*)dalloc(sizeof(double) * (size_t)N * GS_WIDTH);
            goal_size = md.goal_dim;
            d_goals = (float*)dalloc(sizeof(float) * (size_t)N * 4);
            if (!st.goal || !d_goals) return fail("device allocation failed");
            rt_memset(st.goal, 0, sizeof(double) * (size_t)N * GS_WIDTH, stream);      // (GS_KON = 0: the ctx's own draw key; every other slot is written by the first reset)
        }
        // random perturbations (scenes/SceneSimChar.cpp:92-99)
        st.pert = nullptr; md.perturb_on = c.enable_rand_perturbs ? 1 : 0;
        if (c.enable_rand_perturbs) {
            md.perturb_time_min = c.perturb_time_min; md.perturb_time_max = c.perturb_time_max; md.perturb_min = c.min_perturb; md.perturb_max = c.max_perturb;
            md.perturb_dur_min = c.min_perturb_duration; md.perturb_dur_max = c.max_perturb_duration; md.perturb_part_mask = (uint32_t)c.perturb_part_mask;
            if (c.perturb_part_mask < 0 || (h.J < 32 && ((uint32_t)c.perturb_part_mask >> h.J) != 0)) return fail("perturb_part_mask names a body part the character does not have");
            if (!(c.perturb_time_max >= c.perturb_time_min) || !(c.perturb_time_min > 0) || !(c.max_perturb >= c.min_perturb) || !(c.max_perturb_duration >= c.min_perturb_duration) || c.min_perturb_duration < 0)
                return fail("perturbation ranges must satisfy 0 < perturb_time_min <= perturb_time_max, min <= max");
            if (PT_SLOTS * c.perturb_time_min < c.max_perturb_duration) return fail("more than 2 perturbations would act at once (needs 2 * perturb_time_min >= max_perturb_duration)");
            st.pert = (double*)dalloc(sizeof(double) * (size_t)N * PT_WIDTH);
            if (!st.pert) return fail("device allocation failed");
            rt_memset(st.pert, 0, sizeof(double) * (size_t)N * PT_WIDTH, stream);
        }
        d_actions = (float*)dalloc(sizeof(float) * N * h.A); d_states = (float*)dalloc(sizeof(float) * N * h.S); d_rewards = (float*)dalloc(sizeof(float) * N);
        d_term = (int*)dalloc(sizeof(int) * N); d_valid = (int*)dalloc(sizeof(int) * N); d_end = (int*)dalloc(sizeof(int) * N);
        if (!st.pose || !st.flag || !d_end || !md.mdl_blob || upload_failed) return fail("device allocation or table upload failed");
        // PD targets start at identity rotations (cPDController::PostProcessTargetPose, PDController.cpp:425-443)
        std::vector<Real> tar((size_t)N * h.P, 0);
        for (int e = 0; e < N; ++e) for (int j = 1; j < h.J; ++j) if (h.jtype[j] == JT_SPHERICAL) tar[(size_t)e * h.P + h.pose_off[j]] = 1;
        if (rt_h2d(st.tar, tar.data(), sizeof(Real) * tar.size(), stream) != 0) return fail("table upload failed");
        // RNG streams are keyed by the global env id (shard offset) so they do not depend on the partition
        md.env_off = env_off;
        return 0;
    }
    int alloc_dbg() {
        if (dbg.H) return 0;
        const HostModel& h = hm;
        dbg.H = (Real*)dalloc(sizeof(Real) * N * h.D * h.D); dbg.C = (Real*)dalloc(sizeof(Real) * N * h.D); dbg.vstar = (Real*)dalloc(sizeof(Real) * N * h.D);
        dbg.lambda = (Real*)dalloc(sizeof(Real) * N * kMaxRows); dbg.rows = (int*)dalloc(sizeof(int) * N * 2);
        dbg.kin_pose = (Real*)dalloc(sizeof(Real) * N * h.P); dbg.kin_vel = (Real*)dalloc(sizeof(Real) * N * h.P);
        dbg.reward_terms = (Real*)dalloc(sizeof(Real) * N * 5); dbg.links = (Real*)dalloc(sizeof(Real) * N * h.J * 21);
        d_prof = (long long*)dalloc(sizeof(long long) * N * 16);
        return dbg.links ? 0 : fail("device allocation failed");
    }
#define DM_DISPATCH(LAUNCH, grid, ...)                                                       \
    do {                                                                                     \
        if (cls == 0) LAUNCH<Real, ClsBiped>(grid, stream, __VA_ARGS__);                         \
        else if (cls == 2) LAUNCH<Real, ClsBipedObj>(grid, stream, __VA_ARGS__);                 \
        else if (cls == 3) LAUNCH<Real, ClsLargeTree>(grid, stream, __VA_ARGS__);                \
        else if (cls == 4) LAUNCH<Real, ClsBipedTree>(grid, stream, __VA_ARGS__);                \
        else LAUNCH<Real, ClsLarge>(grid, stream, __VA_ARGS__);                                  \
    } while (0)

    int reset(const int* ids_dev, int n, const double* kt_dev, const double* mt_dev) override {
        // a bound draw tape serves the whole reset of an env with a goal row; an env without one (single clip, no yaw) has only its perturbation clock on the
        // device -- its clip time and episode limit are the caller's draws (the facade makes them on the host in the reference's order): the counter-based
        // streams must not fill in silently
        if (md.draw_tape && !st.goal && (!kt_dev || !mt_dev)) return fail("a draw tape is bound and the scene has no goal row: dm_reset needs kin_times and max_times (drawn by the caller in the reference's order)");
        DM_DISPATCH(launch_reset, n, md, st, ids_dev, kt_dev, mt_dev);
        return 0;
    }
    int step(const float* actions_dev, double dt, int n_updates, float* states, float* rewards, int* term, int* valid, int* end, int flags, float* amp) override {
        StepIO<Real> io; memset(&io, 0, sizeof(io));
        io.amp_obs = amp; io.goals = d_goals;
        io.actions = actions_dev; io.states = states; io.rewards = rewards; io.terminate = term; io.valid = valid; io.episode_end = end;
        io.env_ids = step_ids; const int GN = step_ids ? step_n_ids : N;      // workgroups of the one-per-wave launches
        if (md.draw_tape && (flags & DM_AUTO_RESET)) return fail("a draw tape is bound (dm_set_draw_tape): resets go through dm_reset, where the tape serves the reference's draw order");
        io.n_updates = n_updates; io.dt = dt; io.auto_reset = (flags & DM_AUTO_RESET) ? 1 : 0; io.emit = (flags & DM_NO_EMIT) ? 0 : 1; io.open_loop = (flags & DM_OPEN_LOOP) ? 1 : 0; io.end_early = (flags & DM_END_EPISODE_EARLY) ? 1 : 0;
        // two characters per wavefront: biped class, even batch, no debug taps armed (DM_DUO=0 keeps one character per wave)
        if (duo && cls == 0 && hm.D == ClsBiped::ND && (N % 2) == 0 && !dbg.H && !step_ids && !md.draw_tape) {      // (31 row lanes per character assume exactly 34 dofs; a bound draw tape is read by the one-per-wave kernels only -- EnvSim::tape() -- so such a batch takes them: ADVICE r5)
            if (st.manif) launch_step_duo<Real, SV_V2>(N / 2, stream, md, st, io, dbg);               // DM-physics v2, two characters per wavefront (round 4)
            else if (st.hist || st.pert || md.enable_root_rot_fail || md.timer_exp > 0) launch_step_duo<Real, SV_AMP>(N / 2, stream, md, st, io, dbg);      // (the AMP instantiation also carries the perturbation code)
            else launch_step_duo<Real, SV_PLAIN>(N / 2, stream, md, st, io, dbg);
            return 0;
        }
        // biped + free body (dribble_amp), two characters per wavefront (round 6; DM-physics v1 -- v2 and armed taps keep the one-per-wave kernel).  DM_DUO_OBJ=0: one per wave
        if (duo && duo_obj && cls == 2 && hm.D == ClsBiped::ND && (N % 2) == 0 && !dbg.H && !step_ids && !md.draw_tape && !st.manif) {
            launch_step_duo_c<Real, ClsBipedObj, SV_AMP>(N / 2, stream, md, st, io, dbg);
            return 0;
        }
        // production launch: the tap-free instantiation unless a parity test armed the debug taps (dm_probe)
        if (cls == 2) {
            if (dbg.H) launch_step<Real, ClsBipedObj, SV_TAPS>(GN, stream, md, st, io, dbg);
            else if (st.manif) launch_step<Real, ClsBipedObj, SV_V2>(GN, stream, md, st, io, dbg);      // round 5: the links' ground contacts through the persistent manifolds, the ball's single-point contacts as under v1
            else launch_step<Real, ClsBipedObj, SV_AMP>(GN, stream, md, st, io, dbg);
        }
        else if (cls == 4) {
            if (dbg.H) launch_step<Real, ClsBipedTree, SV_TAPS>(GN, stream, md, st, io, dbg);
            else if (st.manif) launch_step<Real, ClsBipedTree, SV_V2>(GN, stream, md, st, io, dbg);
            else if (st.hist || st.pert || md.enable_root_rot_fail || md.timer_exp > 0) launch_step<Real, ClsBipedTree, SV_AMP>(GN, stream, md, st, io, dbg);
            else launch_step<Real, ClsBipedTree, SV_PLAIN>(GN, stream, md, st, io, dbg);
        }
        else if (cls == 3) {
            if (dbg.H) launch_step<Real, ClsLargeTree, SV_TAPS>(GN, stream, md, st, io, dbg);
            else if (st.manif) launch_step<Real, ClsLargeTree, SV_V2>(GN, stream, md, st, io, dbg);
            else if (st.hist || st.pert || md.enable_root_rot_fail || md.timer_exp > 0) launch_step<Real, ClsLargeTree, SV_AMP>(GN, stream, md, st, io, dbg);
            else launch_step<Real, ClsLargeTree, SV_PLAIN>(GN, stream, md, st, io, dbg);
        }
        else if (dbg.H) { if (cls == 0) launch_step<Real, ClsBiped, SV_TAPS>(GN, stream, md, st, io, dbg); else launch_step<Real, ClsLarge, SV_TAPS>(GN, stream, md, st, io, dbg); }
        else if (st.manif) { if (cls == 0) launch_step<Real, ClsBiped, SV_V2>(GN, stream, md, st, io, dbg); else launch_step<Real, ClsLarge, SV_V2>(GN, stream, md, st, io, dbg); }      // DM-physics v2: its own instantiation (AMP code + manifolds)
        else if (st.hist || st.pert || md.enable_root_rot_fail || md.timer_exp > 0) { if (cls == 0) launch_step<Real, ClsBiped, SV_AMP>(GN, stream, md, st, io, dbg); else launch_step<Real, ClsLarge, SV_AMP>(GN, stream, md, st, io, dbg); }
        else { if (cls == 0) launch_step<Real, ClsBiped, SV_PLAIN>(GN, stream, md, st, io, dbg); else launch_step<Real, ClsLarge, SV_PLAIN>(GN, stream, md, st, io, dbg); }
        return 0;
    }
    int amp_expert(int n, const double* times_dev, const double* gh_dev, float* out_dev) override { return amp_expert_clips(n, nullptr, times_dev, gh_dev, out_dev); }
    int amp_expert_clips(int n, const int* clips_dev, const double* times_dev, const double* gh_dev, float* out_dev) override {
        if (cls == 0 || cls == 2 || cls == 4) launch_amp_expert<Real, ClsBiped>(n, stream, md, times_dev, gh_dev, out_dev, clips_dev);
        else launch_amp_expert<Real, ClsLarge>(n, stream, md, times_dev, gh_dev, out_dev, clips_dev);
        return 0;
    }
    int get_goal(double* out) override {
        if (!st.goal) return fail("no goal state: not a goal scene / multi-clip dataset");
        std::vector<double> g((size_t)N * GS_WIDTH);
        if (rt_d2h(g.data(), st.goal, sizeof(double) * g.size(), stream) != 0) return fail("device to host copy failed");
        for (int e = 0; e < N; ++e) for (int k = 0; k < 12; ++k) out[(size_t)e * 12 + k] = g[(size_t)e * GS_WIDTH + k];
        return 0;
    }
    int set_goal(const double* in) override {
        if (!st.goal) return fail("no goal state: not a goal scene / multi-clip dataset");
        std::vector<double> g((size_t)N * GS_WIDTH);
        if (rt_d2h(g.data(), st.goal, sizeof(double) * g.size(), stream) != 0) return fail("device to host copy failed");
        for (int e = 0; e < N; ++e) for (int k = 0; k < 12; ++k) g[(size_t)e * GS_WIDTH + k] = in[(size_t)e * 12 + k];
        return rt_h2d(st.goal, g.data(), sizeof(double) * g.size(), stream) == 0 ? 0 : fail("host to device copy failed");
    }
    int goal_aux(double* out, const double* in) override {
        if (!st.goal) return fail("no goal state: not a goal scene / multi-clip dataset");
        std::vector<double> g((size_t)N * GS_WIDTH);
        if (rt_d2h(g.data(), st.goal, sizeof(double) * g.size(), stream) != 0) return fail("device to host copy failed");
        // the 7 scene-specific columns GS_AUX0 .. GS_OTIMER_MAX; column 7 of the 8-wide interface row is reserved (reads 0, writes are ignored): the goal row's next
        // slot is the env's own draw key (GS_KSEED, dm_set_env_keys), which a restored aux block must not re-key
        static_assert(GS_AUX0 + 7 == GS_KSEED, "the aux window ends where the draw key starts");
        if (out) for (int e = 0; e < N; ++e) for (int k = 0; k < 8; ++k) out[(size_t)e * 8 + k] = (k < 7) ? g[(size_t)e * GS_WIDTH + GS_AUX0 + k] : 0.0;
        if (in) {
            for (int e = 0; e < N; ++e) for (int k = 0; k < 7; ++k) g[(size_t)e * GS_WIDTH + GS_AUX0 + k] = in[(size_t)e * 8 + k];
            if (rt_h2d(st.goal, g.data(), sizeof(double) * g.size(), stream) != 0) return fail("host to device copy failed");
        }
        return 0;
    }
    void set_mode(int test) override { md.mode_test = test ? 1 : 0; }
    int obj_state(double* out, const double* in) override {          // N x 13: pos, rot wxyz, vel, ang vel of the free body
        if (!st.obj) return fail("no free body: not a dribble_amp scene");
        std::vector<Real> o((size_t)N * OB_WIDTH);
        if (rt_d2h(o.data(), st.obj, sizeof(Real) * o.size(), stream) != 0) return fail("device to host copy failed");
        if (out) for (int e = 0; e < N; ++e) for (int k = 0; k < 13; ++k) out[(size_t)e * 13 + k] = (double)o[(size_t)e * OB_WIDTH + k];
        if (in) {
            for (int e = 0; e < N; ++e) for (int k = 0; k < 13; ++k) o[(size_t)e * OB_WIDTH + k] = (Real)in[(size_t)e * 13 + k];
            if (rt_h2d(st.obj, o.data(), sizeof(Real) * o.size(), stream) != 0) return fail("host to device copy failed");
        }
        return 0;
    }
    // dm_set_env_keys: the listed envs become what env 0 of a fresh one-env context created with seeds[i] would be BEFORE its first reset: draw key (seed, env 0)
    // in the goal / perturbation rows, draw counters and the episode counter back to 0, no manifolds.  The caller resets them next (dm_create does the same).
    int set_env_keys(const int* ids, int n, const uint64_t* seeds) override {
        if (rt_sync(stream) != 0) return fail("stream synchronize failed");
        std::vector<double> g, p; std::vector<int> f((size_t)N * 4);
        if (st.goal) { g.resize((size_t)N * GS_WIDTH); if (rt_d2h(g.data(), st.goal, sizeof(double) * g.size(), stream) != 0) return fail("device to host copy failed"); }
        if (st.pert) { p.resize((size_t)N * PT_WIDTH); if (rt_d2h(p.data(), st.pert, sizeof(double) * p.size(), stream) != 0) return fail("device to host copy failed"); }
        if (rt_d2h(f.data(), st.flag, sizeof(int) * f.size(), stream) != 0) return fail("device to host copy failed");
        for (int i = 0; i < n; ++i) {
            const int e = ids[i];
            if (e < 0 || e >= N) return fail("dm_set_env_keys: env id out of range");
            if (seeds[i] >= (1ull << 53)) return fail("dm_set_env_keys: seeds must be below 2^53 (they ride in a double of the env's goal / perturbation row)");
            if (st.goal) { double* r = &g[(size_t)e * GS_WIDTH]; for (int k = 0; k < GS_WIDTH; ++k) r[k] = 0; r[GS_KSEED] = (double)seeds[i]; r[GS_KON] = 1.0; }
            if (st.pert) { double* r = &p[(size_t)e * PT_WIDTH]; for (int k = 0; k < PT_WIDTH; ++k) r[k] = 0; r[PT_KSEED] = (double)seeds[i] + 1.0; }
            f[(size_t)e * 4 + 2] = 0;
        }
        if (st.goal && rt_h2d(st.goal, g.data(), sizeof(double) * g.size(), stream) != 0) return fail("host to device copy failed");
        if (st.pert && rt_h2d(st.pert, p.data(), sizeof(double) * p.size(), stream) != 0) return fail("host to device copy failed");
        if (rt_h2d(st.flag, f.data(), sizeof(int) * f.size(), stream) != 0) return fail("host to device copy failed");
        if (st.manif) {
            std::vector<Real> mz((size_t)hm.J * MF_STRIDE, (Real)0);
            for (int i = 0; i < n; ++i) if (rt_h2d(st.manif + (size_t)ids[i] * hm.J * MF_STRIDE, mz.data(), sizeof(Real) * mz.size(), stream) != 0) return fail("host to device copy failed");
        }
        return rt_sync(stream) == 0 ? 0 : fail("stream synchronize failed");
    }
    // dm_set_draw_tape / dm_get_draw_tape_state: N x TP_STRIDE doubles in, N x TP_HDR doubles out
    int draw_tape(const double* in, int unbind, double* hdr_out) override {
        if (unbind) { md.draw_tape = nullptr; return 0; }
        if (in) {
            if (!d_tape) { d_tape = (double*)dalloc(sizeof(double) * (size_t)N * TP_STRIDE); if (!d_tape) return fail("device allocation failed"); }
            if (rt_h2d(d_tape, in, sizeof(double) * (size_t)N * TP_STRIDE, stream) != 0) return fail("host to device copy failed");
            md.draw_tape = d_tape;
        }
        if (hdr_out) {
            if (!d_tape || !md.draw_tape) return fail("no draw tape is bound");
            for (int e = 0; e < N; ++e)
                if (rt_d2h_async(hdr_out + (size_t)e * TP_HDR, d_tape + (size_t)e * TP_STRIDE, sizeof(double) * TP_HDR, stream) != 0) return fail("device to host copy failed");
            if (rt_sync(stream) != 0) return fail("stream synchronize failed");
        }
        return 0;
    }
    // dm_set_draw_tape_envs / dm_get_draw_tape_state_envs: the rows of the listed envs only (the shared-owner route: W workers, each with generators of its own)
    int draw_tape_envs(const int* ids, int n, const double* rows, double* hdr_out) override {
        for (int i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= N) return fail("env id out of range");
        if (rows) {
            if (!d_tape) { d_tape = (double*)dalloc(sizeof(double) * (size_t)N * TP_STRIDE); if (!d_tape) return fail("device allocation failed"); }
            for (int i = 0; i < n; ++i)
                if (rt_h2d_async(d_tape + (size_t)ids[i] * TP_STRIDE, rows + (size_t)i * TP_STRIDE, sizeof(double) * TP_STRIDE, stream) != 0) return fail("host to device copy failed");
            if (rt_sync(stream) != 0) return fail("stream synchronize failed");
            md.draw_tape = d_tape;
        }
        else if (!hdr_out) {                   // n rows with rows == null: bind what the device already holds (rows nobody has drawn from since they went up)
            if (!d_tape) return fail("dm_set_draw_tape_envs: no tape rows on the device yet");
            md.draw_tape = d_tape;
        }
        if (hdr_out) {
            if (!d_tape || !md.draw_tape) return fail("no draw tape is bound");
            if (n > 4) {                          // one strided copy of every header instead of n small ones
                std::vector<double> all((size_t)N * TP_HDR);
                if (rt_d2h_rows(all.data(), sizeof(double) * TP_HDR, d_tape, sizeof(double) * TP_STRIDE, sizeof(double) * TP_HDR, (size_t)N, stream) != 0) return fail("device to host copy failed");
                for (int i = 0; i < n; ++i) memcpy(hdr_out + (size_t)i * TP_HDR, &all[(size_t)ids[i] * TP_HDR], sizeof(double) * TP_HDR);
            } else {
                for (int i = 0; i < n; ++i)
                    if (rt_d2h_async(hdr_out + (size_t)i * TP_HDR, d_tape + (size_t)ids[i] * TP_STRIDE, sizeof(double) * TP_HDR, stream) != 0) return fail("device to host copy failed");
                if (rt_sync(stream) != 0) return fail("stream synchronize failed");
            }
        }
        return 0;
    }
    int pert_state(double* out, const double* in) override {         // N x PT_WIDTH
        if (!st.pert) return fail("no perturbation state: enable_rand_perturbs is off");
        const size_t bytes = sizeof(double) * (size_t)N * PT_WIDTH;
        if (out && rt_d2h(out, st.pert, bytes, stream) != 0) return fail("device to host copy failed");
        if (in && rt_h2d(st.pert, in, bytes, stream) != 0) return fail("host to device copy failed");
        return 0;
    }
    int manifolds(double* out, const double* in) override {           // N x J x 25: count, 4 x (body-frame point (3), plane point x z, distance)
        if (!st.manif) return fail("no manifold state: the ctx does not run DM-physics v2");
        const size_t n = (size_t)N * hm.J;
        std::vector<Real> buf(n * MF_STRIDE);
        if (out) {
            if (rt_d2h(buf.data(), st.manif, sizeof(Real) * buf.size(), stream) != 0) return fail("device to host copy failed");
            for (size_t i = 0; i < n; ++i) for (int k = 0; k < 25; ++k) out[i * 25 + k] = (double)buf[i * MF_STRIDE + k];
        }
        if (in) {
            std::fill(buf.begin(), buf.end(), (Real)0);
            for (size_t i = 0; i < n; ++i) for (int k = 0; k < 25; ++k) buf[i * MF_STRIDE + k] = (Real)in[i * 25 + k];
            if (rt_h2d(st.manif, buf.data(), sizeof(Real) * buf.size(), stream) != 0) return fail("host to device copy failed");
        }
        return 0;
    }
    int set_clips(const int* in) override {
        if (!st.goal) return fail("no goal state: not a goal scene / multi-clip dataset");
        std::vector<double> g((size_t)N * GS_WIDTH);
        if (rt_d2h(g.data(), st.goal, sizeof(double) * g.size(), stream) != 0) return fail("device to host copy failed");
        for (int e = 0; e < N; ++e) { if (in[e] < 0 || in[e] >= hm.num_clips) return fail("clip id out of range"); g[(size_t)e * GS_WIDTH + GS_CLIP] = (double)in[e]; }
        return rt_h2d(st.goal, g.data(), sizeof(double) * g.size(), stream) == 0 ? 0 : fail("host to device copy failed");
    }
    int get_clips(int* out) override {
        if (!st.goal) { for (int e = 0; e < N; ++e) out[e] = 0; return 0; }
        std::vector<double> g((size_t)N * GS_WIDTH);
        if (rt_d2h(g.data(), st.goal, sizeof(double) * g.size(), stream) != 0) return fail("device to host copy failed");
        for (int e = 0; e < N; ++e) out[e] = (int)g[(size_t)e * GS_WIDTH + GS_CLIP];
        return 0;
    }
    int query(float* states, float* rewards, int* term, int* valid, int* end, float* amp) override {
        StepIO<Real> io; memset(&io, 0, sizeof(io));
        io.amp_obs = amp; io.goals = d_goals;
        io.states = states; io.rewards = rewards; io.terminate = term; io.valid = valid; io.episode_end = end; io.emit = 1;
        DM_DISPATCH(launch_query, N, md, st, io, dbg);
        return 0;
    }
    int probe(int what, double dt) override {
        if (alloc_dbg() != 0) return -1;
        if (what == 3 || what == 4) {   // one profiled control step (20 updates; 3: open-loop, 4: the actions of the last dm_step_batch): per-phase cycle counts -> tap "prof"
            rt_memset(d_prof, 0, sizeof(long long) * N * 16, stream);
            DebugTaps<Real> d2; memset(&d2, 0, sizeof(d2)); d2.prof = d_prof;
            StepIO<Real> io; memset(&io, 0, sizeof(io));
            io.states = d_states; io.rewards = d_rewards; io.terminate = d_term; io.valid = d_valid; io.episode_end = d_end;
            io.n_updates = 20; io.dt = dt; io.auto_reset = 1; io.emit = 1; io.open_loop = (what == 3) ? 1 : 0; io.end_early = 1;
            if (what == 4) io.actions = d_actions;
            if (duo && cls == 0 && hm.D == ClsBiped::ND && (N % 2) == 0 && !st.manif) launch_step_duo<Real, SV_TAPS>(N / 2, stream, md, st, io, d2);
            else if (cls == 0) launch_step<Real, ClsBiped, SV_TAPS>(N, stream, md, st, io, d2);
            else if (cls == 2) launch_step<Real, ClsBipedObj, SV_TAPS>(N, stream, md, st, io, d2);
            else if (cls == 3) launch_step<Real, ClsLargeTree, SV_TAPS>(N, stream, md, st, io, d2);
            else if (cls == 4) launch_step<Real, ClsBipedTree, SV_TAPS>(N, stream, md, st, io, d2);
            else launch_step<Real, ClsLarge, SV_TAPS>(N, stream, md, st, io, d2);
            return 0;
        }
        DM_DISPATCH(launch_probe, N, md, st, dbg, what, dt);
        return 0;
    }
    template <typename T> int dl(const T* dev, size_t n, double* out) {
        std::vector<T> tmp(n); if (rt_d2h(tmp.data(), dev, sizeof(T) * n, stream) != 0) return fail("device to host copy failed");
        for (size_t i = 0; i < n; ++i) out[i] = (double)tmp[i];
        return 0;
    }
    template <typename T> int ul(T* dev, size_t n, const double* in) {
        std::vector<T> tmp(n); for (size_t i = 0; i < n; ++i) tmp[i] = (T)in[i];
        return rt_h2d(dev, tmp.data(), sizeof(T) * n, stream) == 0 ? 0 : fail("host to device copy failed");
    }
    int get_state(double* pose, double* vel, double* tar, double* kin, double* clk, int* flg) override {
        const size_t n = N; const int P = hm.P;
        if (pose && dl(st.pose, n * P, pose)) return -1;
        if (vel && dl(st.vel, n * P, vel)) return -1;
        if (tar && dl(st.tar, n * P, tar)) return -1;
        if (kin) { std::vector<double> t8(n * 8); if (dl(st.kin, n * 8, t8.data())) return -1; for (size_t e = 0; e < n; ++e) for (int k = 0; k < 7; ++k) kin[e * 7 + k] = t8[e * 8 + k]; }
        if (clk) { std::vector<double> t6(n * 6); if (rt_d2h(t6.data(), st.clock, sizeof(double) * n * 6, stream)) return fail("copy failed"); for (size_t e = 0; e < n; ++e) for (int k = 0; k < 5; ++k) clk[e * 5 + k] = t6[e * 6 + k]; }
        if (flg && rt_d2h(flg, st.flag, sizeof(int) * n * 4, stream)) return fail("copy failed");
        return 0;
    }
    int set_state(const double* pose, const double* vel, const double* tar, const double* kin, const double* clk, const int* flg) override {
        const size_t n = N; const int P = hm.P;
        if (pose && ul(st.pose, n * P, pose)) return -1;
        if (pose && st.manif && rt_memset(st.manif, 0, sizeof(Real) * n * hm.J * MF_STRIDE, stream)) return fail("memset failed");      // a new pose starts with empty manifolds (physics 2)
        if (vel && ul(st.vel, n * P, vel)) return -1;
        if (tar && ul(st.tar, n * P, tar)) return -1;
        if (kin) { std::vector<double> t8(n * 8, 0); for (size_t e = 0; e < n; ++e) for (int k = 0; k < 7; ++k) t8[e * 8 + k] = kin[e * 7 + k]; if (ul(st.kin, n * 8, t8.data())) return -1; }
        if (clk) { std::vector<double> t6(n * 6, 0); for (size_t e = 0; e < n; ++e) for (int k = 0; k < 5; ++k) t6[e * 6 + k] = clk[e * 5 + k]; if (rt_h2d(st.clock, t6.data(), sizeof(double) * n * 6, stream)) return fail("copy failed"); }
        if (flg && rt_h2d(st.flag, flg, sizeof(int) * n * 4, stream)) return fail("copy failed");
        return 0;
    }
    int set_tau(const double* tau) override { return ul(st.tau, (size_t)N * hm.D, tau); }
    void set_time_limits(double lo, double hi) override { md.time_lim_min = lo; md.time_lim_max = hi; }
    void set_timer_exp(double ex) override { md.timer_exp = ex > 0 ? ex : 0; }
    int get_debug(const char* name, double* out) override {
        const size_t n = N; const HostModel& h = hm; std::string s(name);
        if (s == "tau") return dl(st.tau, n * h.D, out);
        if (s == "borrowed") {      // N: substeps the env's pair ran with one character's rows 32.. on lanes borrowed from its partner's half (two-per-wave kernel, DuoSim::substep_rows_xd)
            std::vector<double> t6(n * 6); if (rt_d2h(t6.data(), st.clock, sizeof(double) * n * 6, stream)) return fail("copy failed");
            for (size_t e = 0; e < n; ++e) out[e] = t6[e * 6 + 5];
            return 0;
        }
        if (s == "fallback") {      // N: substeps the env's pair ran on the 64-lane fallback of the two-per-wave kernel since dm_create / the last dm_set_state (0 for the one-per-wave kernels)
            std::vector<double> t8(n * 8); if (dl(st.kin, n * 8, t8.data())) return -1;
            for (size_t e = 0; e < n; ++e) out[e] = t8[e * 8 + 7];
            return 0;
        }
        if (!dbg.H) return fail("no debug taps recorded yet (call dm_probe first)");
        if (s == "H") return dl(dbg.H, n * h.D * h.D, out);
        if (s == "C") return dl(dbg.C, n * h.D, out);
        if (s == "vstar") return dl(dbg.vstar, n * h.D, out);
        if (s == "lambda") return dl(dbg.lambda, n * kMaxRows, out);
        if (s == "rows") return dl(dbg.rows, n * 2, out);
        if (s == "kin_pose") return dl(dbg.kin_pose, n * h.P, out);
        if (s == "kin_vel") return dl(dbg.kin_vel, n * h.P, out);
        if (s == "reward_terms") return dl(dbg.reward_terms, n * 5, out);
        if (s == "links") return dl(dbg.links, n * h.J * 21, out);
        if (s == "prof") return dl(d_prof, n * 16, out);
        return fail("unknown debug tap: " + s);
    }
};

struct dm_ctx { CtxBase* c; };

// ---------------------------------------------------------------- C-ABI
extern "C" {

static_assert(DM_TAPE_K == dmk::TP_K && DM_TAPE_HDR == dmk::TP_HDR && DM_TAPE_STRIDE == dmk::TP_STRIDE && dmk::TP_TPIN == 9 && dmk::TP_CLIPDRAW == 5, "include/dm_hip.h DM_TAPE_* and dm_types.h TP_* describe one layout");
int dm_abi_version(void) { return DM_ABI_VERSION; }
int dm_struct_sizes(int32_t* out) {
    if (!out) return fail("null argument");
    out[0] = (int32_t)sizeof(dm_create_info); out[1] = (int32_t)sizeof(dm_scene_tables);
    return 0;
}
int dm_is_emulator(void) {
#ifdef DM_EMU
    return 1;
#else
    return 0;
#endif
}
const char* dm_last_error(void) { return g_err.c_str(); }

int dm_destroy(dm_ctx* ctx);
int dm_create(const dm_create_info* info, const dm_scene_tables* tables, dm_ctx** out) {
    if (!info || !tables || !out) return fail("null argument");
    if (info->num_envs < 1) return fail("num_envs must be >= 1");
    if (tables->scene_goal < 0 || tables->scene_goal > 5) return fail("scene_goal must be 0 (none), 1 (target_amp), 2 (heading_amp), 3 (heading_amp_getup), 4 (strike_amp) or 5 (dribble_amp)");
    if (tables->scene_goal == 5 && !(tables->ball_radius > 0 && tables->ball_mass > 0)) return fail("dribble_amp needs ball_radius > 0 and ball_mass > 0");
    if (tables->scene_goal == 5 && info->wave_packing == 2 && info->physics == 2) return fail("dribble_amp under DM-physics v2 runs one character per wavefront (wave_packing 0 or 1)");
    if (tables->scene_goal == 3 && !(tables->getup_time > 0)) return fail("heading_amp_getup needs getup_time > 0 (the longest get-up clip)");
    if (tables->scene_goal == 3 && (tables->head_id < 0 || tables->head_id >= tables->num_joints)) return fail("head_id out of range");
    if (tables->scene_goal == 4 && tables->strike_mask == 0) return fail("strike_amp needs at least one strike body");
    if (tables->scene_goal == 4 && tables->num_joints < 31 && (((unsigned)tables->strike_mask | (unsigned)tables->fail_tar_mask) >> tables->num_joints) != 0) return fail("strike_amp: strike / fail body id out of range");
    if (tables->scene_goal == 3 && tables->num_clips < 31 && ((unsigned)tables->getup_clip_mask >> (tables->num_clips > 0 ? tables->num_clips : 1)) != 0) return fail("heading_amp_getup: getup_motion_ids out of range");
    if (tables->num_sim_substeps < 1) return fail("num_sim_substeps must be >= 1");
    int precision = info->precision ? info->precision : 32;
    if (precision != 32 && precision != 64) return fail("precision must be 32 or 64");
    if (info->physics < 0 || info->physics > 2) return fail("physics must be 0 / 1 (DM-physics v1) or 2 (v2)");

    int mc = info->max_contacts > 0 ? info->max_contacts : 20;
    if (mc > 20) return fail("max_contacts must be <= 20");
#ifndef DM_EMU
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device available: libdm_hip.so has no CPU fallback");
    if (info->device_id < 0 || info->device_id >= ndev) return fail("invalid device_id");
#endif
    DevGuard guard(info->device_id);
    CtxBase* c = (precision == 64) ? (CtxBase*)new CtxT<double>() : (CtxBase*)new CtxT<float>();
    c->device_id = info->device_id; c->N = info->num_envs; c->seed = info->seed; c->precision = precision; c->max_contacts = mc; c->env_off = info->env_id_offset;
    c->physics = (info->physics == 2) ? 2 : 1;
    if (build_host_model(*tables, mc, c->hm) != 0) { delete c; return -1; }
#ifndef DM_EMU
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail("hipStreamCreate failed"); }
    c->stream = c->own_stream;
#endif
    if (info->wave_packing != 0 && info->wave_packing != 1 && info->wave_packing != 2) { delete c; return fail("wave_packing must be 0, 1 or 2"); }
    // two characters per wavefront is the default for the biped class (step() falls back for odd batches and armed taps)
    { const char* dv = getenv("DM_DUO"); c->duo = info->wave_packing == 2 || (info->wave_packing == 0 && !(dv && dv[0] == '0')); }
    { const char* dv = getenv("DM_DUO_OBJ"); c->duo_obj = !(dv && dv[0] == '0'); }
    // Shard / group invariance by construction: global envs 2b and 2b + 1 share a wavefront, whatever the partition.  A shard that starts at an odd
    // global id would pair (2b + 1, 2b + 2) -- the same physics summed in another order, i.e. trajectories that depend on the partition in the last
    // bits -- so an explicit wave_packing 2 refuses it and the default falls back to one character per wavefront for that ctx.
    if (c->duo && (info->env_id_offset & 1)) {
        if (info->wave_packing == 2) { delete c; return fail("wave_packing 2 needs an even env_id_offset (global envs 2b, 2b+1 share a wavefront in every partition)"); }
        c->duo = false;
    }
    if (c->setup() != 0) { delete c; return -1; }
    dm_ctx* ctx = new dm_ctx(); ctx->c = c;
    if (dm_reset(ctx, nullptr, 0, nullptr, nullptr) != 0) { std::string e = g_err; dm_destroy(ctx); g_err = e; return -1; }
    *out = ctx;
    return 0;
}

int dm_destroy(dm_ctx* ctx) {
    if (!ctx) return 0;
    DevGuard guard(ctx->c->device_id);
#ifndef DM_EMU
    if (ctx->c->own_stream) { (void)hipStreamSynchronize(ctx->c->own_stream); (void)hipStreamDestroy(ctx->c->own_stream); }
#endif
    delete ctx->c; delete ctx; return 0;
}

int dm_dims(const dm_ctx* ctx, int32_t* out) {
    if (!ctx || !out) return fail("null argument");
    const HostModel& h = ctx->c->hm;
    out[0] = h.S; out[1] = ctx->c->goal_size; out[2] = h.A; out[3] = h.P; out[4] = h.J; out[5] = h.D; out[6] = h.F; out[7] = ctx->c->N;
    return 0;
}
int dm_physics_info(const dm_ctx* ctx, int32_t* out) {
    if (!ctx || !out) return fail("null argument");
    out[0] = ctx->c->physics; out[1] = ctx->c->max_contacts;
    return 0;
}
double dm_motion_duration(const dm_ctx* ctx) { return ctx ? ctx->c->hm.duration : 0.0; }

int dm_set_stream(dm_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail("null ctx");
#ifndef DM_EMU
    ctx->c->stream = hip_stream ? (hipStream_t)hip_stream : ctx->c->own_stream;
#endif
    return 0;
}
int dm_set_stream_default(dm_ctx* ctx) {
    if (!ctx) return fail("null ctx");
#ifndef DM_EMU
    ctx->c->stream = (hipStream_t)0;          // the legacy default stream: synchronises with every blocking stream, torch's default included
#endif
    return 0;
}
int dm_get_stream(const dm_ctx* ctx, void** out_own, void** out_current) {
    if (!ctx) return fail("null ctx");
    if (out_own) *out_own = (void*)ctx->c->own_stream;
    if (out_current) *out_current = (void*)ctx->c->stream;
    return 0;
}
int dm_synchronize(dm_ctx* ctx) { if (!ctx) return fail("null ctx"); DevGuard guard(ctx->c->device_id); return rt_sync(ctx->c->stream) == 0 ? 0 : fail("stream synchronize failed"); }

int dm_set_timer_exp(dm_ctx* ctx, double time_lim_exp) {
    if (!ctx) return fail("null ctx");
    if (!(time_lim_exp >= 0)) return fail("time_lim_exp must be >= 0 (0: uniform timer)");
    ctx->c->set_timer_exp(time_lim_exp);
    return 0;
}
int dm_set_time_limits(dm_ctx* ctx, double time_lim_min, double time_lim_max) {
    if (!ctx) return fail("null ctx");
    if (!(time_lim_max >= time_lim_min)) return fail("time_lim_max must be >= time_lim_min");
    ctx->c->set_time_limits(time_lim_min, time_lim_max);
    return 0;
}

int dm_reset(dm_ctx* ctx, const int32_t* env_ids, int n, const double* kin_times, const double* max_times) {
    if (!ctx) return fail("null ctx");
    CtxBase* c = ctx->c;
    DevGuard guard(c->device_id);
    if (!env_ids) n = c->N;
    if (n <= 0) return 0;
    for (int i = 0; env_ids && i < n; ++i) if (env_ids[i] < 0 || env_ids[i] >= c->N) return fail("env id out of range");
    int rc = 0;
    void* tmp[3] = {nullptr, nullptr, nullptr};
    auto stage = [&](int k, const void* host, size_t bytes) {      // a failed allocation must not reach the kernel as "NULL = all envs"
        if (!host || rc) return;
        if (rt_malloc(&tmp[k], bytes) != 0) { tmp[k] = nullptr; rc = fail("device allocation failed"); return; }
        if (rt_h2d(tmp[k], host, bytes, c->stream) != 0) rc = fail("copy failed");
    };
    stage(0, env_ids, sizeof(int) * n); stage(1, kin_times, sizeof(double) * n); stage(2, max_times, sizeof(double) * n);
    if (!rc) rc = launch_status(c->reset((const int*)tmp[0], n, (const double*)tmp[1], (const double*)tmp[2]));
    if (rt_sync(c->stream) != 0 && !rc) rc = fail("stream synchronize failed");
    for (void* p : tmp) if (p) rt_free(p);
    return rc;
}

int dm_set_action(dm_ctx* ctx, const float* actions, int flags) {
    if (!ctx || !actions) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    const float* adev = actions;
    if (!(flags & DM_DEVICE_PTRS)) { if (rt_h2d(c->d_actions, actions, sizeof(float) * c->N * c->hm.A, c->stream)) return fail("copy failed"); adev = c->d_actions; }
    return launch_status(c->step(adev, 0.0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, DM_NO_EMIT));
}

int dm_update(dm_ctx* ctx, double timestep, int n_updates) {
    if (!ctx) return fail("null ctx");
    if (n_updates < 1 || !(timestep > 0)) return 0;   // cImpPDController::UpdateControlForce ignores non-positive steps
    DevGuard guard(ctx->c->device_id);
    return launch_status(ctx->c->step(nullptr, timestep, n_updates, nullptr, nullptr, nullptr, nullptr, nullptr, DM_NO_EMIT));
}

static int copy_out(CtxBase* c, void* host, const void* dev, size_t bytes) { return (host && rt_d2h(host, dev, bytes, c->stream)) ? fail("copy failed") : 0; }

int dm_query(dm_ctx* ctx, float* states, float* rewards, int32_t* terminate, int32_t* valid, int32_t* episode_end, int32_t* need_new_action, int flags) {
    if (!ctx) return fail("null ctx");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (flags & DM_DEVICE_PTRS) { if (launch_status(c->query(states, rewards, terminate, valid, episode_end))) return -1; }
    else {
        if (launch_status(c->query(c->d_states, c->d_rewards, c->d_term, c->d_valid, c->d_end))) return -1;
        if (copy_out(c, states, c->d_states, sizeof(float) * c->N * c->hm.S) || copy_out(c, rewards, c->d_rewards, sizeof(float) * c->N) ||
            copy_out(c, terminate, c->d_term, sizeof(int) * c->N) || copy_out(c, valid, c->d_valid, sizeof(int) * c->N) || copy_out(c, episode_end, c->d_end, sizeof(int) * c->N)) return -1;
    }
    if (need_new_action) {
        std::vector<int> flg((size_t)c->N * 4);
        if (c->get_state(nullptr, nullptr, nullptr, nullptr, nullptr, flg.data())) return -1;
        for (int e = 0; e < c->N; ++e) need_new_action[e] = flg[(size_t)e * 4];
    }
    return 0;
}

int dm_step_batch(dm_ctx* ctx, const float* actions, double timestep, int n_updates, float* states, float* rewards,
                  int32_t* terminate, int32_t* valid, int32_t* episode_end, int flags) {
    if (!ctx) return fail("null ctx");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (flags & DM_DEVICE_PTRS) return launch_status(c->step(actions, timestep, n_updates, states, rewards, terminate, valid, episode_end, flags));
    const float* adev = nullptr;
    if (actions) { if (rt_h2d(c->d_actions, actions, sizeof(float) * c->N * c->hm.A, c->stream)) return fail("copy failed"); adev = c->d_actions; }
    if (launch_status(c->step(adev, timestep, n_updates, c->d_states, c->d_rewards, c->d_term, c->d_valid, c->d_end, flags))) return -1;
    if (copy_out(c, states, c->d_states, sizeof(float) * c->N * c->hm.S) || copy_out(c, rewards, c->d_rewards, sizeof(float) * c->N) ||
        copy_out(c, terminate, c->d_term, sizeof(int) * c->N) || copy_out(c, valid, c->d_valid, sizeof(int) * c->N) || copy_out(c, episode_end, c->d_end, sizeof(int) * c->N)) return -1;
    return 0;
}

int dm_step_envs(dm_ctx* ctx, const int32_t* env_ids, int n, const float* actions, double timestep, int n_updates, float* states, float* rewards,
                 int32_t* terminate, int32_t* valid, int32_t* episode_end, float* amp_obs, double* clocks, int flags) {
    if (!ctx || !env_ids) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    const int N = c->N, S = c->hm.S, A = c->hm.A;
    if (n < 1 || n > N) return fail("dm_step_envs: n must be in [1, num_envs]");
    if (flags & DM_DEVICE_PTRS) return fail("dm_step_envs takes host pointers");
    if (amp_obs && !c->amp_size) return fail("AMP observations need a `--scene imitate_amp` context (dm_scene_tables.scene_amp)");
    std::vector<char> seen((size_t)N, 0);
    for (int i = 0; i < n; ++i) { const int e = env_ids[i]; if (e < 0 || e >= N || seen[e]) return fail("dm_step_envs: env ids must be distinct and in [0, num_envs)"); seen[e] = 1; }
    if (!c->d_ids) { c->d_ids = (int*)c->dalloc(sizeof(int) * (size_t)N); if (!c->d_ids) return fail("device allocation failed"); }
    if (rt_h2d(c->d_ids, env_ids, sizeof(int) * (size_t)n, c->stream)) return fail("copy failed");
    const float* adev = nullptr;
    if (actions) { if (rt_h2d(c->d_actions, actions, sizeof(float) * (size_t)n * A, c->stream)) return fail("copy failed"); adev = c->d_actions; }      // compact: row i = env_ids[i]
    c->step_ids = c->d_ids; c->step_n_ids = n;
    const int rc = launch_status(c->step(adev, timestep, n_updates, c->d_states, c->d_rewards, c->d_term, c->d_valid, c->d_end, flags, amp_obs ? c->d_amp : nullptr));
    c->step_ids = nullptr; c->step_n_ids = 0;
    if (rc) return -1;
    // the kernels write every output at the env's own row: copy the arrays back and hand out the rows asked for
    std::vector<float> hs(states ? (size_t)N * S : 0), hr(rewards ? N : 0), ha(amp_obs ? (size_t)N * c->amp_size : 0); std::vector<int> ht(terminate ? N : 0), hv(valid ? N : 0), he(episode_end ? N : 0);
    if ((states && copy_out(c, hs.data(), c->d_states, sizeof(float) * hs.size())) || (rewards && copy_out(c, hr.data(), c->d_rewards, sizeof(float) * N)) ||
        (terminate && copy_out(c, ht.data(), c->d_term, sizeof(int) * N)) || (valid && copy_out(c, hv.data(), c->d_valid, sizeof(int) * N)) ||
        (episode_end && copy_out(c, he.data(), c->d_end, sizeof(int) * N)) || (amp_obs && copy_out(c, ha.data(), c->d_amp, sizeof(float) * ha.size()))) return -1;
    std::vector<double> hc;
    if (clocks) { hc.resize((size_t)N * 5); if (c->get_state(nullptr, nullptr, nullptr, nullptr, hc.data(), nullptr)) return -1; }
    for (int i = 0; i < n; ++i) {
        const int e = env_ids[i];
        if (states) memcpy(states + (size_t)i * S, hs.data() + (size_t)e * S, sizeof(float) * S);
        if (rewards) rewards[i] = hr[e];
        if (terminate) terminate[i] = ht[e];
        if (valid) valid[i] = hv[e];
        if (episode_end) episode_end[i] = he[e];
        if (amp_obs) memcpy(amp_obs + (size_t)i * c->amp_size, ha.data() + (size_t)e * c->amp_size, sizeof(float) * c->amp_size);
        if (clocks) memcpy(clocks + (size_t)i * 5, hc.data() + (size_t)e * 5, sizeof(double) * 5);
    }
    return 0;
}

int dm_amp_obs_size(const dm_ctx* ctx) { return ctx ? ctx->c->amp_size : 0; }

int dm_query_amp(dm_ctx* ctx, float* amp_obs, int flags) {
    if (!ctx || !amp_obs) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (!c->amp_size) return fail("AMP observations need a `--scene imitate_amp` context (dm_scene_tables.scene_amp)");
    if (flags & DM_DEVICE_PTRS) return launch_status(c->query(nullptr, nullptr, nullptr, nullptr, nullptr, amp_obs));
    if (launch_status(c->query(nullptr, nullptr, nullptr, nullptr, nullptr, c->d_amp))) return -1;
    return copy_out(c, amp_obs, c->d_amp, sizeof(float) * (size_t)c->N * c->amp_size);
}

int dm_step_batch_amp(dm_ctx* ctx, const float* actions, double timestep, int n_updates, float* states, float* rewards,
                      int32_t* terminate, int32_t* valid, int32_t* episode_end, float* amp_obs, int flags) {
    if (!ctx) return fail("null ctx");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (amp_obs && !c->amp_size) return fail("AMP observations need a `--scene imitate_amp` context (dm_scene_tables.scene_amp)");
    if (flags & DM_DEVICE_PTRS) return launch_status(c->step(actions, timestep, n_updates, states, rewards, terminate, valid, episode_end, flags, amp_obs));
    const float* adev = nullptr;
    if (actions) { if (rt_h2d(c->d_actions, actions, sizeof(float) * c->N * c->hm.A, c->stream)) return fail("copy failed"); adev = c->d_actions; }
    if (launch_status(c->step(adev, timestep, n_updates, c->d_states, c->d_rewards, c->d_term, c->d_valid, c->d_end, flags, amp_obs ? c->d_amp : nullptr))) return -1;
    if (copy_out(c, states, c->d_states, sizeof(float) * c->N * c->hm.S) || copy_out(c, rewards, c->d_rewards, sizeof(float) * c->N) ||
        copy_out(c, terminate, c->d_term, sizeof(int) * c->N) || copy_out(c, valid, c->d_valid, sizeof(int) * c->N) || copy_out(c, episode_end, c->d_end, sizeof(int) * c->N) ||
        copy_out(c, amp_obs, c->d_amp, sizeof(float) * (size_t)c->N * c->amp_size)) return -1;
    return 0;
}

int dm_amp_expert(dm_ctx* ctx, int n, const double* times, const double* ground_h, float* out, int flags) {
    if (!ctx || !out) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (!c->amp_size) return fail("AMP observations need a `--scene imitate_amp` context (dm_scene_tables.scene_amp)");
    if (n <= 0) return 0;
    if (flags & DM_DEVICE_PTRS) {
        if (!times) return fail("dm_amp_expert with device pointers needs explicit sample times");
        return launch_status(c->amp_expert(n, times, ground_h, out));
    }
    std::vector<double> t(n);
    if (times) memcpy(t.data(), times, sizeof(double) * n);
    else { for (int i = 0; i < n; ++i) t[i] = c->hm.duration * dm_rand01(c->seed, (uint64_t)c->env_off + 0x414D50ull, c->expert_calls, (uint64_t)i); c->expert_calls++; }
    void *td = nullptr, *gd = nullptr, *od = nullptr; int rc = 0;
    if (rt_malloc(&td, sizeof(double) * n) || rt_malloc(&od, sizeof(float) * (size_t)n * c->amp_size) || (ground_h && rt_malloc(&gd, sizeof(double) * n))) rc = fail("device allocation failed");
    if (!rc && (rt_h2d(td, t.data(), sizeof(double) * n, c->stream) || (ground_h && rt_h2d(gd, ground_h, sizeof(double) * n, c->stream)))) rc = fail("copy failed");
    if (!rc) rc = launch_status(c->amp_expert(n, (const double*)td, (const double*)gd, (float*)od));
    if (!rc && rt_d2h(out, od, sizeof(float) * (size_t)n * c->amp_size, c->stream)) rc = fail("copy failed");
    rt_sync(c->stream);
    if (td) rt_free(td); if (gd) rt_free(gd); if (od) rt_free(od);
    return rc;
}

int dm_amp_expert_clips(dm_ctx* ctx, int n, const int32_t* clips, const double* times, const double* ground_h, float* out) {
    if (!ctx || !out) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (!c->amp_size) return fail("AMP observations need a `--scene imitate_amp` context (dm_scene_tables.scene_amp)");
    if (n <= 0) return 0;
    const HostModel& h = c->hm;
    std::vector<int> cl(n); std::vector<double> t(n);
    for (int i = 0; i < n; ++i) {
        if (clips) { if (clips[i] < 0 || clips[i] >= h.num_clips) return fail("clip id out of range"); cl[i] = clips[i]; }
        else { double u = dm_rand01(c->seed, (uint64_t)c->env_off + 0x434C50ull, c->expert_calls, (uint64_t)i); int k = 0; while (k < h.num_clips - 1 && !(u < h.clip_cdf[k])) ++k; cl[i] = k; }
        t[i] = times ? times[i] : h.clip_dur[cl[i]] * dm_rand01(c->seed, (uint64_t)c->env_off + 0x414D50ull, c->expert_calls, (uint64_t)i);
    }
    c->expert_calls++;
    void *td = nullptr, *gd = nullptr, *od = nullptr, *cd = nullptr; int rc = 0;
    if (rt_malloc(&td, sizeof(double) * n) || rt_malloc(&cd, sizeof(int) * n) || rt_malloc(&od, sizeof(float) * (size_t)n * c->amp_size) || (ground_h && rt_malloc(&gd, sizeof(double) * n))) rc = fail("device allocation failed");
    if (!rc && (rt_h2d(td, t.data(), sizeof(double) * n, c->stream) || rt_h2d(cd, cl.data(), sizeof(int) * n, c->stream) || (ground_h && rt_h2d(gd, ground_h, sizeof(double) * n, c->stream)))) rc = fail("copy failed");
    if (!rc) rc = launch_status(c->amp_expert_clips(n, (const int*)cd, (const double*)td, (const double*)gd, (float*)od));
    if (!rc && rt_d2h(out, od, sizeof(float) * (size_t)n * c->amp_size, c->stream)) rc = fail("copy failed");
    rt_sync(c->stream);
    if (td) rt_free(td); if (gd) rt_free(gd); if (od) rt_free(od); if (cd) rt_free(cd);
    return rc;
}

int dm_query_goal(dm_ctx* ctx, float* goals, int flags) {
    if (!ctx || !goals) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (!c->goal_size) return fail("RecordGoal needs a goal scene (dm_scene_tables.scene_goal)");
    if (launch_status(c->query(nullptr, nullptr, nullptr, nullptr, nullptr))) return -1;
    (void)flags;
    return copy_out(c, goals, c->d_goals, sizeof(float) * (size_t)c->N * c->goal_size);
}
int dm_last_goals(dm_ctx* ctx, float* goals) {
    if (!ctx || !goals) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (!c->goal_size) return fail("RecordGoal needs a goal scene (dm_scene_tables.scene_goal)");
    return copy_out(c, goals, c->d_goals, sizeof(float) * (size_t)c->N * c->goal_size);
}
// the same into a DEVICE buffer, asynchronously on the ctx stream (no host sync: a device-resident learner reads it after the step)
int dm_last_goals_device(dm_ctx* ctx, float* goals_dev) {
    if (!ctx || !goals_dev) return fail("null argument");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    if (!c->goal_size) return fail("RecordGoal needs a goal scene (dm_scene_tables.scene_goal)");
    const size_t bytes = sizeof(float) * (size_t)c->N * c->goal_size;
#ifdef DM_EMU
    memcpy(goals_dev, c->d_goals, bytes);
#else
    HIPCHK(hipMemcpyAsync(goals_dev, c->d_goals, bytes, hipMemcpyDeviceToDevice, c->stream));
#endif
    return 0;
}
int dm_get_goal_state(dm_ctx* ctx, double* out) { if (!ctx || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->get_goal(out); }
int dm_set_goal_state(dm_ctx* ctx, const double* in) { if (!ctx || !in) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->set_goal(in); }
int dm_goal_size(const dm_ctx* ctx) { return ctx ? ctx->c->goal_size : 0; }
int dm_get_goal_aux(dm_ctx* ctx, double* out) { if (!ctx || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->goal_aux(out, nullptr); }
int dm_set_goal_aux(dm_ctx* ctx, const double* in) { if (!ctx || !in) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->goal_aux(nullptr, in); }
int dm_get_perturb_state(dm_ctx* ctx, double* out) { if (!ctx || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->pert_state(out, nullptr); }
int dm_set_perturb_state(dm_ctx* ctx, const double* in) { if (!ctx || !in) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->pert_state(nullptr, in); }
int dm_get_obj_state(dm_ctx* ctx, double* out) { if (!ctx || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->obj_state(out, nullptr); }
int dm_set_obj_state(dm_ctx* ctx, const double* in) { if (!ctx || !in) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->obj_state(nullptr, in); }
int dm_get_manifolds(dm_ctx* ctx, double* out) { if (!ctx || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->manifolds(out, nullptr); }
int dm_set_manifolds(dm_ctx* ctx, const double* in) { if (!ctx || !in) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->manifolds(nullptr, in); }
int dm_set_mode(dm_ctx* ctx, int test_mode) { if (!ctx) return fail("null ctx"); ctx->c->set_mode(test_mode); return 0; }
int dm_set_draw_tape(dm_ctx* ctx, const double* tape) { if (!ctx) return fail("null ctx"); DevGuard guard(ctx->c->device_id); return ctx->c->draw_tape(tape, tape == nullptr, nullptr); }
int dm_set_draw_tape_envs(dm_ctx* ctx, const int32_t* env_ids, int n, const double* rows) {
    if (!ctx) return fail("null ctx");
    if (n > 0 && (!env_ids || !rows)) return fail("null argument");
    DevGuard guard(ctx->c->device_id); return ctx->c->draw_tape_envs(env_ids, n > 0 ? n : 0, n > 0 ? rows : nullptr, nullptr);
}
int dm_get_draw_tape_state_envs(dm_ctx* ctx, const int32_t* env_ids, int n, double* out) {
    if (!ctx || !env_ids || !out) return fail("null argument");
    if (n <= 0) return 0;
    DevGuard guard(ctx->c->device_id); return ctx->c->draw_tape_envs(env_ids, n, nullptr, out);
}
int dm_get_draw_tape_state(dm_ctx* ctx, double* out) { if (!ctx || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->draw_tape(nullptr, 0, out); }
int dm_clip_table(const dm_ctx* ctx, double* durations, double* cdf) {
    if (!ctx) return fail("null ctx");
    const HostModel& h = ctx->c->hm;
    const int nc = h.num_clips > 1 ? h.num_clips : 1;
    for (int k = 0; k < nc; ++k) {
        if (durations) durations[k] = (h.num_clips > 1) ? h.clip_dur[k] : h.duration;
        if (cdf) cdf[k] = (h.num_clips > 1) ? h.clip_cdf[k] : 1.0;
    }
    return 0;
}
int dm_set_env_keys(dm_ctx* ctx, const int32_t* env_ids, int n, const uint64_t* seeds) {
    if (!ctx || !env_ids || !seeds) return fail("null argument");
    if (n <= 0) return 0;
    DevGuard guard(ctx->c->device_id);
    return ctx->c->set_env_keys(env_ids, n, seeds);
}
int dm_set_clips(dm_ctx* ctx, const int32_t* in) { if (!ctx || !in) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->set_clips(in); }
int dm_get_clips(dm_ctx* ctx, int32_t* out) { if (!ctx || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->get_clips(out); }

int dm_build_offsets_scales(const dm_ctx* ctx, double* s_off, double* s_scale, double* a_off, double* a_scale, double* a_min, double* a_max, int32_t* s_norm_groups) {
    if (!ctx) return fail("null ctx");
    const HostModel& h = ctx->c->hm;
    if (s_off) memcpy(s_off, h.s_off.data(), sizeof(double) * h.S);
    if (s_scale) memcpy(s_scale, h.s_scale.data(), sizeof(double) * h.S);
    if (a_off) memcpy(a_off, h.a_off.data(), sizeof(double) * h.A);
    if (a_scale) memcpy(a_scale, h.a_scale.data(), sizeof(double) * h.A);
    if (a_min) memcpy(a_min, h.a_min.data(), sizeof(double) * h.A);
    if (a_max) memcpy(a_max, h.a_max.data(), sizeof(double) * h.A);
    if (s_norm_groups) for (int i = 0; i < h.S; ++i) s_norm_groups[i] = h.s_groups[i];
    return 0;
}

int dm_get_state(dm_ctx* ctx, double* pose, double* vel, double* tar, double* kin, double* clocks, int32_t* flags) {
    if (!ctx) return fail("null ctx");
    DevGuard guard(ctx->c->device_id);
    return ctx->c->get_state(pose, vel, tar, kin, clocks, flags);
}
int dm_set_state(dm_ctx* ctx, const double* pose, const double* vel, const double* tar, const double* kin, const double* clocks, const int32_t* flags) {
    if (!ctx) return fail("null ctx");
    DevGuard guard(ctx->c->device_id);
    return ctx->c->set_state(pose, vel, tar, kin, clocks, flags);
}
int dm_probe(dm_ctx* ctx, int what, double dt) { if (!ctx) return fail("null ctx"); DevGuard guard(ctx->c->device_id); int rc = launch_status(ctx->c->probe(what, dt)); if (rt_sync(ctx->c->stream) != 0 && !rc) rc = fail("stream synchronize failed"); return rc; }
int dm_set_tau(dm_ctx* ctx, const double* tau) { if (!ctx || !tau) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->set_tau(tau); }
int dm_get_debug(dm_ctx* ctx, const char* name, double* out) { if (!ctx || !name || !out) return fail("null argument"); DevGuard guard(ctx->c->device_id); return ctx->c->get_debug(name, out); }

int dm_bench_rollout(dm_ctx* ctx, int warmup, int steps, double timestep, int n_updates, int flags, float* states_dev, float* rewards_dev, double* elapsed_ms) {
    if (!ctx) return fail("null ctx");
    CtxBase* c = ctx->c; DevGuard guard(c->device_id);
    float* sdev = states_dev ? states_dev : c->d_states; float* rdev = rewards_dev ? rewards_dev : c->d_rewards;
    for (int k = 0; k < warmup; ++k) if (c->step(nullptr, timestep, n_updates, sdev, rdev, c->d_term, c->d_valid, c->d_end, flags)) return -1;
#ifndef DM_EMU
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, c->stream));
#endif
    for (int k = 0; k < steps; ++k) if (c->step(nullptr, timestep, n_updates, sdev, rdev, c->d_term, c->d_valid, c->d_end, flags)) return -1;
#ifndef DM_EMU
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (elapsed_ms) *elapsed_ms = ms;
    hipError_t le = hipGetLastError(); if (le != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(le));
#else
    if (elapsed_ms) *elapsed_ms = 0;
#endif
    return 0;
}

// ---------------------------------------------------------------- the reference's random generator (host side, N = 1 drop-in route)
// cRand (util/Rand.h, util/Rand.cpp:6-41, 128-135): one std::default_random_engine feeding <random> distributions that persist across
// calls.  The same standard-library types here, so that a host which replays the reference's call order (the cDeepMimicCore facade:
// SeedRand -> scene constructor -> timers -> CalcRandKinResetTime, deepmimic_amd/compat) draws the reference's numbers bit for bit on a
// platform whose C++ library is the reference build's (libstdc++: minstd_rand0, generate_canonical<double, 53>).  Nothing on the device
// uses it: the batched path keeps its counter-based streams.
struct dm_refrand {
    std::default_random_engine gen;
    std::uniform_real_distribution<double> dbl{0, 1};
    // std::normal_distribution<double> keeps the second deviate of a polar pair for the next call; held here in the open (a fresh distribution object per
    // pair: its first call returns y * mult and saves x * mult, its second call returns the saved value without touching the engine), so that the
    // state can cross the boundary to the device draw tape (dm_refrand_tape / dm_refrand_norm_state)
    int navail = 0; double nsaved = 0;
    std::uniform_int_distribution<int> sint{std::numeric_limits<int>::min() + 1, std::numeric_limits<int>::max()};
    std::uniform_int_distribution<unsigned int> uint{std::numeric_limits<unsigned int>::min(), std::numeric_limits<unsigned int>::max()};
};
namespace {
// raw engine values consumed between two states of one engine (the distributions above take a bounded, data-dependent number)
int refrand_consumed(std::default_random_engine from, const std::default_random_engine& to) { int c = 0; while (!(from == to) && c < (1 << 20)) { from(); ++c; } return c; }
}
int dm_refrand_create(unsigned long seed, dm_refrand** out) {
    if (!out) return fail("null argument");
    dm_refrand* r = new dm_refrand(); r->gen = std::default_random_engine(seed); *out = r; return 0;
}
int dm_refrand_destroy(dm_refrand* r) { delete r; return 0; }
int dm_refrand_seed(dm_refrand* r, unsigned long seed) {            // cRand::Seed (:128-135)
    if (!r) return fail("null generator");
    r->gen.seed(seed); r->dbl.reset(); r->navail = 0; r->sint.reset(); r->uint.reset(); return 0;
}
double dm_refrand_double(dm_refrand* r, double mn, double mx) {     // cRand::RandDouble(min, max) (:30-41): no draw when min == max
    if (mn == mx) return mn;
    double u = r->dbl(r->gen);
    u = mn + (u * (mx - mn));
    return u;
}
double dm_refrand_exp(dm_refrand* r, double lambda) { std::exponential_distribution<double> d(lambda); return d(r->gen); }      // (:43-48)
double dm_refrand_norm(dm_refrand* r, double mean, double stdev) {                                                              // (:50-55)
    double v;
    if (r->navail) { v = r->nsaved; r->navail = 0; }
    else { std::normal_distribution<double> d(0, 1); v = d(r->gen); r->nsaved = d(r->gen); r->navail = 1; }
    v = mean + stdev * v;
    return v;
}
int dm_refrand_int(dm_refrand* r) { return r->sint(r->gen); }                                                                   // (:57-60)
int dm_refrand_int_range(dm_refrand* r, int mn, int mx) {           // cRand::RandInt(min, max) (:62-75)
    if (mn == mx) return mn;
    int delta = mx - mn, v = std::abs(dm_refrand_int(r));
    return mn + v % delta;
}
int dm_refrand_uint(dm_refrand* r) { return (int)r->uint(r->gen); }                                                             // (:77-80: returns int)
int dm_refrand_discard(dm_refrand* r, long n) { if (!r || n < 0) return fail("dm_refrand_discard: bad argument"); r->gen.discard((unsigned long long)n); return 0; }
int dm_refrand_norm_state(dm_refrand* r, int set, int* avail, double* saved) {
    if (!r || !avail || !saved) return fail("null argument");
    if (set) { r->navail = *avail ? 1 : 0; r->nsaved = *saved; } else { *avail = r->navail; *saved = r->nsaved; }
    return 0;
}
int dm_refrand_engine_state(dm_refrand* r, int set, unsigned long* state) {      // minstd_rand0: one integer in [1, 2^31 - 2]
    if (!r || !state) return fail("null argument");
    if (set) { std::stringstream ss; ss << *state; ss >> r->gen; return ss.fail() ? fail("dm_refrand_engine_state: not an engine state") : 0; }
    std::stringstream ss; ss << r->gen; ss >> *state; return 0;
}
// The draw tape of one generator (DM_TAPE_* of include/dm_hip.h): entry k of every table is what the distribution would return if its next call found
// the engine k raw values past the generator's current state -- computed with the <random> types themselves, nothing restated.  The generator is
// not advanced.  u: uniform_real_distribution<double>(0, 1) (2 raw values each); e: exponential_distribution<double>(1) = -log(1 - u) (the
// caller divides by lambda exactly as the library does); n3: {first deviate of a fresh polar pair, the saved second one, raw values consumed};
// i2: {|uniform_int_distribution<int>(INT_MIN + 1, INT_MAX)|, raw values consumed}.  Any table may be null.
int dm_refrand_tape(const dm_refrand* r, int K, double* u, double* e, double* n3, double* i2) {
    if (!r || K < 0) return fail("dm_refrand_tape: bad argument");
    std::default_random_engine g = r->gen;
    for (int k = 0; k < K; ++k, g()) {
        if (u) { std::default_random_engine h = g; std::uniform_real_distribution<double> d(0, 1); u[k] = d(h); }
        if (e) { std::default_random_engine h = g; std::exponential_distribution<double> d(1.0); e[k] = d(h); }
        if (n3) { std::default_random_engine h = g; std::normal_distribution<double> d(0, 1); n3[3 * k] = d(h); n3[3 * k + 2] = (double)refrand_consumed(g, h); n3[3 * k + 1] = d(h); }
        if (i2) { std::default_random_engine h = g; std::uniform_int_distribution<int> d(std::numeric_limits<int>::min() + 1, std::numeric_limits<int>::max()); i2[2 * k] = (double)std::abs(d(h)); i2[2 * k + 1] = (double)refrand_consumed(g, h); }
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- record all-gather over RCCL
#ifndef DM_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
// librccl is resolved at the first multi-GPU call, not at load time: a single-GPU user never needs it, and a process that already
// carries an RCCL (torch) gets that copy back from the loader by soname
int rccl_load() {
    if (g_rccl.lib) return 0;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) return fail(std::string("cannot load librccl: ") + dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) { g_rccl.lib = nullptr; return fail("librccl lacks the nccl* entry points"); }
    return 0;
}
int rccl_fail(const char* what, ncclResult_t r) { return fail(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error")); }
}  // namespace
#endif

struct dm_comm {
    int world = 1, rank = 0, device_id = 0;
#ifndef DM_EMU
    ncclComm_t comm = nullptr; hipStream_t stream = nullptr;
    hipEvent_t ready[4] = {nullptr, nullptr, nullptr, nullptr}, done[4] = {nullptr, nullptr, nullptr, nullptr};
#endif
    bool in_flight[4] = {false, false, false, false};
};

extern "C" {

int dm_comm_unique_id(void* out) {
    if (!out) return fail("null argument");
#ifdef DM_EMU
    memset(out, 0, 128); return 0;
#else
    if (rccl_load() != 0) return -1;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclResult_t r = g_rccl.GetUniqueId((ncclUniqueId*)out);
    return r == ncclSuccess ? 0 : rccl_fail("ncclGetUniqueId", r);
#endif
}

int dm_comm_create(const void* uid, int world, int rank, int device_id, dm_comm** out) {
    if (!out || world < 1 || rank < 0 || rank >= world) return fail("dm_comm_create: bad arguments");
    dm_comm* c = new dm_comm(); c->world = world; c->rank = rank; c->device_id = device_id;
#ifdef DM_EMU
    if (world != 1) { delete c; return fail("the CPU emulator build has no collective: world must be 1"); }
#else
    DevGuard guard(device_id);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail("hipStreamCreate failed"); }
    for (int i = 0; i < 4; ++i)
        if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) != hipSuccess) { dm_comm_destroy(c); return fail("hipEventCreate failed"); }
    if (world > 1 || uid) {            // a unique id with world == 1 runs the one-rank collective through RCCL (tests, profiling)
        if (!uid) { dm_comm_destroy(c); return fail("dm_comm_create: world > 1 needs the unique id of rank 0"); }
        if (rccl_load() != 0) { dm_comm_destroy(c); return -1; }
        ncclUniqueId id; memcpy(&id, uid, sizeof(id));
        ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) { dm_comm_destroy(c); return rccl_fail("ncclCommInitRank", r); }
    }
#endif
    *out = c;
    return 0;
}

int dm_comm_destroy(dm_comm* c) {
    if (!c) return 0;
#ifndef DM_EMU
    DevGuard guard(c->device_id);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    for (int i = 0; i < 4; ++i) { if (c->ready[i]) (void)hipEventDestroy(c->ready[i]); if (c->done[i]) (void)hipEventDestroy(c->done[i]); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
#endif
    delete c;
    return 0;
}

int dm_gather_records(dm_ctx* ctx, dm_comm* c, int slot, const float* send_dev, float* recv_dev, size_t count) {
    if (!ctx || !c || !send_dev || !recv_dev) return fail("null argument");
    if (slot < 0 || slot >= 4) return fail("slot must be in [0, 4)");
#ifdef DM_EMU
    memcpy(recv_dev + (size_t)c->rank * count, send_dev, sizeof(float) * count);
#else
    DevGuard guard(c->device_id);
    HIPCHK(hipEventRecord(c->ready[slot], ctx->c->stream));              // the step that wrote send_dev
    HIPCHK(hipStreamWaitEvent(c->stream, c->ready[slot], 0));
    if (!c->comm) HIPCHK(hipMemcpyAsync(recv_dev, send_dev, sizeof(float) * count, hipMemcpyDeviceToDevice, c->stream));
    else { ncclResult_t r = g_rccl.AllGather(send_dev, recv_dev, count, ncclFloat, c->comm, c->stream); if (r != ncclSuccess) return rccl_fail("ncclAllGather", r); }
    HIPCHK(hipEventRecord(c->done[slot], c->stream));
#endif
    c->in_flight[slot] = true;
    return 0;
}

int dm_gather_wait(dm_ctx* ctx, dm_comm* c, int slot) {
    if (!ctx || !c) return fail("null argument");
    if (slot < 0 || slot >= 4) return fail("slot must be in [0, 4)");
    if (!c->in_flight[slot]) return 0;
#ifndef DM_EMU
    DevGuard guard(c->device_id);
    HIPCHK(hipStreamWaitEvent(ctx->c->stream, c->done[slot], 0));
#endif
    c->in_flight[slot] = false;
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- on-device policy inference
#include "dm_policy_host.h"
#include "dm_scene_load.h"
#include "dm_norm.h"          // (after dm_scene_load.h: <map>; after dm_policy_host.h: dm_policy)
