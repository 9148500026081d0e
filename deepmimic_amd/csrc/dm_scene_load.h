// Native scene loader: arg file + character / controller / motion JSON files -> dm_scene_tables, in C++ (round 4).
//
// What it replaces on the reference's side: cDeepMimicCore::ParseArgs (DeepMimicCore.cpp:25-44: cArgParser::LoadArgs / LoadFile, util/ArgParser.cpp:31-120)
// and the ParseArgs + file loading of the scene classes the path serves -- cScene / cRLSceneSimChar / cSceneSimChar / cSceneImitate(AMP) and the five
// task scenes (scenes/*.cpp ParseArgs), cKinTree::Load (anim/KinTree.cpp:1022-1130: "Skeleton"."Joints", "BodyDefs"), the PD controller file
// (sim/CtPDController.cpp / PDController.cpp:50-93: "PDControllers" Kp / Kd; sim/CtController.cpp:161-172: EnablePhaseInput, RecordWorldRootPos/Rot,
// QueryRate), cMotion::Load (anim/Motion.cpp:302-378: "Loop", "Frames") and cClipsController::LoadMotions (anim/ClipsController.cpp:150-188: "Motions").
// A native host (DeepMimicCore/Main.cpp:38-75) can therefore go from the reference's own files to a running context without Python:
//     dm_scene_load(args, n, data_root, test_mode, &scene); dm_create(&info, dm_scene_get_tables(scene), &ctx);
// It is the C++ twin of deepmimic_amd/model.py (load_scene_from_args) + core.py (fill_scene_tables); tests/test_scene_load.py holds the two against each
// other field by field and array by array on every arg file of the reference.  Included at the end of dm_host.cpp (uses its fail()).
#include <charconv>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace dmscene {

// locale-independent, correctly rounded (like Python's float()): strtod would read "0.5" as 0 under a host's LC_NUMERIC with a decimal comma
static inline bool parse_double(const char* b, const char* e, double& out, const char** end = nullptr) {
    if (b < e && *b == '+') ++b;                    // (arg files may carry an explicit sign; from_chars takes '-' only)
    const auto r = std::from_chars(b, e, out);
    if (end) *end = r.ptr;
    return r.ec == std::errc();
}
static inline double to_double(const std::string& s, double dflt = 0.0) { double d = dflt; return parse_double(s.data(), s.data() + s.size(), d) ? d : dflt; }

// ---- a small JSON reader (objects, arrays, strings, numbers, true / false / null): the reference's data files need nothing else
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false; double num = 0; std::string str;
    std::vector<JVal> arr; std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const std::string& k) const { for (const auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
    bool is_num() const { return kind == NUM; }
};
struct JParser {
    const std::string& s; size_t i = 0; std::string err; int depth = 0;
    explicit JParser(const std::string& s_) : s(s_) {}
    struct Depth { int& d; explicit Depth(int& d_) : d(d_) { ++d; } ~Depth() { --d; } };
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i; }
    bool parse(JVal& v) {
        Depth guard(depth);
        if (depth > 64) { err = "nesting deeper than 64"; return false; }      // (the data files nest 4 deep; a damaged file must not overflow the stack)
        ws();
        if (i >= s.size()) { err = "unexpected end of file"; return false; }
        const char c = s[i];
        if (c == '{') {
            v.kind = JVal::OBJ; ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return true; }
            for (;;) {
                ws(); JVal k;
                if (i >= s.size() || s[i] != '"' || !parse(k)) { if (err.empty()) err = "expected a key"; return false; }
                ws(); if (i >= s.size() || s[i] != ':') { err = "expected ':'"; return false; } ++i;
                JVal x; if (!parse(x)) return false;
                v.obj.emplace_back(k.str, std::move(x));
                ws(); if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return true; }
                err = "expected ',' or '}'"; return false;
            }
        }
        if (c == '[') {
            v.kind = JVal::ARR; ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return true; }
            for (;;) {
                JVal x; if (!parse(x)) return false;
                v.arr.push_back(std::move(x));
                ws(); if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return true; }
                err = "expected ',' or ']'"; return false;
            }
        }
        if (c == '"') {
            v.kind = JVal::STR; ++i;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) { const char e = s[i + 1]; v.str += (e == 'n') ? '\n' : (e == 't') ? '\t' : e; i += 2; }
                else v.str += s[i++];
            }
            if (i >= s.size()) { err = "unterminated string"; return false; }
            ++i; return true;
        }
        if (s.compare(i, 4, "true") == 0) { v.kind = JVal::BOOL; v.b = true; i += 4; return true; }
        if (s.compare(i, 5, "false") == 0) { v.kind = JVal::BOOL; v.b = false; i += 5; return true; }
        if (s.compare(i, 4, "null") == 0) { v.kind = JVal::NUL; i += 4; return true; }
        const char* end = nullptr; double d = 0;
        if (!parse_double(s.data() + i, s.data() + s.size(), d, &end)) { err = std::string("unexpected character '") + c + "'"; return false; }
        v.kind = JVal::NUM; v.num = d; i = (size_t)(end - s.data());
        return true;
    }
};
static bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss; ss << f.rdbuf(); out = ss.str();
    return true;
}
static int load_json(const std::string& path, JVal& v) {
    std::string txt;
    if (!read_file(path, txt)) return fail("cannot open " + path);
    JParser p(txt);
    if (!p.parse(v)) return fail(path + ": JSON error at byte " + std::to_string(p.i) + ": " + p.err);
    return 0;
}

// ---- cArgParser (util/ArgParser.cpp:31-120): "--key v0 v1 ..." ; the first occurrence of a key wins; '#' starts a comment token / line
struct Args {
    std::map<std::string, std::vector<std::string>> table;
    void load(const std::vector<std::string>& toks) {
        std::string key; std::vector<std::string> vals;
        auto flush = [&]() { if (!key.empty() && !table.count(key)) table[key] = vals; };
        for (const std::string& s : toks) {
            if (!s.empty() && s[0] == '#') continue;
            if (s.size() >= 3 && s[0] == '-' && s[1] == '-') { flush(); key = s.substr(2); vals.clear(); }
            else vals.push_back(s);
        }
        flush();
    }
    bool load_file(const std::string& path) {
        std::ifstream f(path);
        if (!f) return false;
        std::vector<std::string> toks; std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::string cur;
            for (char c : line) {
                if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == ',') { if (!cur.empty()) { toks.push_back(cur); cur.clear(); } }
                else cur += c;
            }
            if (!cur.empty()) toks.push_back(cur);
        }
        load(toks);
        return true;
    }
    const std::vector<std::string>* get(const std::string& k) const { auto it = table.find(k); return it == table.end() ? nullptr : &it->second; }
    std::string str(const std::string& k, const std::string& d) const { auto v = get(k); return (v && !v->empty()) ? (*v)[0] : d; }
    double num(const std::string& k, double d) const { auto v = get(k); return (v && !v->empty()) ? to_double((*v)[0], d) : d; }
    int inum(const std::string& k, int d) const { auto v = get(k); return (v && !v->empty()) ? (int)strtol((*v)[0].c_str(), nullptr, 10) : d; }
    bool flag(const std::string& k, bool d) const {           // cArgParser::ParseBool
        auto v = get(k); if (!v || v->empty()) return d;
        const std::string& x = (*v)[0]; return x == "true" || x == "1" || x == "True" || x == "T" || x == "t";
    }
    bool ints(const std::string& k, std::vector<int>& out) const { auto v = get(k); if (!v || v->empty()) return false; out.clear(); for (auto& x : *v) out.push_back((int)strtol(x.c_str(), nullptr, 10)); return true; }
    bool nums(const std::string& k, std::vector<double>& out) const { auto v = get(k); if (!v || v->empty()) return false; out.clear(); for (auto& x : *v) out.push_back(to_double(x)); return true; }
};

static const char* kJointTypes[] = {"revolute", "planar", "prismatic", "fixed", "spherical", "none"};      // anim/KinTree.cpp: gJointTypeNames
static const char* kJointKeys[19] = {"Type", "Parent", "AttachX", "AttachY", "AttachZ", "AttachThetaX", "AttachThetaY", "AttachThetaZ", "LimLow0", "LimLow1", "LimLow2",
                                     "LimHigh0", "LimHigh1", "LimHigh2", "TorqueLim", "ForceLim", "IsEndEffector", "DiffWeight", "Offset"};
static const char* kShapes[] = {"null", "box", "capsule", "sphere", "cylinder", "plane"};                  // anim/Shape.h:8-17
static const char* kBodyKeys[17] = {"Shape", "Mass", "ColGroup", "EnableFallContact", "AttachX", "AttachY", "AttachZ", "AttachThetaX", "AttachThetaY", "AttachThetaZ",
                                    "Param0", "Param1", "Param2", "ColorR", "ColorG", "ColorB", "ColorA"};
static int param_size(int jt, bool root) { if (root) return 7; switch (jt) { case 0: return 1; case 1: return 3; case 2: return 1; case 3: return 0; case 4: return 4; default: return 0; } }

}  // namespace dmscene

struct dm_scene {
    dm_scene_tables t;
    std::vector<double> joint_mat, body_defs, pd_params, frames, clip_weights;
    std::vector<int32_t> fall_mask, clip_starts, clip_loops;
    std::string scene_name; int num_update_substeps = 1;
    double anneal_samples = -1, time_end_lim_min = 0, time_end_lim_max = 0, time_end_lim_exp = 0, time_lim_exp = 1; int timer_exp_type = 0;
};

extern "C" {

int dm_scene_load(const char* const* args, int n_args, const char* data_root, int test_mode, dm_scene** out) {
    using namespace dmscene;
    if (!out || (n_args > 0 && !args)) return fail("null argument");
    const std::string root = (data_root && data_root[0]) ? std::string(data_root) : std::string(".");
    auto res = [&](const std::string& p) { return (!p.empty() && p[0] == '/') ? p : root + "/" + p; };
    Args a;
    { std::vector<std::string> toks; for (int i = 0; i < n_args; ++i) toks.push_back(args[i] ? args[i] : ""); a.load(toks); }
    const std::string arg_file = a.str("arg_file", "");
    if (!arg_file.empty() && !a.load_file(res(arg_file))) return fail("Failed to load args from: " + arg_file);

    std::unique_ptr<dm_scene> sc(new dm_scene());
    dm_scene_tables& t = sc->t; memset(&t, 0, sizeof(t));
    const std::string scene = a.str("scene", "imitate");
    sc->scene_name = scene;
    int goal = 0;
    if (scene == "target_amp") goal = 1; else if (scene == "heading_amp") goal = 2; else if (scene == "heading_amp_getup") goal = 3;
    else if (scene == "strike_amp") goal = 4; else if (scene == "dribble_amp") goal = 5;
    else if (scene != "imitate" && scene != "imitate_amp")
        return fail("only `--scene imitate`, `imitate_amp`, `heading_amp`, `heading_amp_getup`, `target_amp`, `strike_amp` and `dribble_amp` are on the accelerated path (got '" + scene + "')");
    const bool amp = scene != "imitate";

    // ---- files
    JVal cj, kj, mj;
    if (load_json(res(a.str("character_files", "")), cj) || load_json(res(a.str("char_ctrl_files", "")), kj)) return -1;
    const std::string motion_file = res(a.str("motion_file", ""));
    if (load_json(motion_file, mj)) return -1;
    const JVal* skel = cj.get("Skeleton"); const JVal* joints = skel ? skel->get("Joints") : nullptr; const JVal* bdefs = cj.get("BodyDefs");
    if (!joints || joints->kind != JVal::ARR || !bdefs || bdefs->kind != JVal::ARR) return fail("character file: missing Skeleton.Joints / BodyDefs");
    const int J = (int)joints->arr.size();
    if (J < 1 || J > 64) return fail("character file: the skeleton needs 1 .. 64 joints");
    if ((int)bdefs->arr.size() != J) return fail("joint / body-def count mismatch");
    // cKinTree::BuildJointDesc defaults (anim/KinTree.cpp:1132-1156) + the file's keys; PostProcessJointMat (:1005-1020)
    sc->joint_mat.assign((size_t)J * 19, 0.0);
    const double inf = std::numeric_limits<double>::infinity();
    for (int j = 0; j < J; ++j) {
        double* d = &sc->joint_mat[(size_t)j * 19];
        d[0] = 0; d[1] = -1; d[8] = d[9] = d[10] = 1; d[11] = d[12] = d[13] = 0; d[14] = inf; d[15] = inf; d[17] = 1;
        const JVal& jj = joints->arr[j];
        const JVal* ty = jj.get("Type");
        if (!ty || ty->kind != JVal::STR) return fail("joint without a Type");
        int jt = -1; for (int k = 0; k < 6; ++k) if (ty->str == kJointTypes[k]) jt = k;
        if (jt < 0) return fail("unknown joint type " + ty->str);
        d[0] = jt;
        for (int i = 1; i < 19; ++i) { const JVal* v = jj.get(kJointKeys[i]); if (v && v->kind == JVal::NUM) d[i] = v->num; else if (v && v->kind == JVal::BOOL) d[i] = v->b ? 1 : 0; }
    }
    { int off = 0;
      for (int j = 0; j < J; ++j) {
          double* d = &sc->joint_mat[(size_t)j * 19];
          if (!(d[1] >= -1 && d[1] < (double)j) || d[1] != std::floor(d[1])) return fail("parent id must be an integer in [-1, child id)");
          d[18] = off; off += param_size((int)d[0], j == 0);
      }
      sc->joint_mat[2] = sc->joint_mat[3] = sc->joint_mat[4] = 0; }
    sc->body_defs.assign((size_t)J * 17, 0.0);
    for (int b = 0; b < J; ++b) {
        double* d = &sc->body_defs[(size_t)b * 17];
        d[2] = -1; d[16] = 1;                                  // cKinTree::BuildBodyDef (:1158-1179)
        const JVal& bj = bdefs->arr[b];
        const JVal* sh = bj.get("Shape");
        int shape = 0;
        if (sh && sh->kind == JVal::STR) { shape = -1; for (int k = 0; k < 6; ++k) if (sh->str == kShapes[k]) shape = k; if (shape < 0) return fail("unknown shape " + sh->str); }
        d[0] = shape;
        for (int i = 1; i < 17; ++i) { const JVal* v = bj.get(kBodyKeys[i]); if (v && v->kind == JVal::NUM) d[i] = v->num; }
    }
    const JVal* pds = kj.get("PDControllers");
    if (!pds || pds->kind != JVal::ARR || (int)pds->arr.size() != J) return fail("PD controller count mismatch");
    sc->pd_params.assign((size_t)J * 2, 0.0);
    for (int j = 0; j < J; ++j) { const JVal* kp = pds->arr[j].get("Kp"); const JVal* kd = pds->arr[j].get("Kd"); if (kp && kp->is_num()) sc->pd_params[2 * j] = kp->num; if (kd && kd->is_num()) sc->pd_params[2 * j + 1] = kd->num; }
    const int P = (int)sc->joint_mat[(size_t)(J - 1) * 19 + 18] + param_size((int)sc->joint_mat[(size_t)(J - 1) * 19], J == 1);

    auto take_frames = [&](const JVal& m, const std::string& path, int& loop) -> int {
        const JVal* fr = m.get("Frames");
        if (!fr || fr->kind != JVal::ARR || fr->arr.empty()) return fail(path + ": no \"Frames\"");
        const JVal* lp = m.get("Loop"); const std::string ls = (lp && lp->kind == JVal::STR) ? lp->str : "none";
        if (ls != "none" && ls != "wrap") return fail("unsupported loop mode '" + ls + "' in " + path);
        loop = ls == "wrap";
        for (const JVal& row : fr->arr) {
            if (row.kind != JVal::ARR) return fail(path + ": a frame that is not an array");
            if ((int)row.arr.size() != P + 1) return fail("DOF mismatch, char dof " + std::to_string(P) + ", motion dof " + std::to_string((int)row.arr.size() - 1));
            for (const JVal& x : row.arr) { if (!x.is_num()) return fail(path + ": a frame entry that is not a number"); sc->frames.push_back(x.num); }
        }
        return (int)fr->arr.size();
    };
    int loop0 = 0;
    if (mj.get("Frames")) {
        if (take_frames(mj, motion_file, loop0) < 0) return -1;
        t.num_clips = 0;
    } else {
        const JVal* ms = mj.get("Motions");
        if (!ms || ms->kind != JVal::ARR || ms->arr.empty()) return fail(motion_file + " has neither \"Frames\" nor \"Motions\"");
        sc->clip_starts.push_back(0);
        for (const JVal& ent : ms->arr) {                     // cClipsController::LoadMotions: files relative to the data root
            const JVal* fl = ent.get("File"); const JVal* w = ent.get("Weight");
            const std::string path = res((fl && fl->kind == JVal::STR) ? fl->str : "");
            JVal clip; if (load_json(path, clip)) return -1;
            int lp = 0; const int n = take_frames(clip, path, lp);
            if (n < 0) return -1;
            sc->clip_starts.push_back(sc->clip_starts.back() + n); sc->clip_weights.push_back((w && w->is_num()) ? w->num : 1.0); sc->clip_loops.push_back(lp);
        }
        loop0 = sc->clip_loops[0];
        t.num_clips = (int)sc->clip_weights.size();
    }
    const int F = (int)(sc->frames.size() / (size_t)(P + 1));

    // ---- dm_scene_tables (the C++ twin of core.py fill_scene_tables / model.py parse_scene_config)
    t.num_joints = J; t.joint_mat = sc->joint_mat.data(); t.body_defs = sc->body_defs.data(); t.pd_params = sc->pd_params.data();
    t.num_frames = F; t.frames = sc->frames.data(); t.loop = loop0;
    sc->fall_mask.assign(J, 0);
    { std::vector<int> fb;
      if (a.ints("fall_contact_bodies", fb)) { for (int b : fb) { if (b < 0 || b >= J) return fail("fall_contact_bodies out of range"); sc->fall_mask[b] = 1; } }
      else for (int j = 0; j < J; ++j) sc->fall_mask[j] = sc->body_defs[(size_t)j * 17 + 3] != 0 ? 1 : 0; }
    t.fall_mask = sc->fall_mask.data();
    sc->num_update_substeps = a.inum("num_update_substeps", 1);
    t.num_sim_substeps = a.inum("num_sim_substeps", 1); t.world_scale = a.num("world_scale", 1.0);
    t.gravity[0] = 0; t.gravity[1] = -9.8; t.gravity[2] = 0;
    { std::vector<double> g; if (a.nums("gravity", g)) for (int k = 0; k < 3 && k < (int)g.size(); ++k) t.gravity[k] = g[k]; }
    t.sync_char_root_pos = a.flag("sync_char_root_pos", true); t.sync_char_root_rot = a.flag("sync_char_root_rot", false);
    t.enable_fall_end = a.flag("enable_fall_end", true); t.enable_char_contact_fall = a.flag("enable_char_contact_fall", true);
    t.enable_root_rot_fail = a.flag("enable_root_rot_fail", false); t.enable_rand_char_placement = a.flag("enable_rand_char_placement", true);
    t.enable_rand_rot_reset = a.flag("enable_rand_rot_reset", false);
    double tmin = a.num("time_lim_min", inf), tmax = a.num("time_lim_max", inf);
    sc->time_lim_exp = a.num("time_lim_exp", 1.0);
    sc->time_end_lim_min = a.num("time_end_lim_min", tmin); sc->time_end_lim_max = a.num("time_end_lim_max", tmax); sc->time_end_lim_exp = a.num("time_end_lim_exp", sc->time_lim_exp);
    sc->anneal_samples = a.inum("anneal_samples", -1);
    { std::string tt = a.str("timer_type", "uniform"); if (tt.empty()) tt = "uniform";
      if (tt != "uniform" && tt != "exp") return fail("unsupported timer type '" + tt + "' (util/Timer.cpp:27-45: uniform | exp)");
      sc->timer_exp_type = tt == "exp"; }
    if (test_mode) tmin = tmax = sc->time_end_lim_max;        // cRLSceneSimChar::ResetTimers (scenes/RLSceneSimChar.cpp:277-284)
    t.time_lim_min = tmin; t.time_lim_max = tmax;
    auto jbool = [&](const char* k) { const JVal* v = kj.get(k); return (v && ((v->kind == JVal::BOOL && v->b) || (v->kind == JVal::NUM && v->num != 0))) ? 1 : 0; };
    t.enable_phase_input = jbool("EnablePhaseInput"); t.record_world_root_pos = jbool("RecordWorldRootPos"); t.record_world_root_rot = jbool("RecordWorldRootRot");
    { const JVal* q = kj.get("QueryRate"); t.query_rate = (q && q->is_num()) ? q->num : 30.0; }
    t.friction = 0; t.erp = 0; t.solver_iters = 0; t.disable_self_collision = 0;
    t.scene_amp = amp ? 1 : 0; t.enable_amp_obs_local_root = a.flag("enable_amp_obs_local_root", false) && !goal;      // only cSceneImitateAMP::ParseArgs reads the key; the task scenes bypass it (SceneTargetAMP.cpp:103-105 -> cSceneImitate::ParseArgs): false for them whatever the arg file says
    t.scene_goal = goal;
    const bool heading = goal == 2 || goal == 3;
    t.rand_target_time_min = a.num("rand_target_time_min", heading ? 0.2 : (goal == 5 ? 50.0 : 1.0));      // constructor defaults: SceneHeadingAMP.cpp:45-48, SceneDribbleAMP.cpp:124-133
    t.rand_target_time_max = a.num("rand_target_time_max", heading ? 0.5 : (goal == 5 ? 100.0 : 5.0));
    t.max_target_dist = a.num("max_target_dist", 3.0); t.target_succ_dist = a.num("target_succ_dist", 0.5); t.tar_fail_dist = a.num("tar_fail_dist", inf);
    t.tar_speed = a.num("tar_speed", 1.0); t.enable_min_tar_vel = a.flag("enable_min_tar_vel", false);
    t.pos_reward_scale = a.num("pos_reward_scale", 1.0); t.max_heading_turn_rate = a.num("max_heading_turn_rate", 0.15); t.sharp_turn_prob = a.num("sharp_turn_prob", 0.025);
    t.speed_change_prob = a.num("speed_change_prob", 0.1); t.vel_reward_scale = a.num("vel_reward_scale", 1.0);
    t.tar_speed_min = a.num("tar_speed_min", t.tar_speed); t.tar_speed_max = a.num("tar_speed_max", t.tar_speed);
    if (heading) t.tar_speed = std::min(std::max(t.tar_speed, t.tar_speed_min), t.tar_speed_max);          // cSceneHeadingAMP::ParseArgs (:66-84)
    if (t.num_clips) { t.clip_starts = sc->clip_starts.data(); t.clip_weights = sc->clip_weights.data(); t.clip_loops = sc->clip_loops.data(); }
    t.mode_test = test_mode ? 1 : 0;
    { std::vector<int> ids; double gt = 0; int mask = 0;
      if (a.ints("getup_motion_ids", ids)) for (int c : ids) {
          const int nc = t.num_clips ? t.num_clips : 1;
          if (c < 0 || c >= std::min(nc, 31)) return fail("getup_motion_ids: not a clip of the dataset");
          const int f0 = t.num_clips ? sc->clip_starts[c] : 0, f1 = t.num_clips ? sc->clip_starts[c + 1] : F;
          double dur = 0; for (int f = f0; f < f1 - 1; ++f) dur += sc->frames[(size_t)f * (P + 1)];      // cMotion::GetDuration: every frame duration but the last
          gt = std::max(gt, dur); mask |= 1 << c;
      }
      t.getup_time = gt; t.getup_clip_mask = mask; }
    t.getup_height_root = a.num("getup_height_root", 0.5); t.getup_height_head = a.num("getup_height_head", 0.5); t.recover_episode_prob = a.num("recover_episode_prob", 0.0);
    t.head_id = a.inum("head_id", 0);
    t.tar_near_dist = a.num("tar_near_dist", 1.4); t.tar_far_prob = a.num("tar_far_prob", 0.4); t.target_radius = a.num("target_radius", 0.2);
    t.target_hit_reset_time = a.num("target_hit_reset_time", 2.0); t.init_hit_prob = a.num("init_hit_prob", 0.0); t.hit_tar_speed = a.num("hit_tar_speed", 1.5);
    t.tar_reward_scale = a.num("tar_reward_scale", 2.0);
    { const double dmin[3] = {-0.5, 1.2, 0.6}, dmax[3] = {0.5, 1.4, 1.1}; std::vector<double> v;
      for (int k = 0; k < 3; ++k) { t.target_min[k] = dmin[k]; t.target_max[k] = dmax[k]; }
      if (a.nums("target_min", v)) for (int k = 0; k < 3 && k < (int)v.size(); ++k) t.target_min[k] = v[k];
      if (a.nums("target_max", v)) for (int k = 0; k < 3 && k < (int)v.size(); ++k) t.target_max[k] = v[k]; }
    auto mask_of = [&](const char* key, int& outm) -> int {
        std::vector<int> ids; outm = 0;
        if (a.ints(key, ids)) for (int b : ids) { if (b < 0 || b >= std::min(J, 31)) return fail(std::string(key) + " out of range"); outm |= 1 << b; }
        return 0;
    };
    if (mask_of("strike_bodies", t.strike_mask) || mask_of("fail_tar_contact_bodies", t.fail_tar_mask)) return -1;
    t.rand_tar_obj_time_min = a.num("rand_tar_obj_time_min", 100.0); t.rand_tar_obj_time_max = a.num("rand_tar_obj_time_max", 200.0);
    t.min_tar_obj_dist = a.num("min_tar_obj_dist", 0.5); t.max_tar_obj_dist = a.num("max_tar_obj_dist", 10.0); t.ball_radius = a.num("ball_radius", 0.2);
    t.ball_mass = 0.43; t.ball_friction = 0.4 * 0.9; t.ball_lin_damping = 0.4; t.ball_ang_damping = 0.4;      // cSceneDribbleAMP::BuildTarObjs (:398-420); friction combined with the 0.9 of links / ground
    { const double ptmin = a.num("perturb_time_min", inf);
      t.enable_rand_perturbs = (a.flag("enable_rand_perturbs", false) && std::isfinite(ptmin)) ? 1 : 0;
      if (t.enable_rand_perturbs) {
          t.perturb_time_min = ptmin; t.perturb_time_max = a.num("perturb_time_max", inf); t.min_perturb = a.num("min_perturb", 50.0); t.max_perturb = a.num("max_perturb", 100.0);
          t.min_perturb_duration = a.num("min_pertrub_duration", 0.1); t.max_perturb_duration = a.num("max_perturb_duration", 0.5);      // [sic] the reference's key
          std::vector<int> parts; int m = 0;
          if (a.ints("perturb_part_ids", parts)) for (size_t i = 1; i < parts.size(); ++i) if (parts[i] <= parts[i - 1])
              return fail("perturb_part_ids must be ascending and without repeats: the reference indexes the list as written (scenes/SceneSimChar.cpp:244-252), the device draws the idx-th id of the set");
          if (!parts.empty()) for (int b : parts) { if (b < 0 || b >= std::min(J, 31)) return fail("perturb_part_ids names a body part the character does not have (or beyond bit 30 of the 32-bit part mask)"); m |= 1 << b; }
          t.perturb_part_mask = m;
      } }
    *out = sc.release();
    return 0;
}

const dm_scene_tables* dm_scene_get_tables(const dm_scene* s) { return s ? &s->t : nullptr; }

int dm_scene_info(const dm_scene* s, double* out) {
    if (!s || !out) return fail("null argument");
    out[0] = s->num_update_substeps; out[1] = s->anneal_samples; out[2] = s->time_end_lim_min; out[3] = s->time_end_lim_max; out[4] = s->time_end_lim_exp;
    out[5] = s->time_lim_exp; out[6] = s->timer_exp_type; out[7] = 0;
    return 0;
}

int dm_scene_free(dm_scene* s) { delete s; return 0; }

}  // extern "C"
