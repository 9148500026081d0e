// TEST INFRASTRUCTURE ONLY -- lets the reference's OWN routines run where their translation units could not be compiled before.
//
// sim/{Controller, CharController, DeepMimicCharController, CtController, CtPDController, PDController, ExpPDController,
// ImpPDController}.cpp and scenes/{Scene, RLScene, SceneSimChar, RLSceneSimChar, SceneImitate, SceneImitateAMP, SceneHeadingAMP,
// SceneTargetAMP, SceneStrikeAMP, SceneDribbleAMP}.cpp are compiled UNMODIFIED, where they lie, against oracle/bullet_stub (Bullet
// TYPE NAMES only) into oracle/_ref/libdm_ref.so (oracle/build_ref.sh).  What those routines need of a simulated character is an object
// that answers the (virtual) getters of cSimCharacter / cSimBodyLink / cSimBodyJoint / cGround, whose own translation units are Bullet
// code: this file provides such objects as LINK-TIME STAND-INS -- subclasses that answer the getters from a generalized state
// (pose, vel) through the reference's compiled kinematic-tree functions (cKinTree::BodyWorldTrans, CalcBodyPartVel,
// CalcJointWorldAngularVel, ...), which is what Bullet's link transforms are for a consistent multibody state.  Every other method of
// those classes resolves to ref_unreachable() (build_ref.sh aliases what stays undefined): a routine that wanders into Bullet aborts
// loudly instead of computing something made up.
//
// The C entry points at the bottom construct the reference's controller / scene objects the way the reference does (real Init from
// the shipped controller file) and call the routine named in each comment.  They replace the compositions of ref_glue.cpp
// (ref_spd_tau, ref_record_state, ref_reward_terms, ref_action_to_target, ref_amp_obs), which stay as a second witness.
// Only tests/ (and tests/golden/make_ref_golden.py) load the library.
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "anim/KinCharacter.h"
#include "anim/KinTree.h"
#include "anim/MotionController.h"
#include "scenes/SceneDribbleAMP.h"
#include "scenes/SceneHeadingAMP.h"
#include "scenes/SceneHeadingAMPGetup.h"
#include "scenes/SceneImitate.h"
#include "scenes/SceneImitateAMP.h"
#include "scenes/SceneStrikeAMP.h"
#include "scenes/SceneTargetAMP.h"
#include "sim/AgentRegistry.h"
#include "sim/CtPDController.h"
#include "sim/CtrlBuilder.h"
#include "sim/Ground.h"
#include "sim/SimCharacter.h"
#include "anim/KinCtrlBuilder.h"
#include "anim/ClipsController.h"
#include "util/ArgParser.h"
#include <unistd.h>

extern "C" void ref_unreachable() {
    fprintf(stderr, "libdm_ref: a reference method without a stand-in was called (it lives in a Bullet translation unit); callers:\n");
    void* bt[12]; int n = backtrace(bt, 12); backtrace_symbols_fd(bt, n, 2);
    abort();
}

// ---- constructors / destructors of the reference classes whose own translation units are Bullet code -----------------------------
cContactManager::tContactHandle::tContactHandle() : mID(-1), mFlags(0), mFilterFlags(0) {}       // (sim/ContactManager.cpp:9-14: invalid handle)
cSimObj::cSimObj() : mEnableContactFall(false), mType(eTypeDynamic), mColGroup(0), mColMask(0) {}
cSimObj::~cSimObj() {}
cSimRigidBody::cSimRigidBody() {}
cSimRigidBody::~cSimRigidBody() {}
cGround::cGround() {}
cGround::~cGround() {}
cGround::tParams::tParams() : mType(eTypePlane), mFriction(0.9), mOrigin(tVector::Zero()), mBlend(0), mGroundWidth(0), mVertSpacingX(0), mVertSpacingZ(0), mRandSeed(0), mHasRandSeed(false) {}
cSimBodyLink::tParams::tParams() : mMass(0), mJointID(-1) {}
cSimBodyLink::cSimBodyLink() : mJointID(-1), mMass(0), mSize(tVector::Zero()), mObjShape(cShape::eShapeNull), mLinVel(tVector::Zero()), mAngVel(tVector::Zero()) {}
cSimBodyLink::~cSimBodyLink() {}
cSimBodyJoint::tParams::tParams() : mID(-1), mLimLow(tVector::Zero()), mLimHigh(tVector::Zero()), mTorqueLimit(0), mForceLimit(0),
    mParentPos(tVector::Zero()), mChildPos(tVector::Zero()), mParentRot(tQuaternion::Identity()), mChildRot(tQuaternion::Identity()) {}
cSimBodyJoint::cSimBodyJoint() : mType(cKinTree::eJointTypeNone) { mTotalTau.setZero(); }
cSimBodyJoint::~cSimBodyJoint() {}
cSimCharacter::tParams::tParams() : mID(-1), mInitPos(tVector::Zero()), mLoadDrawShapes(false), mEnableContactFall(true) {}
cSimCharacter::cSimCharacter() : mFriction(0.9), mInvRootAttachRot(tQuaternion::Identity()) {}
cSimCharacter::~cSimCharacter() {}
// scene plumbing members that a scene object default-constructs (never used by the routines called below)
cWorld::tParams::tParams() : mNumSubsteps(1), mScale(1), mGravity(tVector(0, -9.8, 0, 0)) {}
cSimCharBuilder::eCharType dm_unused_char_type_;
cCtrlBuilder::tCtrlParams::tCtrlParams() : mCharCtrl(cCtrlBuilder::eCharCtrlNone), mGravity(tVector(0, -9.8, 0, 0)) {}

namespace {

typedef Eigen::VectorXd VecX;
typedef Eigen::MatrixXd MatX;
VecX vin(const double* p, int n) { VecX v(n); for (int i = 0; i < n; ++i) v[i] = p[i]; return v; }
void vout(const VecX& v, double* p) { for (int i = 0; i < (int)v.size(); ++i) p[i] = v[i]; }

class StandinChar;

// flat ground at a given height (cGroundPlane::SampleHeight, sim/GroundPlane.cpp)
class StandinGround : public cGround {
   public:
    double h;
    explicit StandinGround(double h_) : h(h_) {}
    double SampleHeight(const tVector&) const override { return h; }
    double SampleHeight(const tVector&, bool& valid) const override { valid = true; return h; }
    eClass GetGroundClass() const override { return eClassPlane; }
    void SamplePlacement(const tVector&, tVector& out_pos, tQuaternion& out_rot) override { out_pos.setZero(); out_pos[1] = h; out_rot.setIdentity(); }      // cGround::SamplePlacement (sim/Ground.cpp:154-159)
    // a plane has no horizontal bounds (sim/GroundPlane.cpp CalcAABB: +-infinity in x, z)
    void CalcAABB(tVector& mn, tVector& mx) const override { const double inf = std::numeric_limits<double>::infinity(); mn = tVector(-inf, h, -inf, 0); mx = tVector(inf, h, inf, 0); }
};

// one body part: transforms and velocities of the link for the character's current (pose, vel)
class StandinLink : public cSimBodyLink {
   public:
    const StandinChar* ch; int id;
    StandinLink(const StandinChar* c, int j, double mass) : ch(c), id(j) { mJointID = j; mMass = mass; }
    tVector GetPos() const override;
    tQuaternion GetRotation() const override;
    void GetRotation(tVector& axis, double& theta) const override { cMathUtil::QuaternionToAxisAngle(GetRotation(), axis, theta); }
    tMatrix GetWorldTransform() const override;
    tVector GetLinearVelocity() const override;
    tVector GetAngularVelocity() const override;
    double GetMass() const override { return mMass; }
    int GetJointID() const override { return id; }
};

// one joint: type, limits and its slice of the character's pose / vel (sim/SimBodyJoint.cpp:342-445 read the same numbers from Bullet)
class StandinJoint : public cSimBodyJoint {
   public:
    StandinChar* ch = nullptr; int id = -1; bool valid = false;
    VecX tau;
    bool IsValid() const override { return valid; }
    cKinTree::eJointType GetType() const override;
    bool IsRoot() const override;
    int GetParamSize() const override;
    void BuildPose(VecX& out) const override;
    void BuildVel(VecX& out) const override;
    double GetTorqueLimit() const override;
    double GetForceLimit() const override;
    void AddTau(const VecX& t) override { tau = t; }
    tVector CalcWorldPos() const override;
    tQuaternion CalcWorldRotation() const override;
    void CalcWorldRotation(tVector& axis, double& theta) const override { cMathUtil::QuaternionToAxisAngle(CalcWorldRotation(), axis, theta); }
    tMatrix BuildWorldTrans() const override;
};

class StandinChar : public cSimCharacter {
   public:
    std::vector<std::shared_ptr<cSimBodyLink>> links; tEigenArr<StandinJoint> joints; std::shared_ptr<cSimBodyLink> null_link;
    VecX applied_tau; std::shared_ptr<cWorld> no_world; std::shared_ptr<cCharController> ctrl; int contact_mask = 0;

    // skeleton and body definitions by the reference's own loaders (anim/Character.cpp: cCharacter::Init -> LoadSkeleton;
    // anim/KinTree.cpp:125-172 LoadBodyDefs), as cSimCharacter::Init does before it builds the Bullet body (SimCharacter.cpp:35-71)
    bool Load(const std::string& char_file) {
        if (!cCharacter::Init(char_file, false)) return false;
        if (!cKinTree::LoadBodyDefs(char_file, mBodyDefs)) return false;
        const int J = GetNumJoints();
        links.resize(J); joints.resize(J);
        for (int j = 0; j < J; ++j) {
            const bool has_body = cKinTree::IsValidBody(mBodyDefs, j);
            if (has_body) links[j] = std::shared_ptr<cSimBodyLink>(new StandinLink(this, j, cKinTree::GetBodyMass(mBodyDefs, j)));
            joints[j].ch = this; joints[j].id = j; joints[j].valid = has_body;      // BuildJoints: one joint per valid body part (SimCharacter.cpp:1036-1086)
        }
        mPose = mPose0; mVel = mVel0;
        SetID(0);                                   // character 0 of the scene (cSceneSimChar::BuildCharacters numbers them)
        return true;
    }
    void Clear() override { cCharacter::Clear(); }
    void Reset() override { cCharacter::Reset(); }
    void Set(const VecX& p, const VecX& v) { mPose = p; mVel = v; }
    void SetPose(const VecX& p) override { mPose = p; }
    void SetVel(const VecX& v) override { mVel = v; }
    // cSimCharacter::SetRootTransform (sim/SimCharacter.cpp:185-202) without the write-through to the Bullet body
    void SetRootTransform(const tVector& pos, const tQuaternion& rot) override {
        const tQuaternion delta = rot * cKinTree::GetRootRot(mPose).inverse();
        const tVector v = cMathUtil::QuatRotVec(delta, cKinTree::GetRootVel(mVel)), w = cMathUtil::QuatRotVec(delta, cKinTree::GetRootAngVel(mVel));
        cKinTree::SetRootPos(pos, mPose); cKinTree::SetRootRot(rot, mPose); cKinTree::SetRootVel(v, mVel); cKinTree::SetRootAngVel(w, mVel);
    }
    // pose-vector getters: what SimCharacter.cpp:124-161 reads back from the Bullet base are the root slots of the pose it just built
    tVector GetRootPos() const override { return cKinTree::GetRootPos(mPose); }
    tQuaternion GetRootRotation() const override { return cKinTree::GetRootRot(mPose); }
    void GetRootRotation(tVector& axis, double& theta) const override { cMathUtil::QuaternionToAxisAngle(GetRootRotation(), axis, theta); }
    tVector GetRootVel() const override { return cKinTree::GetRootVel(mVel); }
    tVector GetRootAngVel() const override { return cKinTree::GetRootAngVel(mVel); }
    tQuaternion CalcHeadingRot() const override { return cKinTree::CalcHeadingRot(mPose); }
    const MatX& GetBodyDefs() const override { return mBodyDefs; }
    int GetNumBodyParts() const override { return (int)links.size(); }
    bool IsValidBodyPart(int idx) const override { return links[idx] != nullptr; }
    const std::shared_ptr<cSimBodyLink>& GetBodyPart(int idx) const override { return links[idx]; }
    std::shared_ptr<cSimBodyLink>& GetBodyPart(int idx) override { return links[idx]; }
    tVector GetBodyPartPos(int idx) const override { return links[idx]->GetPos(); }
    tVector GetBodyPartVel(int idx) const override { return links[idx]->GetLinearVelocity(); }
    const cSimBodyJoint& GetJoint(int j) const override { return joints[j]; }
    cSimBodyJoint& GetJoint(int j) override { return joints[j]; }
    // SimCharacter.cpp:308-331: the joint's world position when the joint is valid, the body part's otherwise
    tVector CalcJointPos(int j) const override { return cKinTree::CalcJointWorldPos(mJointMat, mPose, j); }
    tQuaternion CalcJointWorldRotation(int j) const override { tVector a; double th; cKinTree::CalcJointWorldTheta(mJointMat, mPose, j, a, th); return cMathUtil::AxisAngleToQuaternion(a, th); }
    tMatrix BuildJointWorldTrans(int j) const override { return cKinTree::JointWorldTrans(mJointMat, mPose, j); }
    // SimCharacter.cpp:398-436: mass-weighted mean of the body parts' positions / linear velocities
    tVector CalcCOM() const override {
        tVector com = tVector::Zero(); double m = 0;
        for (int i = 0; i < GetNumBodyParts(); ++i) if (IsValidBodyPart(i)) { const double mi = links[i]->GetMass(); com += mi * links[i]->GetPos(); m += mi; }
        return com / m;
    }
    tVector CalcCOMVel() const override {
        tVector v = tVector::Zero(); double m = 0;
        for (int i = 0; i < GetNumBodyParts(); ++i) if (IsValidBodyPart(i)) { const double mi = links[i]->GetMass(); v += mi * links[i]->GetLinearVelocity(); m += mi; }
        return v / m;
    }
    double CalcTotalMass() const override { return cKinTree::CalcTotalMass(mBodyDefs); }
    void ApplyControlForces(const VecX& tau) override { applied_tau = tau; }
    bool fallen = false;
    bool HasFallen() const override { return fallen; }
    bool IsInContact(int idx) const override { return (contact_mask >> idx) & 1; }
    bool IsInContact() const override { return contact_mask != 0; }
    const std::shared_ptr<cWorld>& GetWorld() const override { return no_world; }
    const std::shared_ptr<cCharController>& GetController() override { return ctrl; }
    const std::shared_ptr<cCharController>& GetController() const override { return ctrl; }
    void SetController(std::shared_ptr<cCharController> c) override { ctrl = c; }
    // cSimObj face of the character (SimCharacter.cpp: the root link's)
    tVector GetPos() const override { return GetRootPos(); }
    tQuaternion GetRotation() const override { return GetRootRotation(); }
    tVector GetLinearVelocity() const override { return GetRootVel(); }
    tVector GetAngularVelocity() const override { return GetRootAngVel(); }
    const MatX& JointMat() const { return mJointMat; }
    const VecX& Pose() const { return mPose; }
    const VecX& Vel() const { return mVel; }
};

tVector StandinLink::GetPos() const { tMatrix m = cKinTree::BodyWorldTrans(ch->JointMat(), ch->GetBodyDefs(), ch->Pose(), id); return tVector(m(0, 3), m(1, 3), m(2, 3), 0); }
tQuaternion StandinLink::GetRotation() const { return cMathUtil::RotMatToQuaternion(cKinTree::BodyWorldTrans(ch->JointMat(), ch->GetBodyDefs(), ch->Pose(), id)); }
tMatrix StandinLink::GetWorldTransform() const { return cKinTree::BodyWorldTrans(ch->JointMat(), ch->GetBodyDefs(), ch->Pose(), id); }
tVector StandinLink::GetLinearVelocity() const { return cKinTree::CalcBodyPartVel(ch->JointMat(), ch->GetBodyDefs(), ch->Pose(), ch->Vel(), id); }
tVector StandinLink::GetAngularVelocity() const { return cKinTree::CalcJointWorldAngularVel(ch->JointMat(), ch->Pose(), ch->Vel(), id); }

cKinTree::eJointType StandinJoint::GetType() const { return cKinTree::GetJointType(ch->JointMat(), id); }
bool StandinJoint::IsRoot() const { return cKinTree::IsRoot(ch->JointMat(), id); }
int StandinJoint::GetParamSize() const { return cKinTree::GetParamSize(ch->JointMat(), id); }
void StandinJoint::BuildPose(VecX& out) const { cKinTree::GetJointParams(ch->JointMat(), ch->Pose(), id, out); }
void StandinJoint::BuildVel(VecX& out) const { cKinTree::GetJointParams(ch->JointMat(), ch->Vel(), id, out); }
double StandinJoint::GetTorqueLimit() const { return cKinTree::GetTorqueLimit(ch->JointMat(), id); }
double StandinJoint::GetForceLimit() const { return cKinTree::GetForceLimit(ch->JointMat(), id); }
tVector StandinJoint::CalcWorldPos() const { return cKinTree::CalcJointWorldPos(ch->JointMat(), ch->Pose(), id); }
tQuaternion StandinJoint::CalcWorldRotation() const { tVector a; double th; cKinTree::CalcJointWorldTheta(ch->JointMat(), ch->Pose(), id, a, th); return cMathUtil::AxisAngleToQuaternion(a, th); }
tMatrix StandinJoint::BuildWorldTrans() const { return cKinTree::JointWorldTrans(ch->JointMat(), ch->Pose(), id); }

// a free rigid body (the ball of cSceneDribbleAMP): cSimRigidBody reads these from its btRigidBody (sim/SimRigidBody.cpp)
class StandinBody : public cSimRigidBody {
   public:
    tVector pos = tVector::Zero(), lin = tVector::Zero(), ang = tVector::Zero(); tQuaternion rot = tQuaternion::Identity();
    tVector GetPos() const override { return pos; }
    tQuaternion GetRotation() const override { return rot; }
    void GetRotation(tVector& axis, double& theta) const override { cMathUtil::QuaternionToAxisAngle(rot, axis, theta); }
    tVector GetLinearVelocity() const override { return lin; }
    tVector GetAngularVelocity() const override { return ang; }
    tVector GetSize() const override { return tVector::Zero(); }      // (cSimSphere::GetSize; no routine called here asks)
    void SetPos(const tVector& p) override { pos = p; }
    void SetRotation(const tVector& axis, double theta) override { rot = cMathUtil::AxisAngleToQuaternion(axis, theta); }
    void SetRotation(const tQuaternion& q) override { rot = q; }
    void SetLinearVelocity(const tVector& v) override { lin = v; }
    void SetAngularVelocity(const tVector& v) override { ang = v; }
};

// the reference's controller with its protected parts reachable
class CtrlX : public cCtPDController {
   public:
    using cCtPDController::UpdateBuildTau;
    using cCtPDController::ApplyAction;
    using cCtController::CheckNeedNewAction;
    cImpPDController& pd() { return mPDCtrl; }
    void set_time(double t) { mTime = t; }
    void set_prev_action(double t, const tVector& com) { mPrevActionTime = t; mPrevActionCOM = com; }
};
// the reference's scenes with the few protected members the routines read set from outside
template <class SCENE>
class SceneX : public SCENE {
   public:
    void setup(const std::shared_ptr<cSimCharacter>& ch, const std::shared_ptr<cKinCharacter>& kin, double ground_h) {
        this->mChars.clear(); this->mChars.push_back(ch);
        this->mGround = std::shared_ptr<cGround>(new StandinGround(ground_h));
        this->mKinChar = kin;
        this->CalcJointWeights(ch, this->mJointWeights);                 // scenes/SceneImitate.cpp:236-248 (InitJointWeights)
        this->mAgentReg.Clear(); this->mAgentReg.AddAgent(ch->GetController(), ch.get());      // cRLSceneSimChar::RegisterAgents (RLSceneSimChar.cpp:27-37)
    }
    void amp_prev(const VecX& pp, const VecX& pv, bool local_root) { this->mPrevPose = pp; this->mPrevVel = pv; this->mEnableAMPObsLocalRoot = local_root; }
    void target(const tVector& pos, double speed, double succ_dist, double fail_dist, bool min_tar_vel, double pos_scale) {
        this->mTargetPos = pos; this->mTargetSpeed = speed; this->mTargetSuccDist = succ_dist; this->mTarFailDist = fail_dist; this->mEnableMinTarVel = min_tar_vel; this->mPosRewardScale = pos_scale;
    }
    void heading(double h, double vel_scale) { this->mTargetHeading = h; this->mVelRewardScale = vel_scale; }
    void scene_time(double t) { this->mTimer.SetTime(t); }
    // cSceneHeadingAMPGetup (instantiated for that scene only)
    void getup(double getup_time, double timer_time, double h_root, double h_head, int head_id) {
        this->mGetupTime = getup_time; this->mGetupTimer.SetMaxTime(getup_time); this->mGetupTimer.SetTime(timer_time);
        this->mGetupHeightRoot = h_root; this->mGetupHeightHead = h_head; this->mHeadID = head_id;
    }
    bool getup_running() const { return this->CheckGettingUp(); }
    bool getup_fallen_contact(const cSimCharacter& c) const { return this->HasFallenContact(c); }
    // cSceneStrikeAMP (instantiated for that scene only)
    void strike(double near_dist, double radius, double tar_scale, double hit_speed, unsigned strike_mask, unsigned fail_mask, bool hit, double hit_time, double hit_reset_time) {
        this->mTarNearDist = near_dist; this->mTargetRadius = radius; this->mTarRewardScale = tar_scale; this->mHitTarSpeed = hit_speed;
        this->mStrikeBodies.clear(); this->mFailTarContactBodies.clear();
        for (int j = 0; j < 32; ++j) { if ((strike_mask >> j) & 1u) this->mStrikeBodies.push_back(j); if ((fail_mask >> j) & 1u) this->mFailTarContactBodies.push_back(j); }
        this->mTargetHit = hit; this->mTargetHitTime = hit_time; this->mTargetHitResetTime = hit_reset_time;
    }
    bool strike_check_hit() const { return this->CheckTargetHit(); }
    bool strike_contact_fail() const { return this->CheckTarContactFail(0); }
    bool strike_hit_succ() const { return this->CheckTarHitSucc(); }
    // cSceneDribbleAMP (instantiated for that scene only): the target object is the stand-in body
    std::shared_ptr<cSimRigidBody> ball;
    const std::shared_ptr<cSimRigidBody>& GetObj(int) const override { return ball; }
    void dribble(const std::shared_ptr<cSimRigidBody>& b, const tVector& prev_ball, double max_target_dist, double max_tar_obj_dist) {
        ball = b; this->mTarObjID = 0; this->mAgentPrevTarObjPos.assign(1, prev_ball); this->mMaxTargetDist = max_target_dist; this->mMaxTarObjDist = max_tar_obj_dist;
    }
    void dribble_task_state(VecX& out) const { this->RecordTaskState(0, out); }
    bool dribble_succ() const { return this->CheckTargetSucc(); }
    bool dribble_tar_obj_fail() const { return this->CheckTarObjDistFail(); }
    bool dribble_char_obj_fail(const cSimCharacter& c) const { return this->CheckCharObjDistFail(c); }
    bool dribble_has_fallen(const cSimCharacter& c) const { return this->HasFallen(c); }
    double reward_imitate(const cSimCharacter& sim, const cKinCharacter& kin) const { return this->CalcRewardImitate(sim, kin); }
    // ---- draw session (ref3_*): the parts of Init / Reset / Update that draw random numbers, each the reference's own compiled method
    cSimObj* pert_obj = nullptr; tVector pert_force = tVector::Zero(); double pert_dur = 0; int n_perturbs = 0;
    void AddPerturb(const tPerturb& p) override { pert_obj = p.mObj; pert_force = p.mPerturb; pert_dur = p.mDuration; ++n_perturbs; }      // (cSceneSimChar::AddPerturb hands it to the Bullet world; kept here instead)
    void d_parse(const std::shared_ptr<cArgParser>& parser) { this->ParseArgs(parser); this->mCharParams.resize(1); }
    void d_mode(int test) { this->mMode = test ? cRLScene::eModeTest : cRLScene::eModeTrain; }
    void d_rl_scene_init() { this->cRLScene::Init(); }                               // cScene::Init: InitTimers + ResetParams (virtual)
    void d_scene_init() { this->cScene::Init(); }
    bool d_perturbs() const { return this->mPerturbParams.mEnableRandPerturbs; }
    void d_reset_perturb() { this->ResetRandPertrub(); }
    void d_update_perturb(double dt) { this->UpdateRandPerturb(dt); }
    void d_setup_annealer() { this->SetupTimerAnnealer(this->mTimerAnnealer); }
    void d_rl_reset_scene() { this->cRLScene::ResetScene(); }
    void d_base_reset_scene() { this->cScene::ResetScene(); }
    void d_reset_characters() { this->ResetCharacters(); }                           // virtual: cSceneImitate::ResetCharacters -> ResetKinChar, SyncCharacters
    void d_update_timers(double dt) { this->UpdateTimers(dt); }
    void d_update_kin(double dt) { this->UpdateKinChar(dt); }
    void d_sync_kin_root() { if (this->EnableSyncChar()) this->SyncKinCharRoot(); }      // the part of cSceneImitate::ResolveCharGroundIntersect (:386-394) that is not Bullet's
    void d_init_char_pos() { this->InitCharacterPos(); }                             // rand placement on a plane: the root goes to x = z = 0 (SceneSimChar.cpp:478-531)
    void d_amp_reset() { this->InitHist(); }
    double d_timer_max() const { return this->mTimer.GetMaxTime(); }
    double d_time() const { return this->mTimer.GetTime(); }
    double d_pert_next() const { return this->mPerturbParams.mNextTime; }
    double d_pert_timer() const { return this->mPerturbParams.mTimer; }
    // task scenes
    void d_target_init() { this->InitTarget(); this->ResetTarget(); }                // cSceneTargetAMP::Init after cSceneImitate::Init (:118-123)
    void d_target_reset() { this->mTargetTimer.Reset(); this->ResetTarget(); }       // cSceneTargetAMP::Reset after cSceneImitate::Reset (:125-130)
    void d_target_update(double dt) { this->UpdateTarget(dt); if (this->mTargetTimer.IsEnd()) this->mTargetTimer.Reset(); }      // cSceneTargetAMP::Update after the scene update (:132-141)
    tVector d_target_pos() const { return this->mTargetPos; }
    double d_target_speed() const { return this->mTargetSpeed; }
    double d_target_timer_max() const { return this->mTargetTimer.GetMaxTime(); }
    double d_target_timer_time() const { return this->mTargetTimer.GetTime(); }
    double d_heading() const { return this->mTargetHeading; }
    bool d_strike_hit() const { return this->mTargetHit; }
    double d_strike_hit_time() const { return this->mTargetHitTime; }
    // dribble_amp (InitTarObjs without BuildTarObjs, which builds the Bullet sphere: the ball is the stand-in body)
    void d_dribble_init() { this->mTarObjTimer.Init(this->mTarObjTimerParams); this->ResetTarObjs(); this->InitAgentTarObjRecord(); this->mTargetTimer.Reset(); this->ResetTarget(); }
    void d_dribble_reset_head() { this->mTarObjTimer.Reset(); this->ResetTarObjs(); this->ResetAgentTarObjRecord(); }
    void d_dribble_update_objs(double dt) { this->UpdateTarObjs(dt); if (this->mTarObjTimer.IsEnd()) this->mTarObjTimer.Reset(); }      // cSceneDribbleAMP::UpdateObjs (:310-319)
    double d_obj_timer_max() const { return this->mTarObjTimer.GetMaxTime(); }
    void d_dribble_ball(const std::shared_ptr<cSimRigidBody>& b) { ball = b; this->mTarObjID = 0; }
    void d_dribble_prev_ball(const tVector& p) { this->mAgentPrevTarObjPos.assign(1, p); }
    int d_expert(VecX& out) { this->RecordAMPObsExpert(0, out); return (int)out.size(); }
    int d_amp_agent(VecX& out) { this->RecordAMPObsAgent(0, out); return (int)out.size(); }
    // cRLSceneSimChar::PreUpdate -> NewActionUpdate (RLSceneSimChar.cpp:262-275): the pose history of the AMP observation (and the ball record of dribble_amp) is
    // latched when a new action starts.  (The test-mode time-warp sampler, an evaluation return of imitate_amp, is not built in these sessions.)
    void d_new_action() { this->NewActionUpdate(0); }
    // the test-mode time-warp return of imitate_amp (cSceneImitateAMP::BuildTimeWarper / ResetTimeWarper, :417-447): built once, restarted at every reset
    void d_build_time_warper() { if (!this->mTimeWarper) this->BuildTimeWarper(); }
    void d_reset_time_warper() { if (this->mTimeWarper) this->ResetTimeWarper(); }
    // heading_amp_getup
    void d_getup_init(const std::vector<int>& ids) { this->mGetupMotionIDs = ids; this->RecordGetupMotionFlags(ids); this->mGetupTime = this->CalcGetupTime(ids); this->InitGetupTimer(); this->ResetGetupTimer(); this->SyncGetupTimer(); }
    bool d_getup_activate_recovery() { return this->ActivateRecoveryEpisode(); }
    void d_getup_reset_recovery() { this->ResetRecoveryEpisode(); }
    void d_getup_sync() { this->SyncGetupTimer(); }
    double d_getup_timer() const { return this->mGetupTimer.GetTime(); }
    void d_getup_test_update() { if (this->mMode == cRLScene::eModeTest) this->UpdateTestGetup(); }      // cSceneHeadingAMPGetup::Update (:95-103): a fall starts a get-up instead of ending the episode
};

struct Rig {
    std::shared_ptr<StandinChar> ch;
    std::shared_ptr<CtrlX> ctrl;
    std::shared_ptr<cKinCharacter> kin;
    std::shared_ptr<SceneX<cSceneImitate>> imitate;
    std::shared_ptr<SceneX<cSceneImitateAMP>> amp;
    std::shared_ptr<SceneX<cSceneTargetAMP>> target;
    std::shared_ptr<SceneX<cSceneHeadingAMP>> heading;
    std::shared_ptr<SceneX<cSceneStrikeAMP>> strike;
    std::shared_ptr<SceneX<cSceneHeadingAMPGetup>> getup;
    std::shared_ptr<SceneX<cSceneDribbleAMP>> dribble;
    std::shared_ptr<StandinBody> ball;
};

}  // namespace

extern "C" {

// character file + controller file -> stand-in character, the reference's cCtPDController initialised the way
// cCtrlBuilder::BuildCtPDController does (sim/CtrlBuilder.cpp:89-99: SetGravity, Init(character, file)), and (optionally) the reference's
// kinematic character on `motion_file`
void* ref2_create(const char* char_file, const char* ctrl_file, const char* motion_file, const double* gravity3) {
    Rig* r = new Rig();
    r->ch = std::shared_ptr<StandinChar>(new StandinChar());
    if (!r->ch->Load(char_file)) { delete r; return nullptr; }
    r->ctrl = std::shared_ptr<CtrlX>(new CtrlX());
    r->ctrl->SetGravity(tVector(gravity3[0], gravity3[1], gravity3[2], 0));
    r->ctrl->Init(r->ch.get(), ctrl_file);
    r->ch->SetController(r->ctrl);
    if (motion_file && motion_file[0]) {
        r->kin = std::shared_ptr<cKinCharacter>(new cKinCharacter());
        cKinCharacter::tParams p; p.mCharFile = char_file; p.mLoadDrawShapes = false;
        if (!r->kin->Init(p)) { delete r; return nullptr; }
        std::shared_ptr<cMotionController> mc(new cMotionController());
        mc->Init(r->kin.get(), motion_file);
        r->kin->SetController(mc);
    }
    return r;
}
// (the scene objects are left alone: their destructors tear down a world this harness never built)
void ref2_destroy(void* h) {
    Rig* r = (Rig*)h;
    new std::shared_ptr<SceneX<cSceneImitate>>(r->imitate); new std::shared_ptr<SceneX<cSceneImitateAMP>>(r->amp);
    new std::shared_ptr<SceneX<cSceneTargetAMP>>(r->target); new std::shared_ptr<SceneX<cSceneHeadingAMP>>(r->heading);
    new std::shared_ptr<SceneX<cSceneHeadingAMPGetup>>(r->getup); new std::shared_ptr<SceneX<cSceneStrikeAMP>>(r->strike); new std::shared_ptr<SceneX<cSceneDribbleAMP>>(r->dribble); new std::shared_ptr<StandinBody>(r->ball);
    new std::shared_ptr<StandinChar>(r->ch); new std::shared_ptr<CtrlX>(r->ctrl);
    delete r;
}
int ref2_num_dof(void* h) { return ((Rig*)h)->ch->GetNumDof(); }
int ref2_state_size(void* h) { return ((Rig*)h)->ctrl->GetStateSize(); }
int ref2_action_size(void* h) { return ((Rig*)h)->ctrl->GetActionSize(); }
void ref2_set_state(void* h, const double* pose, const double* vel) { Rig* r = (Rig*)h; const int P = r->ch->GetNumDof(); r->ch->Set(vin(pose, P), vin(vel, P)); }

// cCtPDController::ApplyAction -> SetPDTargets -> ConvertActionToTargetPose (sim/CtPDController.cpp:97-166) -> cImpPDController::SetTargetTheta;
// out_tar: the latched PD targets in pose layout (cExpPDController::GetTargetTheta per joint; root slots 0), i.e. what
// cImpPDController::BuildTargetPose (sim/ImpPDController.cpp:197-219) hands to the solve
void ref2_apply_action(void* h, const double* action, double* out_tar) {
    Rig* r = (Rig*)h;
    r->ctrl->ApplyAction(vin(action, r->ctrl->GetActionSize()));
    const int P = r->ch->GetNumDof();
    for (int i = 0; i < P; ++i) out_tar[i] = 0;
    for (int j = 0; j < r->ch->GetNumJoints(); ++j) {
        if (!r->ctrl->pd().GetPDCtrl(j).IsValid()) continue;
        VecX th; r->ctrl->pd().GetTargetTheta(j, th);
        const int off = r->ch->GetParamOffset(j);
        for (int k = 0; k < (int)th.size(); ++k) out_tar[off + k] = th[k];
    }
}
// PD targets handed in as a pose vector: cExpPDController::SetTargetTheta per joint (sim/ExpPDController.cpp; the call ApplyAction ends in)
void ref2_set_targets(void* h, const double* tar_pose) {
    Rig* r = (Rig*)h;
    for (int j = 0; j < r->ch->GetNumJoints(); ++j) {
        if (!r->ctrl->pd().GetPDCtrl(j).IsValid()) continue;
        const int off = r->ch->GetParamOffset(j), sz = r->ch->GetParamSize(j);
        r->ctrl->pd().SetTargetTheta(j, vin(tar_pose + off, sz));
    }
}
// cCtPDController::UpdateBuildTau -> cImpPDController::UpdateControlForce -> UpdateRBDModel + CalcControlForces (sim/CtPDController.cpp:85-95;
// sim/ImpPDController.cpp:47-73,129-195): the stable-PD torque for the current state and the latched targets, pose layout, before the clamp
void ref2_spd_tau(void* h, double dt, double* out_tau) {
    Rig* r = (Rig*)h;
    VecX tau = VecX::Zero(r->ch->GetNumDof());
    r->ctrl->UpdateBuildTau(dt, tau);
    vout(tau, out_tau);
}
// cCtController::RecordState (sim/CtController.cpp:281-293 -> BuildStatePose / BuildStateVel / BuildStatePhase :373-478) with the
// controller's own flags from the file; phase = controller time / cycle period
int ref2_record_state(void* h, double phase, double ground_h, double* out) {
    Rig* r = (Rig*)h;
    r->ctrl->SetGround(std::shared_ptr<cGround>(new StandinGround(ground_h)));
    r->ctrl->SetCyclePeriod(1.0); r->ctrl->SetInitTime(0.0); r->ctrl->set_time(phase);
    VecX s; r->ctrl->RecordState(s); vout(s, out);
    return (int)s.size();
}
// the learner-facing tables of the controller (sim/CtController.cpp:54-69,183-195; sim/CtPDController.cpp BuildActionBounds /
// BuildActionOffsetScale; sim/CharController.cpp:64-89): state offset / scale / norm groups, action offset / scale / bounds
void ref2_tables(void* h, double* s_off, double* s_scale, int* s_groups, double* a_off, double* a_scale, double* a_min, double* a_max) {
    Rig* r = (Rig*)h; VecX a, b; Eigen::VectorXi g;
    r->ctrl->BuildStateOffsetScale(a, b); vout(a, s_off); vout(b, s_scale);
    r->ctrl->BuildStateNormGroups(g); for (int i = 0; i < (int)g.size(); ++i) s_groups[i] = g[i];
    r->ctrl->BuildActionOffsetScale(a, b); vout(a, a_off); vout(b, a_scale);
    r->ctrl->BuildActionBounds(a, b); vout(a, a_min); vout(b, a_max);
}
// kinematic character to a clip time with an origin (as ref_kinchar_*): cKinCharacter::SetTime + Pose
void ref2_kin_set(void* h, double t, const double* origin_pos3, const double* origin_rot4) {
    Rig* r = (Rig*)h;
    // rotation first: cKinCharacter::SetOriginRot pivots the origin about the CURRENT root position (KinCharacter.cpp:276-300), SetOriginPos then
    // pins the origin exactly (:255-260) -- in this order the result does not depend on where the last Pose() left the root
    r->kin->SetOriginRot(tQuaternion(origin_rot4[0], origin_rot4[1], origin_rot4[2], origin_rot4[3]));
    r->kin->SetOriginPos(tVector(origin_pos3[0], origin_pos3[1], origin_pos3[2], 0));
    r->kin->SetTime(t); r->kin->Pose();
}
void ref2_kin_state(void* h, double* pose, double* vel) { Rig* r = (Rig*)h; vout(r->kin->GetPose(), pose); vout(r->kin->GetVel(), vel); }
// cSceneImitate::CalcRewardImitate (scenes/SceneImitate.cpp:7-127) on the stand-in sim character and the reference's kin character,
// joint weights by cSceneImitate::CalcJointWeights (:236-248)
double ref2_reward_imitate(void* h, double ground_h) {
    Rig* r = (Rig*)h;
    if (!r->imitate) r->imitate = std::shared_ptr<SceneX<cSceneImitate>>(new SceneX<cSceneImitate>());
    r->imitate->setup(r->ch, r->kin, ground_h);
    return r->imitate->reward_imitate(*r->ch, *r->kin);
}

// cSceneImitateAMP::RecordAMPObsAgent -> BuildAMPObs / RecordAMPObsPose / RecordAMPObsVel (scenes/SceneImitateAMP.cpp:101-113,279-396) with the
// history (mPrevPose / mPrevVel) handed in; returns the observation size (GetAMPObsSize)
int ref2_amp_obs(void* h, const double* prev_pose, const double* prev_vel, int local_root, double ground_h, double* out) {
    Rig* r = (Rig*)h; const int P = r->ch->GetNumDof();
    if (!r->amp) r->amp = std::shared_ptr<SceneX<cSceneImitateAMP>>(new SceneX<cSceneImitateAMP>());
    r->amp->setup(r->ch, r->kin, ground_h);
    r->amp->amp_prev(vin(prev_pose, P), vin(prev_vel, P), local_root != 0);
    VecX o; r->amp->RecordAMPObsAgent(0, o); vout(o, out);
    return (int)o.size();
}
// task scenes: cSceneTargetAMP::CalcReward / RecordGoal (scenes/SceneTargetAMP.cpp:3-81,192-218) and cSceneHeadingAMP::CalcReward / RecordGoal
// (scenes/SceneHeadingAMP.cpp:3-43,134-149) on the stand-in character.  par: [target x, y, z, target speed, succ dist, fail dist,
// enable_min_tar_vel, pos reward scale, target heading, vel reward scale, prev action time, prev action COM x, y, z, controller time, fallen];
// out: [reward, goal...]; returns the goal size.
// kind 3, cSceneHeadingAMPGetup::CalcReward / RecordGoal / CheckGettingUp / HasFallenContact (scenes/SceneHeadingAMPGetup.cpp:4-38, 119-126,
// 255-264, 292-303): par continues [16 get-up time, get-up timer, root height, head height, head body id, contact mask]; out continues behind
// the goal with [getting up, fallen by contact].
// kind 4, cSceneStrikeAMP::CalcReward (train) / RecordGoal / CheckTargetHit / CheckTarContactFail / CheckTarHitSucc (scenes/SceneStrikeAMP.cpp:
// 23-187, 390-434, 441-520): par continues [16 near dist, target radius, target reward scale, hit speed, strike body mask, forbidden body mask,
// target hit, hit time, scene time, hit reset time]; out continues behind the goal with [check hit, contact fail, hit succ].
// kind 5, cSceneDribbleAMP::CalcReward (train) / RecordGoal / RecordTaskState / CheckTargetSucc / the distance failures / HasFallen
// (scenes/SceneDribbleAMP.cpp:21-106, 267-292, 343-381, 454-466, 556-589): par continues [16..28 ball position, rotation wxyz, linear and angular
// velocity, 29..31 ball position at the last action, max target dist, max target-object dist]; out continues behind the goal with
// [task state (15), succ, target-object distance fail, character-object distance fail, has fallen]
int ref2_task_scene(void* h, int kind, const double* par, double* out) {
    Rig* r = (Rig*)h;
    r->ctrl->set_time(par[14]); r->ctrl->set_prev_action(par[10], tVector(par[11], par[12], par[13], 0));
    r->ch->fallen = par[15] != 0;
    VecX g; double rew = 0;
    if (kind == 1) {
        if (!r->target) r->target = std::shared_ptr<SceneX<cSceneTargetAMP>>(new SceneX<cSceneTargetAMP>());
        r->target->setup(r->ch, r->kin, 0.0);
        r->target->target(tVector(par[0], par[1], par[2], 0), par[3], par[4], par[5], par[6] != 0, par[7]);
        rew = r->target->CalcReward(0); r->target->RecordGoal(0, g);
    } else if (kind == 3) {
        if (!r->getup) r->getup = std::shared_ptr<SceneX<cSceneHeadingAMPGetup>>(new SceneX<cSceneHeadingAMPGetup>());
        auto& sc = *r->getup;
        sc.setup(r->ch, r->kin, 0.0);
        sc.target(tVector(par[0], par[1], par[2], 0), par[3], par[4], par[5], par[6] != 0, par[7]);
        sc.heading(par[8], par[9]);
        sc.getup(par[16], par[17], par[18], par[19], (int)par[20]);
        r->ch->contact_mask = (int)par[21];
        rew = sc.CalcReward(0); sc.RecordGoal(0, g);
        out[0] = rew; for (int i = 0; i < (int)g.size(); ++i) out[1 + i] = g[i];
        double* x = out + 1 + g.size();
        x[0] = sc.getup_running(); x[1] = sc.getup_fallen_contact(*r->ch);
        r->ch->contact_mask = 0;
        return (int)g.size();
    } else if (kind == 4) {
        if (!r->strike) r->strike = std::shared_ptr<SceneX<cSceneStrikeAMP>>(new SceneX<cSceneStrikeAMP>());
        auto& sc = *r->strike;
        sc.setup(r->ch, r->kin, 0.0);
        sc.target(tVector(par[0], par[1], par[2], 0), par[3], par[4], par[5], par[6] != 0, par[7]);
        sc.strike(par[16], par[17], par[18], par[19], (unsigned)par[20], (unsigned)par[21], par[22] != 0, par[23], par[25]);
        sc.scene_time(par[24]);
        rew = sc.CalcReward(0); sc.RecordGoal(0, g);
        out[0] = rew; for (int i = 0; i < (int)g.size(); ++i) out[1 + i] = g[i];
        double* x = out + 1 + g.size();
        x[0] = sc.strike_check_hit(); x[1] = sc.strike_contact_fail(); x[2] = sc.strike_hit_succ();
        return (int)g.size();
    } else if (kind == 5) {
        if (!r->dribble) r->dribble = std::shared_ptr<SceneX<cSceneDribbleAMP>>(new SceneX<cSceneDribbleAMP>());
        if (!r->ball) r->ball = std::shared_ptr<StandinBody>(new StandinBody());
        auto& sc = *r->dribble;
        sc.setup(r->ch, r->kin, 0.0);
        sc.target(tVector(par[0], par[1], par[2], 0), par[3], par[4], par[5], par[6] != 0, par[7]);
        StandinBody& b = *r->ball;
        b.pos = tVector(par[16], par[17], par[18], 0); b.rot = tQuaternion(par[19], par[20], par[21], par[22]);
        b.lin = tVector(par[23], par[24], par[25], 0); b.ang = tVector(par[26], par[27], par[28], 0);
        sc.dribble(r->ball, tVector(par[29], par[30], par[31], 0), par[32], par[33]);
        rew = sc.CalcReward(0); sc.RecordGoal(0, g);
        out[0] = rew; for (int i = 0; i < (int)g.size(); ++i) out[1 + i] = g[i];
        double* x = out + 1 + g.size();
        VecX ts; sc.dribble_task_state(ts);
        for (int i = 0; i < (int)ts.size(); ++i) x[i] = ts[i];
        x += ts.size();
        x[0] = sc.dribble_succ(); x[1] = sc.dribble_tar_obj_fail(); x[2] = sc.dribble_char_obj_fail(*r->ch); x[3] = sc.dribble_has_fallen(*r->ch);
        return (int)g.size();
    } else {
        if (!r->heading) r->heading = std::shared_ptr<SceneX<cSceneHeadingAMP>>(new SceneX<cSceneHeadingAMP>());
        r->heading->setup(r->ch, r->kin, 0.0);
        r->heading->target(tVector(par[0], par[1], par[2], 0), par[3], par[4], par[5], par[6] != 0, par[7]);
        r->heading->heading(par[8], par[9]);
        rew = r->heading->CalcReward(0); r->heading->RecordGoal(0, g);
    }
    out[0] = rew; for (int i = 0; i < (int)g.size(); ++i) out[1 + i] = g[i];
    return (int)g.size();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------
// Draw session: the reference's draw ORDER on its two generators (cMathUtil::gRand and the scene's cScene::mRand), produced by the reference's own
// compiled methods in the order cDeepMimicCore::Init / cScene::Reset / cScene::Update call them.  What cannot run here is the Bullet world between
// those calls; the functions below therefore issue the sequence of calls of cSceneSimChar::Init (scenes/SceneSimChar.cpp:106-123), ::ResetScene
// (:628-644), ::Update (:136-167), cSceneTargetAMP::Init / Reset / Update (scenes/SceneTargetAMP.cpp:118-141), cSceneDribbleAMP::Init / Reset /
// UpdateObjs (scenes/SceneDribbleAMP.cpp:151-170, 310-319) and cSceneHeadingAMPGetup::Init / Reset (scenes/SceneHeadingAMPGetup.cpp:83-121), each
// element of which is the reference's compiled method.  kind: 0 imitate_amp, 1 target_amp, 2 heading_amp, 3 heading_amp_getup, 4 strike_amp,
// 5 dribble_amp.  `tokens`: the scene's arg-file keys as command-line tokens, parsed by the scene's own ParseArgs (the keys that name builders
// whose translation units are Bullet code -- char_types, char_ctrls, kin_ctrl, terrain_file -- are left out by the caller).
namespace {
struct Draw {
    Rig* rig = nullptr; int kind = 0; bool clips = false;
    std::shared_ptr<cClipsController> clips_ctrl;
    std::vector<int> getup_ids;
};
template <class F> void with_scene(Draw* d, F f) {
    Rig* r = d->rig;
    switch (d->kind) {
    case 0: f(*r->amp); break;
    case 1: f(*r->target); break;
    case 2: f(*r->heading); break;
    case 3: f(*r->getup); break;
    case 4: f(*r->strike); break;
    case 5: f(*r->dribble); break;
    default: f(*r->imitate); break;          // 6: `--scene imitate`
    }
}
// what only the AMP scene classes have
template <class S> void amp_init_hist(S& s) { s.d_amp_reset(); }
template <> void amp_init_hist(SceneX<cSceneImitate>&) {}
template <class S> int amp_expert(S& s, VecX& v) { return s.d_expert(v); }
template <> int amp_expert(SceneX<cSceneImitate>&, VecX&) { return 0; }
template <class S> int amp_agent(S& s, VecX& v) { return s.d_amp_agent(v); }
template <> int amp_agent(SceneX<cSceneImitate>&, VecX&) { return 0; }
template <class S> void amp_new_action(S& s) { s.d_new_action(); }
template <> void amp_new_action(SceneX<cSceneImitate>&) {}
template <class S> int amp_tables(S& s, int which, VecX& a) {
    Eigen::VectorXi g;
    if (which == 10) s.GetAMPObsOffset(a); else if (which == 11) s.GetAMPObsScale(a); else { s.GetAMPObsNormGroup(g); a = g.cast<double>(); }
    return (int)a.size();
}
template <> int amp_tables(SceneX<cSceneImitate>&, int, VecX&) { return 0; }
template <class S> void amp_time_warper(S& s, bool build) { if (build) s.d_build_time_warper(); else s.d_reset_time_warper(); }
template <> void amp_time_warper(SceneX<cSceneImitate>&, bool) {}
template <class S> void draw_task_init(S&, Draw*) {}
template <> void draw_task_init(SceneX<cSceneTargetAMP>& s, Draw*) { s.d_target_init(); }
template <> void draw_task_init(SceneX<cSceneHeadingAMP>& s, Draw*) { s.d_target_init(); }
template <> void draw_task_init(SceneX<cSceneStrikeAMP>& s, Draw*) { s.d_target_init(); }
template <> void draw_task_init(SceneX<cSceneHeadingAMPGetup>& s, Draw* d) { s.d_target_init(); s.d_getup_init(d->getup_ids); }
template <> void draw_task_init(SceneX<cSceneDribbleAMP>& s, Draw*) { s.d_target_init(); s.d_dribble_init(); }
template <class S> void draw_task_reset(S&) {}
template <> void draw_task_reset(SceneX<cSceneTargetAMP>& s) { s.d_target_reset(); }
template <> void draw_task_reset(SceneX<cSceneHeadingAMP>& s) { s.d_target_reset(); }
template <> void draw_task_reset(SceneX<cSceneStrikeAMP>& s) { s.d_target_reset(); }
template <> void draw_task_reset(SceneX<cSceneHeadingAMPGetup>& s) { s.d_target_reset(); s.d_getup_sync(); }
template <> void draw_task_reset(SceneX<cSceneDribbleAMP>& s) { s.d_target_reset(); }
template <class S> void draw_task_update(S&, double) {}
template <> void draw_task_update(SceneX<cSceneTargetAMP>& s, double dt) { s.d_target_update(dt); }
template <> void draw_task_update(SceneX<cSceneHeadingAMP>& s, double dt) { s.d_target_update(dt); }
template <> void draw_task_update(SceneX<cSceneStrikeAMP>& s, double dt) { s.d_target_update(dt); }
template <> void draw_task_update(SceneX<cSceneHeadingAMPGetup>& s, double dt) { s.d_target_update(dt); }
template <> void draw_task_update(SceneX<cSceneDribbleAMP>& s, double dt) { s.d_target_update(dt); }
template <class S> void draw_reset_scene(S& s) {
    // cRLSceneSimChar::ResetScene (RLSceneSimChar.cpp:240-244) = cRLScene::ResetScene + cSceneSimChar::ResetScene; of the latter (SceneSimChar.cpp:628-644):
    // cScene::ResetScene, ResetRandPertrub, [ResetWorld: Bullet], ResetCharacters, [ResetGround, CleanObjs, InitCharacterPos, ResolveCharGroundIntersect: no draws]
    s.d_rl_reset_scene();
    s.d_base_reset_scene();
    if (s.d_perturbs()) s.d_reset_perturb();
    s.d_reset_characters();
    s.d_init_char_pos();                              // (ResolveCharGroundIntersect lifts the character off the ground: heights only, and the parts' AABBs are Bullet's;
    s.d_sync_kin_root();                              //  cSceneImitate's override then moves the kinematic character's root onto the simulated one)
}
}  // namespace

extern "C" {

void* ref3_open(int kind, long seed, const char** tokens, int ntok, const char* char_file, const char* ctrl_file, const char* motion_file, int clips_ctrl,
                const char* cwd, int test_mode, const int* getup_ids, int n_getup) {
    char old_cwd[4096]; if (!getcwd(old_cwd, sizeof(old_cwd))) return nullptr;
    if (cwd && cwd[0] && chdir(cwd) != 0) return nullptr;                      // the dataset file names its motion files relative to the reference's root
    Draw* d = new Draw(); d->kind = kind; d->clips = clips_ctrl != 0;
    for (int i = 0; i < n_getup; ++i) d->getup_ids.push_back(getup_ids[i]);
    cMathUtil::SeedRand((unsigned long)seed);                                   // cDeepMimicCore::SeedRand (DeepMimicCore.cpp:20-23)
    const double g[3] = {0, -9.8, 0};
    Rig* r = (Rig*)ref2_create(char_file, ctrl_file, "", g);
    if (!r) { delete d; (void)!chdir(old_cwd); return nullptr; }
    d->rig = r;
    // the scene object: its constructor seeds mRand from gRand (scenes/Scene.cpp:5)
    switch (kind) {
    case 0: r->amp = std::shared_ptr<SceneX<cSceneImitateAMP>>(new SceneX<cSceneImitateAMP>()); break;
    case 1: r->target = std::shared_ptr<SceneX<cSceneTargetAMP>>(new SceneX<cSceneTargetAMP>()); break;
    case 2: r->heading = std::shared_ptr<SceneX<cSceneHeadingAMP>>(new SceneX<cSceneHeadingAMP>()); break;
    case 3: r->getup = std::shared_ptr<SceneX<cSceneHeadingAMPGetup>>(new SceneX<cSceneHeadingAMPGetup>()); break;
    case 4: r->strike = std::shared_ptr<SceneX<cSceneStrikeAMP>>(new SceneX<cSceneStrikeAMP>()); break;
    case 5: r->dribble = std::shared_ptr<SceneX<cSceneDribbleAMP>>(new SceneX<cSceneDribbleAMP>()); break;
    default: r->imitate = std::shared_ptr<SceneX<cSceneImitate>>(new SceneX<cSceneImitate>()); break;
    }
    std::vector<std::string> args; for (int i = 0; i < ntok; ++i) args.push_back(tokens[i]);
    std::shared_ptr<cArgParser> parser(new cArgParser(args));
    with_scene(d, [&](auto& s) { s.d_parse(parser); s.d_mode(test_mode); });
    // cSceneImitate::Init (scenes/SceneImitate.cpp:153-161): the kinematic character and its controller first -- a cClipsController selects its first clip
    r->kin = std::shared_ptr<cKinCharacter>(new cKinCharacter());
    cKinCharacter::tParams kp; kp.mCharFile = char_file; kp.mLoadDrawShapes = false;
    bool ok = r->kin->Init(kp);
    if (ok && d->clips) { d->clips_ctrl = std::shared_ptr<cClipsController>(new cClipsController()); d->clips_ctrl->Init(r->kin.get(), motion_file); r->kin->SetController(d->clips_ctrl); }
    else if (ok) { std::shared_ptr<cMotionController> mc(new cMotionController()); mc->Init(r->kin.get(), motion_file); r->kin->SetController(mc); }
    (void)!chdir(old_cwd);
    if (!ok) { delete d; return nullptr; }
    if (kind == 5) r->ball = std::shared_ptr<StandinBody>(new StandinBody());
    r->ctrl->SetGround(std::shared_ptr<cGround>(new StandinGround(0.0)));
    r->ctrl->SetCyclePeriod(r->kin->GetMotionDuration());                       // cSceneImitate::BuildController (SceneImitate.cpp:250-264)
    with_scene(d, [&](auto& s) {
        s.d_rl_scene_init();                                                    // cRLSceneSimChar::Init (RLSceneSimChar.cpp:27-37): cRLScene::Init ...
        s.d_scene_init();                                                       // ... cSceneSimChar::Init (SceneSimChar.cpp:106-123): cScene::Init,
        if (s.d_perturbs()) s.d_reset_perturb();                                // ResetRandPertrub,
        cMathUtil::RandUint();                                                  // BuildGround: cGround::cGround seeds the terrain generator (sim/Ground.cpp:68); the ground here is the stand-in
        s.setup(r->ch, r->kin, 0.0);
        s.d_init_char_pos();
        s.d_sync_kin_root();
        s.d_setup_annealer();
    });
    if (kind == 5) r->dribble->d_dribble_ball(r->ball);
    with_scene(d, [&](auto& s) { draw_task_init(s, d); });
    return d;
}
void ref3_close(void* h) { Draw* d = (Draw*)h; new std::shared_ptr<cClipsController>(d->clips_ctrl); ref2_destroy(d->rig); delete d; }
// state of the simulated character as the scene reads it (root position, link positions / velocities, the fall test of CheckTerminate)
void ref3_set_char(void* h, const double* pose, const double* vel, int fallen) { Draw* d = (Draw*)h; ref2_set_state(d->rig, pose, vel); d->rig->ch->fallen = fallen != 0; }
void ref3_set_ball(void* h, const double* pos3) { Draw* d = (Draw*)h; if (d->rig->ball) d->rig->ball->pos = tVector(pos3[0], pos3[1], pos3[2], 0); }
// where the scene's SyncKinCharRoot left the kinematic character's origin (it follows the simulated root, which ResolveCharGroundIntersect lifted: Bullet-side AABBs)
void ref3_set_kin_origin_pos(void* h, const double* p3) { ((Draw*)h)->rig->kin->SetOriginPos(tVector(p3[0], p3[1], p3[2], 0)); }
// contact flags of the body parts (bit j: part j touches something; cSimCharacter::IsInContact), for the scenes' fall logic
void ref3_set_contacts(void* h, int mask) { ((Draw*)h)->rig->ch->contact_mask = mask; }
// cSceneImitate::UpdateKinChar (scenes/SceneImitate.cpp:306-318): the kinematic character's clock, and the root sync at a cycle boundary
void ref3_update_kin(void* h, double dt) { with_scene((Draw*)h, [&](auto& s) { s.d_update_kin(dt); }); }
// CheckTerminate(0) and IsEpisodeEnd() of the scene class (virtual: the task scenes' success / failure conditions, the episode and motion clocks)
void ref3_flags(void* h, int* out2) { with_scene((Draw*)h, [&](auto& s) { out2[0] = (int)s.CheckTerminate(0); out2[1] = s.IsEpisodeEnd() ? 1 : 0; }); }
void ref3_set_sample_count(void* h, int n) { with_scene((Draw*)h, [&](auto& s) { s.SetSampleCount(n); }); }
void ref3_set_mode(void* h, int test) { with_scene((Draw*)h, [&](auto& s) { s.SetMode(test ? cRLScene::eModeTest : cRLScene::eModeTrain); }); }
// cScene::Reset of the scene class.  Returns 1 when heading_amp_getup continued the episode as a recovery episode.
int ref3_reset(void* h) {
    Draw* d = (Draw*)h; int rec = 0;
    if (d->kind == 3) {                                                          // cSceneHeadingAMPGetup::Reset (:105-117)
        auto& s = *d->rig->getup;
        if (s.d_getup_activate_recovery()) { s.d_getup_reset_recovery(); return 1; }
    }
    if (d->kind == 5) d->rig->dribble->d_dribble_reset_head();                   // cSceneDribbleAMP::Reset (:162-169): the ball first
    with_scene(d, [&](auto& s) { draw_reset_scene(s); if (d->kind == 0) amp_init_hist(s); draw_task_reset(s); });      // (InitHist: cSceneImitateAMP::Reset only; cSceneTargetAMP::Reset calls cSceneImitate::Reset, SceneTargetAMP.cpp:125-130)
    return rec;
}
// the drawing part of one scene update: cScene::Update (timers), UpdateRandPerturb, [the world and character update], cSceneDribbleAMP::UpdateObjs,
// cSceneTargetAMP::Update's target part.  Call ref3_set_char with the state AFTER the update first (the target code reads it).
void ref3_update(void* h, double dt) {
    Draw* d = (Draw*)h;
    with_scene(d, [&](auto& s) { s.d_update_timers(dt); if (s.d_perturbs()) s.d_update_perturb(dt); });
    if (d->kind == 5) d->rig->dribble->d_dribble_update_objs(dt);
    with_scene(d, [&](auto& s) { draw_task_update(s, dt); });
    if (d->kind == 3) d->rig->getup->d_getup_test_update();
}
// cSceneImitateAMP::RecordAMPObsExpert (:115-138) itself: SampleExpertMotion draws the clip (gRand, clips controller), then the clip time (mRand)
int ref3_expert(void* h, double* out) { VecX v; int n = 0; with_scene((Draw*)h, [&](auto& s) { n = amp_expert(s, v); }); vout(v, out); return n; }
// the controller of the session's character: cCtPDController::ApplyAction at an action boundary, and the stable-PD torque of the reference's own cImpPDController /
// cRBDModel for the stand-in's current state and the latched targets (ref2_apply_action / ref2_spd_tau on the session's rig): what cCtController::Update computes
// before the world steps
void ref3_apply_action(void* h, const double* action, double* out_tar) { ref2_apply_action(((Draw*)h)->rig, action, out_tar); }
void ref3_spd_tau(void* h, double dt, double* out_tau) { ref2_spd_tau(((Draw*)h)->rig, dt, out_tau); }
// the learner-facing tables of the SCENE (cRLScene's pure virtuals, scenes/RLScene.h:43-66, as each scene class answers them -- dribble_amp appends its task state to the
// controller's): which: 0 state offset, 1 state scale, 2 state norm groups, 3 goal offset, 4 goal scale, 5 goal norm groups, 6 action offset, 7 action scale, 8 action
// bound min, 9 action bound max, 10 AMP obs offset, 11 AMP obs scale, 12 AMP obs norm group, 13 {reward min, max, fail, succ}.  Returns the length written.
int ref3_tables(void* h, int which, double* out) {
    VecX a, b; Eigen::VectorXi g; int n = 0;
    with_scene((Draw*)h, [&](auto& s) {
        switch (which) {
        case 0: case 1: s.BuildStateOffsetScale(0, a, b); if (which == 1) a = b; break;
        case 2: s.BuildStateNormGroups(0, g); a = g.cast<double>(); break;
        case 3: case 4: s.BuildGoalOffsetScale(0, a, b); if (which == 4) a = b; break;
        case 5: s.BuildGoalNormGroups(0, g); a = g.cast<double>(); break;
        case 6: case 7: s.BuildActionOffsetScale(0, a, b); if (which == 7) a = b; break;
        case 8: case 9: s.BuildActionBounds(0, a, b); if (which == 9) a = b; break;
        case 13: a.resize(4); a[0] = s.GetRewardMin(0); a[1] = s.GetRewardMax(0); a[2] = s.GetRewardFail(0); a[3] = s.GetRewardSucc(0); break;
        default: amp_tables(s, which, a); break;
        }
    });
    n = (int)a.size(); vout(a, out);
    return n;
}
// cCtController::CheckNeedNewAction (sim/CtController.cpp:221-227) at the caller's controller clock: the 30 Hz latch with the init-time offset the scene's own
// SyncCharacters gave the controller at the reset (cCtController::SetInitTime, scenes/SceneImitate.cpp:351-368)
int ref3_need_new_action(void* h, double ctrl_time, double dt) { Rig* r = ((Draw*)h)->rig; r->ctrl->set_time(ctrl_time); return r->ctrl->CheckNeedNewAction(dt) ? 1 : 0; }
// cSceneImitateAMP::RecordAMPObsAgent (:101-113) on the stand-in character with the history the session's own NewActionUpdate latched
int ref3_amp_agent(void* h, double* out) { VecX v; int n = 0; with_scene((Draw*)h, [&](auto& s) { n = amp_agent(s, v); }); vout(v, out); return n; }
// cSceneImitateAMP::InitHist (:153-165) again, after the caller put the kinematic origin where the reset's ground-intersection lift (Bullet-side) left it
void ref3_init_hist(void* h) { with_scene((Draw*)h, [&](auto& s) { amp_init_hist(s); }); }
// imitate_amp only (the task scenes' Init / Reset bypass cSceneImitateAMP's: no time warper there): build = 1 once after ref3_open, 0 after every reset -- with the
// stand-in character and the kinematic origin where the device's are (the warper samples both characters at once)
void ref3_time_warper(void* h, int build) { Draw* d = (Draw*)h; if (d->kind == 0) with_scene(d, [&](auto& s) { amp_time_warper(s, build != 0); }); }
void ref3_new_action(void* h) { with_scene((Draw*)h, [&](auto& s) { amp_new_action(s); }); }
// cCtController::RecordState (sim/CtController.cpp:281-293) of the scene's controller at the caller's controller clock (the clock advances inside the simulated update)
int ref3_record_state(void* h, double ctrl_time, double* out) {
    Rig* r = ((Draw*)h)->rig;
    r->ctrl->set_time(ctrl_time);
    VecX s;
    with_scene((Draw*)h, [&](auto& sc) { sc.RecordState(0, s); });              // the scene's RecordState: the controller's, + the task state of dribble_amp (SceneDribbleAMP.cpp:193-210)
    vout(s, out);
    return (int)s.size();
}
// the kinematic character's pose and velocity (cKinCharacter::GetPose / GetVel): what the imitation reward and the root sync compare the simulated character with
int ref3_kin_pose(void* h, double* pose, double* vel) { Rig* r = ((Draw*)h)->rig; vout(r->kin->GetPose(), pose); vout(r->kin->GetVel(), vel); return (int)r->kin->GetPose().size(); }
// CalcReward(0) and RecordGoal(0) of the scene class at an action boundary.  The controller's bookkeeping of the last action (cDeepMimicCharController::mPrevActionTime /
// mPrevActionCOM, written by HandleNewAction inside the simulated update) and its clock come from the caller, as does the ball's record at the last action (dribble_amp:
// cSceneDribbleAMP::NewActionUpdate); everything else is the session's own state -- target, heading, speed, hit time, scene clock.  out: reward, goal...; returns the goal size.
int ref3_reward_goal(void* h, double ctrl_time, double prev_action_time, const double* prev_action_com3, const double* prev_ball3, double* out) {
    Draw* d = (Draw*)h; Rig* r = d->rig;
    r->ctrl->set_time(ctrl_time); r->ctrl->set_prev_action(prev_action_time, tVector(prev_action_com3[0], prev_action_com3[1], prev_action_com3[2], 0));
    if (d->kind == 5 && prev_ball3) r->dribble->d_dribble_prev_ball(tVector(prev_ball3[0], prev_ball3[1], prev_ball3[2], 0));
    VecX g; double rew = 0;
    with_scene(d, [&](auto& s) { rew = s.CalcReward(0); s.RecordGoal(0, g); });
    out[0] = rew; for (int i = 0; i < (int)g.size(); ++i) out[1 + i] = g[i];
    return (int)g.size();
}
void ref3_set_ball_full(void* h, const double* s13) {
    Draw* d = (Draw*)h; if (!d->rig->ball) return;
    StandinBody& b = *d->rig->ball;
    b.pos = tVector(s13[0], s13[1], s13[2], 0); b.rot = tQuaternion(s13[3], s13[4], s13[5], s13[6]); b.lin = tVector(s13[7], s13[8], s13[9], 0); b.ang = tVector(s13[10], s13[11], s13[12], 0);
}
void ref3_get(void* h, double* out) {
    Draw* d = (Draw*)h; Rig* r = d->rig;
    for (int i = 0; i < 48; ++i) out[i] = 0;
    { const tVector op = r->kin->GetOriginPos(); out[40] = op[0]; out[41] = op[1]; out[42] = op[2]; out[43] = r->kin->GetPhase(); out[44] = r->kin->GetCycle(); out[45] = r->kin->IsMotionOver() ? 1 : 0; }
    out[1] = r->kin->GetTime(); out[2] = d->clips_ctrl ? d->clips_ctrl->GetCurrMotionID() : 0;
    { const tVector rp = r->ch->GetRootPos(), kp = r->kin->GetRootPos(); out[35] = rp[0]; out[36] = rp[1]; out[37] = rp[2]; out[38] = kp[0]; out[39] = kp[2]; }
    const tQuaternion q = r->kin->GetOriginRot(); out[3] = q.w(); out[4] = q.x(); out[5] = q.y(); out[6] = q.z();
    with_scene(d, [&](auto& s) {
        out[0] = s.d_timer_max(); out[34] = s.d_time(); out[14] = s.d_pert_next(); out[15] = s.d_pert_timer(); out[31] = s.n_perturbs;
        if (s.n_perturbs) {
            const cSimBodyLink* lk = dynamic_cast<const cSimBodyLink*>(s.pert_obj);
            out[16] = lk ? lk->GetJointID() : -1; out[17] = s.pert_force[0]; out[18] = s.pert_force[1]; out[19] = s.pert_force[2]; out[20] = s.pert_dur;
        }
    });
    auto task = [&](auto& s) { const tVector t = s.d_target_pos(); out[7] = t[0]; out[8] = t[1]; out[9] = t[2]; out[11] = s.d_target_speed(); out[12] = s.d_target_timer_max(); out[13] = s.d_target_timer_time(); };
    switch (d->kind) {
    case 1: task(*r->target); break;
    case 2: task(*r->heading); out[10] = r->heading->d_heading(); break;
    case 3: task(*r->getup); out[10] = r->getup->d_heading(); out[32] = r->getup->d_getup_timer(); break;
    case 4: task(*r->strike); out[21] = r->strike->d_strike_hit(); out[22] = r->strike->d_strike_hit_time(); break;
    case 5: task(*r->dribble); out[30] = r->dribble->d_obj_timer_max();
            out[23] = r->ball->pos[0]; out[24] = r->ball->pos[1]; out[25] = r->ball->pos[2]; out[26] = r->ball->rot.w(); out[27] = r->ball->rot.x(); out[28] = r->ball->rot.y(); out[29] = r->ball->rot.z(); break;
    default: break;
    }
}

}  // extern "C"
