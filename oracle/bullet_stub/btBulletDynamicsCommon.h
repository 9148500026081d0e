// TEST INFRASTRUCTURE (like oracle/gl_stub): TYPE NAMES of Bullet 2.88 so that the reference's own headers (sim/World.h, sim/SimObj.h,
// sim/SimCharacter.h, ...) parse and the reference routines whose translation units include them -- sim/ImpPDController.cpp,
// sim/CtController.cpp, scenes/SceneImitate.cpp, ... -- compile UNMODIFIED into oracle/_ref/libdm_ref.so.  Nothing here computes
// anything: Bullet (un-vendored, absent from this image) is not restated.  The Bullet-backed getters those routines call are
// link-time stand-ins in oracle/ref_standins.cpp.
#pragma once
#include <vector>
typedef float btScalar;
struct btVector3 { btScalar m[4]; btVector3() : m{0, 0, 0, 0} {} btVector3(btScalar x, btScalar y, btScalar z) : m{x, y, z, 0} {} btScalar& operator[](int i) { return m[i]; } const btScalar& operator[](int i) const { return m[i]; } };
struct btQuaternion { btScalar m[4]; btQuaternion() : m{0, 0, 0, 1} {} };
struct btMatrix3x3 { btVector3 r[3]; };
struct btTransform { btMatrix3x3 basis; btVector3 origin; };
template <class T> class btAlignedObjectArray { public: std::vector<T> v; int size() const { return (int)v.size(); } void resize(int n) { v.resize(n); } T& operator[](int i) { return v[i]; } const T& operator[](int i) const { return v[i]; } };
class btCollisionShape { public: virtual ~btCollisionShape() {} };
class btBoxShape : public btCollisionShape {}; class btCapsuleShape : public btCollisionShape {}; class btStaticPlaneShape : public btCollisionShape {};
class btSphereShape : public btCollisionShape {}; class btCylinderShape : public btCollisionShape {};
class btCollisionObject { public: virtual ~btCollisionObject() {} };
class btTypedConstraint { public: virtual ~btTypedConstraint() {} };
class btConstraintSolver { public: virtual ~btConstraintSolver() {} };
class btCollisionDispatcher { public: virtual ~btCollisionDispatcher() {} };
class btDefaultCollisionConfiguration { public: virtual ~btDefaultCollisionConfiguration() {} };
class btBroadphaseInterface { public: virtual ~btBroadphaseInterface() {} };
class btRigidBody : public btCollisionObject {};
class btMotionState { public: virtual ~btMotionState() {} };
class btDefaultMotionState : public btMotionState {};
class btPersistentManifold;
