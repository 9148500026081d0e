#pragma once
#include "BulletDynamics/Featherstone/btMultiBody.h"
#ifndef DM_BULLET_STUB_FEATHERSTONE
#define DM_BULLET_STUB_FEATHERSTONE
class btMultiBodyConstraint { public: virtual ~btMultiBodyConstraint() {} };
class btMultiBodyLinkCollider : public btCollisionObject {};
class btMultiBodyJointLimitConstraint : public btMultiBodyConstraint {};
class btMultiBodyDynamicsWorld { public: virtual ~btMultiBodyDynamicsWorld() {} };
#endif
