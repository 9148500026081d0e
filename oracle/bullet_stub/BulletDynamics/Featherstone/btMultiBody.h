#pragma once
#include "btBulletDynamicsCommon.h"
class btMultiBody { public: btMultiBody(int, btScalar, const btVector3&, bool, bool, bool = true) {} virtual ~btMultiBody() {} };
