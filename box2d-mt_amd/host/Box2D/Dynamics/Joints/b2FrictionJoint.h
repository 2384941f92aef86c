// Drop-in header: friction joint definition (reference: Box2D/Dynamics/Joints/b2FrictionJoint.h:26-117).
// Top-down friction between two bodies with capped force and torque. Solved on the device (csrc/b2d_joint.h).
#ifndef B2_FRICTION_JOINT_H
#define B2_FRICTION_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2FrictionJointDef : public b2JointDef
{
	b2FrictionJointDef()
	{
		type = e_frictionJoint;
		localAnchorA.SetZero();
		localAnchorB.SetZero();
		maxForce = 0.0f;
		maxTorque = 0.0f;
	}
	void Initialize(b2Body* bodyA, b2Body* bodyB, const b2Vec2& anchor);

	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	float32 maxForce;
	float32 maxTorque;
};

class b2FrictionJoint : public b2Joint
{
public:
	const b2Vec2& GetLocalAnchorA() const { return m_localAnchorA; }
	const b2Vec2& GetLocalAnchorB() const { return m_localAnchorB; }
	float32 GetMaxForce() const { return m_maxForce; }
	float32 GetMaxTorque() const { return m_maxTorque; }
	void SetMaxForce(float32 force);   // b2FrictionJoint.cpp:206-228
	void SetMaxTorque(float32 torque);

protected:
	friend class b2World;
	b2FrictionJoint(const b2FrictionJointDef* def) : b2Joint(def), m_localAnchorA(def->localAnchorA),
		m_localAnchorB(def->localAnchorB), m_maxForce(def->maxForce), m_maxTorque(def->maxTorque) {}

	b2Vec2 m_localAnchorA, m_localAnchorB;
	float32 m_maxForce, m_maxTorque;
};

#endif
