// Drop-in header: weld joint definition (reference: Box2D/Dynamics/Joints/b2WeldJoint.h:28-123).
// Glues two bodies together at an anchor; frequencyHz > 0 softens the angular part into a damped spring.
// Solved on the device (csrc/b2d_joint.h); this class only keeps the definition for the accessors.
#ifndef B2_WELD_JOINT_H
#define B2_WELD_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2WeldJointDef : public b2JointDef
{
	b2WeldJointDef()
	{
		type = e_weldJoint;
		localAnchorA.Set(0.0f, 0.0f);
		localAnchorB.Set(0.0f, 0.0f);
		referenceAngle = 0.0f;
		frequencyHz = 0.0f;
		dampingRatio = 0.0f;
	}
	void Initialize(b2Body* bodyA, b2Body* bodyB, const b2Vec2& anchor);

	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	float32 referenceAngle;
	float32 frequencyHz;
	float32 dampingRatio;
};

class b2WeldJoint : public b2Joint
{
public:
	const b2Vec2& GetLocalAnchorA() const { return m_localAnchorA; }
	const b2Vec2& GetLocalAnchorB() const { return m_localAnchorB; }
	float32 GetReferenceAngle() const { return m_referenceAngle; }
	float32 GetFrequency() const { return m_frequencyHz; }
	float32 GetDampingRatio() const { return m_dampingRatio; }
	void SetFrequency(float32 hz);     // b2WeldJoint.h:80-86
	void SetDampingRatio(float32 ratio);

protected:
	friend class b2World;
	b2WeldJoint(const b2WeldJointDef* def) : b2Joint(def), m_localAnchorA(def->localAnchorA), m_localAnchorB(def->localAnchorB),
		m_referenceAngle(def->referenceAngle), m_frequencyHz(def->frequencyHz), m_dampingRatio(def->dampingRatio) {}

	b2Vec2 m_localAnchorA, m_localAnchorB;
	float32 m_referenceAngle, m_frequencyHz, m_dampingRatio;
};

#endif
