// Drop-in header: mouse joint definition (reference: Box2D/Dynamics/Joints/b2MouseJoint.h:27-126).
// A soft constraint dragging one point of bodyB towards a world target that the user keeps moving (Testbed mouse
// picking). Solved on the device (csrc/b2d_joint.h); SetTarget forwards to the device record.
#ifndef B2_MOUSE_JOINT_H
#define B2_MOUSE_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2MouseJointDef : public b2JointDef
{
	b2MouseJointDef()
	{
		type = e_mouseJoint;
		target.Set(0.0f, 0.0f);
		maxForce = 0.0f;
		frequencyHz = 5.0f;
		dampingRatio = 0.7f;
	}
	b2Vec2 target;
	float32 maxForce;
	float32 frequencyHz;
	float32 dampingRatio;
};

class b2MouseJoint : public b2Joint
{
public:
	void SetTarget(const b2Vec2& target);
	const b2Vec2& GetTarget() const { return m_targetA; }
	float32 GetMaxForce() const { return m_maxForce; }
	float32 GetFrequency() const { return m_frequencyHz; }
	float32 GetDampingRatio() const { return m_dampingRatio; }
	void SetMaxForce(float32 force);   // b2MouseJoint.cpp:48-76
	void SetFrequency(float32 hz);
	void SetDampingRatio(float32 ratio);

protected:
	friend class b2World;
	b2MouseJoint(const b2MouseJointDef* def) : b2Joint(def), m_targetA(def->target), m_maxForce(def->maxForce),
		m_frequencyHz(def->frequencyHz), m_dampingRatio(def->dampingRatio) {}

	b2Vec2 m_targetA;
	float32 m_maxForce, m_frequencyHz, m_dampingRatio;
};

#endif
