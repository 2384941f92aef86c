// Drop-in header: rope joint definition (reference: Box2D/Dynamics/Joints/b2RopeJoint.h:28-117).
// Caps the distance between two anchor points. Solved on the device (csrc/b2d_joint.h).
#ifndef B2_ROPE_JOINT_H
#define B2_ROPE_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2RopeJointDef : public b2JointDef
{
	b2RopeJointDef()
	{
		type = e_ropeJoint;
		localAnchorA.Set(-1.0f, 0.0f);
		localAnchorB.Set(1.0f, 0.0f);
		maxLength = 0.0f;
	}
	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	float32 maxLength;
};

class b2RopeJoint : public b2Joint
{
public:
	const b2Vec2& GetLocalAnchorA() const { return m_localAnchorA; }
	const b2Vec2& GetLocalAnchorB() const { return m_localAnchorB; }
	float32 GetMaxLength() const { return m_maxLength; }
	void SetMaxLength(float32 length) ; // b2RopeJoint.h:80
	b2LimitState GetLimitState() const; // b2RopeJoint.h:84: e_atUpperLimit while the rope is taut (the last step's solver state)

protected:
	friend class b2World;
	b2RopeJoint(const b2RopeJointDef* def) : b2Joint(def), m_localAnchorA(def->localAnchorA), m_localAnchorB(def->localAnchorB),
		m_maxLength(def->maxLength) {}

	b2Vec2 m_localAnchorA, m_localAnchorB;
	float32 m_maxLength;
};

#endif
