// Drop-in header: pulley joint definition (reference: Box2D/Dynamics/Joints/b2PulleyJoint.h:28-150).
// Two bodies hang from two fixed ground points on one idealised rope: lengthA + ratio * lengthB is constant.
// Solved on the device (csrc/b2d_joint.h); this class only keeps the definition for the accessors.
#ifndef B2_PULLEY_JOINT_H
#define B2_PULLEY_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

const float32 b2_minPulleyLength = 2.0f;

struct b2PulleyJointDef : public b2JointDef
{
	b2PulleyJointDef()
	{
		type = e_pulleyJoint;
		groundAnchorA.Set(-1.0f, 1.0f);
		groundAnchorB.Set(1.0f, 1.0f);
		localAnchorA.Set(-1.0f, 0.0f);
		localAnchorB.Set(1.0f, 0.0f);
		lengthA = 0.0f;
		lengthB = 0.0f;
		ratio = 1.0f;
		collideConnected = true;
	}
	// world ground anchors and world body anchors -> local anchors and the two rest lengths
	void Initialize(b2Body* bodyA, b2Body* bodyB, const b2Vec2& groundAnchorA, const b2Vec2& groundAnchorB,
		const b2Vec2& anchorA, const b2Vec2& anchorB, float32 ratio);

	b2Vec2 groundAnchorA;
	b2Vec2 groundAnchorB;
	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	float32 lengthA;
	float32 lengthB;
	float32 ratio;
};

class b2PulleyJoint : public b2Joint
{
public:
	b2Vec2 GetGroundAnchorA() const { return m_groundAnchorA; }
	b2Vec2 GetGroundAnchorB() const { return m_groundAnchorB; }
	float32 GetLengthA() const { return m_lengthA; }
	float32 GetLengthB() const { return m_lengthB; }
	float32 GetRatio() const { return m_ratio; }
	/// distance of each body anchor from its ground anchor, from the current poses (b2PulleyJoint.cpp:310-324)
	float32 GetCurrentLengthA() const;
	float32 GetCurrentLengthB() const;

protected:
	friend class b2World;
	b2PulleyJoint(const b2PulleyJointDef* def) : b2Joint(def), m_groundAnchorA(def->groundAnchorA), m_groundAnchorB(def->groundAnchorB),
		m_localAnchorA(def->localAnchorA), m_localAnchorB(def->localAnchorB), m_lengthA(def->lengthA), m_lengthB(def->lengthB),
		m_ratio(def->ratio) {}

	b2Vec2 m_groundAnchorA, m_groundAnchorB, m_localAnchorA, m_localAnchorB;
	float32 m_lengthA, m_lengthB, m_ratio;
};

#endif
