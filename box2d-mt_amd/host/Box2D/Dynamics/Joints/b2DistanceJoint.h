// Drop-in header: distance joint definition (reference: Box2D/Dynamics/Joints/b2DistanceJoint.h:31-166).
// A rigid rod between two anchor points, or a damped spring when frequencyHz > 0. Solved on the device
// (csrc/b2d_joint.h); this class only keeps the definition for the accessors.
#ifndef B2_DISTANCE_JOINT_H
#define B2_DISTANCE_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2DistanceJointDef : public b2JointDef
{
	b2DistanceJointDef()
	{
		type = e_distanceJoint;
		localAnchorA.Set(0.0f, 0.0f);
		localAnchorB.Set(0.0f, 0.0f);
		length = 1.0f;
		frequencyHz = 0.0f;
		dampingRatio = 0.0f;
	}
	// world anchors -> local anchors and rest length
	void Initialize(b2Body* bodyA, b2Body* bodyB, const b2Vec2& anchorA, const b2Vec2& anchorB);

	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	float32 length;
	float32 frequencyHz;
	float32 dampingRatio;
};

class b2DistanceJoint : public b2Joint
{
public:
	const b2Vec2& GetLocalAnchorA() const { return m_localAnchorA; }
	const b2Vec2& GetLocalAnchorB() const { return m_localAnchorB; }
	float32 GetLength() const { return m_length; }
	/// b2DistanceJoint.h:117-131 (plain assignments, read by the next step)
	void SetLength(float32 length);
	void SetFrequency(float32 hz);
	void SetDampingRatio(float32 ratio);
	float32 GetFrequency() const { return m_frequencyHz; }
	float32 GetDampingRatio() const { return m_dampingRatio; }

protected:
	friend class b2World;
	b2DistanceJoint(const b2DistanceJointDef* def) : b2Joint(def), m_localAnchorA(def->localAnchorA),
		m_localAnchorB(def->localAnchorB), m_length(def->length), m_frequencyHz(def->frequencyHz),
		m_dampingRatio(def->dampingRatio) {}

	b2Vec2 m_localAnchorA, m_localAnchorB;
	float32 m_length, m_frequencyHz, m_dampingRatio;
};

#endif
