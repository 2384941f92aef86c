// Drop-in header: joint base types (reference: Box2D/Dynamics/Joints/b2Joint.h:31-232).
// Device path: all eleven joint types (revolute, distance, prismatic, weld, wheel, rope, friction, motor, pulley, mouse, gear).
#ifndef B2_JOINT_H
#define B2_JOINT_H

#include "Box2D/Common/b2Math.h"

class b2Body;
class b2Joint;
class b2World;

enum b2JointType
{
	e_unknownJoint,
	e_revoluteJoint,
	e_prismaticJoint,
	e_distanceJoint,
	e_pulleyJoint,
	e_mouseJoint,
	e_gearJoint,
	e_wheelJoint,
	e_weldJoint,
	e_frictionJoint,
	e_ropeJoint,
	e_motorJoint
};

enum b2LimitState { e_inactiveLimit, e_atLowerLimit, e_atUpperLimit, e_equalLimits };

struct b2JointEdge
{
	b2Body* other;
	b2Joint* joint;
	b2JointEdge* prev;
	b2JointEdge* next;
};

struct b2JointDef
{
	b2JointDef()
	{
		type = e_unknownJoint;
		userData = nullptr;
		bodyA = nullptr;
		bodyB = nullptr;
		collideConnected = false;
	}
	b2JointType type;
	void* userData;
	b2Body* bodyA;
	b2Body* bodyB;
	bool collideConnected;
};

class b2Joint
{
public:
	virtual ~b2Joint() {}
	b2JointType GetType() const { return m_type; }
	b2Body* GetBodyA() { return m_bodyA; }
	b2Body* GetBodyB() { return m_bodyB; }
	b2Joint* GetNext() { return m_next; }
	const b2Joint* GetNext() const { return m_next; }
	void* GetUserData() const { return m_userData; }
	void SetUserData(void* data) { m_userData = data; }
	bool GetCollideConnected() const { return m_collideConnected; }
	int32 GetDeviceId() const { return m_id; }
	/// The force / torque the joint put on bodyB in the last step, for every joint type (b2Joint.h:129-133): computed from the
	/// device record's accumulated impulses (b2hip_get_joint_reaction).
	b2Vec2 GetReactionForce(float32 inv_dt) const;
	float32 GetReactionTorque(float32 inv_dt) const;
	bool IsActive() const;
	/// The anchor points in world coordinates (pure virtual in the reference, one pair per joint type: the bodies' local
	/// anchors; the mouse joint's target and its hold on bodyB, b2MouseJoint.cpp:200-208; the body origins for the motor
	/// joint, b2MotorJoint.cpp:212-220; the second anchors of its two joints for the gear, b2GearJoint.cpp:371-379).
	b2Vec2 GetAnchorA() const;
	b2Vec2 GetAnchorB() const;

protected:
	friend class b2World;
	float32 MotorReaction(float32 inv_dt) const; // the motor's share (GetMotorTorque / GetMotorForce of the types that have one)
	b2Joint(const b2JointDef* def) : m_type(def->type), m_prev(nullptr), m_next(nullptr), m_bodyA(def->bodyA),
		m_bodyB(def->bodyB), m_collideConnected(def->collideConnected), m_userData(def->userData), m_id(-1), m_anchorKind(0)
	{
		m_anchorA.SetZero();
		m_anchorB.SetZero();
	}

	b2JointType m_type;
	b2Joint* m_prev;
	b2Joint* m_next;
	b2Body* m_bodyA;
	b2Body* m_bodyB;
	bool m_collideConnected;
	void* m_userData;
	int32 m_id;
	// GetAnchorA / B: local anchors on the two bodies (kind 0), the mouse joint (1: A is the target), body origins (2)
	b2Vec2 m_anchorA, m_anchorB;
	int32 m_anchorKind;
	// nodes of the two bodies' joint lists (b2Body::GetJointList, b2Joint.h:73-79, b2World.cpp:258-270)
	b2JointEdge m_edgeA, m_edgeB;
};

#endif
