// Drop-in header: revolute joint definition (reference: Box2D/Dynamics/Joints/b2RevoluteJoint.h:35-204).
#ifndef B2_REVOLUTE_JOINT_H
#define B2_REVOLUTE_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2RevoluteJointDef : public b2JointDef
{
	b2RevoluteJointDef()
	{
		type = e_revoluteJoint;
		localAnchorA.Set(0.0f, 0.0f);
		localAnchorB.Set(0.0f, 0.0f);
		referenceAngle = 0.0f;
		lowerAngle = 0.0f;
		upperAngle = 0.0f;
		maxMotorTorque = 0.0f;
		motorSpeed = 0.0f;
		enableLimit = false;
		enableMotor = false;
	}
	void Initialize(b2Body* bodyA, b2Body* bodyB, const b2Vec2& anchor);

	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	float32 referenceAngle;
	bool enableLimit;
	float32 lowerAngle;
	float32 upperAngle;
	bool enableMotor;
	float32 motorSpeed;
	float32 maxMotorTorque;
};

class b2RevoluteJoint : public b2Joint
{
public:
	const b2Vec2& GetLocalAnchorA() const { return m_localAnchorA; }
	const b2Vec2& GetLocalAnchorB() const { return m_localAnchorB; }
	float32 GetReferenceAngle() const { return m_referenceAngle; }
	bool IsMotorEnabled() const { return m_enableMotor; }
	float32 GetMotorTorque(float32 inv_dt) const { return MotorReaction(inv_dt); }
	float32 GetMotorSpeed() const { return m_motorSpeed; }
	float32 GetMaxMotorTorque() const { return m_maxMotorTorque; }
	bool IsLimitEnabled() const { return m_enableLimit; }
	float32 GetLowerLimit() const { return m_lowerAngle; }
	float32 GetUpperLimit() const { return m_upperAngle; }
	// computed from the body states of the last step (reference: b2RevoluteJoint.cpp:399-411)
	float32 GetJointAngle() const;
	float32 GetJointSpeed() const;
	// setters forward to the device record; like the reference's they wake both bodies when something changes
	void EnableLimit(bool flag);
	void SetLimits(float32 lower, float32 upper);
	void EnableMotor(bool flag);
	void SetMotorSpeed(float32 speed);
	void SetMaxMotorTorque(float32 torque);

protected:
	friend class b2World;
	void PushMotor();
	void PushLimits();
	b2RevoluteJoint(const b2RevoluteJointDef* def) : b2Joint(def), m_localAnchorA(def->localAnchorA),
		m_localAnchorB(def->localAnchorB), m_referenceAngle(def->referenceAngle), m_enableLimit(def->enableLimit),
		m_lowerAngle(def->lowerAngle), m_upperAngle(def->upperAngle), m_enableMotor(def->enableMotor),
		m_motorSpeed(def->motorSpeed), m_maxMotorTorque(def->maxMotorTorque) {}

	b2Vec2 m_localAnchorA, m_localAnchorB;
	float32 m_referenceAngle;
	bool m_enableLimit;
	float32 m_lowerAngle, m_upperAngle;
	bool m_enableMotor;
	float32 m_motorSpeed, m_maxMotorTorque;
};

#endif
