// Drop-in header: wheel joint definition (reference: Box2D/Dynamics/Joints/b2WheelJoint.h:31-211).
// bodyB's anchor stays on a line fixed in bodyA; a spring acts along the line and a motor turns bodyB.
// Solved on the device (csrc/b2d_joint.h); this class keeps the definition and forwards the motor setters.
#ifndef B2_WHEEL_JOINT_H
#define B2_WHEEL_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2WheelJointDef : public b2JointDef
{
	b2WheelJointDef()
	{
		type = e_wheelJoint;
		localAnchorA.SetZero();
		localAnchorB.SetZero();
		localAxisA.Set(1.0f, 0.0f);
		enableMotor = false;
		maxMotorTorque = 0.0f;
		motorSpeed = 0.0f;
		frequencyHz = 2.0f;
		dampingRatio = 0.7f;
	}
	void Initialize(b2Body* bodyA, b2Body* bodyB, const b2Vec2& anchor, const b2Vec2& axis);

	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	b2Vec2 localAxisA;
	bool enableMotor;
	float32 maxMotorTorque;
	float32 motorSpeed;
	float32 frequencyHz;
	float32 dampingRatio;
};

class b2WheelJoint : public b2Joint
{
public:
	const b2Vec2& GetLocalAnchorA() const { return m_localAnchorA; }
	const b2Vec2& GetLocalAnchorB() const { return m_localAnchorB; }
	const b2Vec2& GetLocalAxisA() const { return m_localAxisA; }
	bool IsMotorEnabled() const { return m_enableMotor; }
	float32 GetMotorTorque(float32 inv_dt) const { return MotorReaction(inv_dt); }
	float32 GetMotorSpeed() const { return m_motorSpeed; }
	float32 GetMaxMotorTorque() const { return m_maxMotorTorque; }
	float32 GetSpringFrequencyHz() const { return m_frequencyHz; }
	/// b2WheelJoint.cpp:350-397: from the bodies' current states
	float32 GetJointTranslation() const;
	float32 GetJointLinearSpeed() const;
	float32 GetJointAngle() const;
	float32 GetJointAngularSpeed() const;
	float32 GetSpringDampingRatio() const { return m_dampingRatio; }
	void EnableMotor(bool flag);
	void SetMotorSpeed(float32 speed);
	void SetMaxMotorTorque(float32 torque);
	void SetSpringFrequencyHz(float32 hz);
	void SetSpringDampingRatio(float32 ratio);

protected:
	friend class b2World;
	void PushMotor();
	b2WheelJoint(const b2WheelJointDef* def) : b2Joint(def), m_localAnchorA(def->localAnchorA), m_localAnchorB(def->localAnchorB),
		m_localAxisA(def->localAxisA), m_enableMotor(def->enableMotor), m_maxMotorTorque(def->maxMotorTorque),
		m_motorSpeed(def->motorSpeed), m_frequencyHz(def->frequencyHz), m_dampingRatio(def->dampingRatio) {}

	b2Vec2 m_localAnchorA, m_localAnchorB, m_localAxisA;
	bool m_enableMotor;
	float32 m_maxMotorTorque, m_motorSpeed, m_frequencyHz, m_dampingRatio;
};

#endif
