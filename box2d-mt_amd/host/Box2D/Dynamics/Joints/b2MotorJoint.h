// Drop-in header: motor joint definition (reference: Box2D/Dynamics/Joints/b2MotorJoint.h:26-133).
// Drives bodyB towards a pose given in bodyA's frame with capped force and torque. Solved on the device
// (csrc/b2d_joint.h); the offset setters forward to the device record.
#ifndef B2_MOTOR_JOINT_H
#define B2_MOTOR_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2MotorJointDef : public b2JointDef
{
	b2MotorJointDef()
	{
		type = e_motorJoint;
		linearOffset.SetZero();
		angularOffset = 0.0f;
		maxForce = 1.0f;
		maxTorque = 1.0f;
		correctionFactor = 0.3f;
	}
	void Initialize(b2Body* bodyA, b2Body* bodyB);

	b2Vec2 linearOffset;
	float32 angularOffset;
	float32 maxForce;
	float32 maxTorque;
	float32 correctionFactor;
};

class b2MotorJoint : public b2Joint
{
public:
	const b2Vec2& GetLinearOffset() const { return m_linearOffset; }
	float32 GetAngularOffset() const { return m_angularOffset; }
	float32 GetMaxForce() const { return m_maxForce; }
	float32 GetMaxTorque() const { return m_maxTorque; }
	float32 GetCorrectionFactor() const { return m_correctionFactor; }
	void SetMaxForce(float32 force);   // b2MotorJoint.cpp:222-251
	void SetMaxTorque(float32 torque);
	void SetCorrectionFactor(float32 factor);
	void SetLinearOffset(const b2Vec2& linearOffset);
	void SetAngularOffset(float32 angularOffset);

protected:
	friend class b2World;
	b2MotorJoint(const b2MotorJointDef* def) : b2Joint(def), m_linearOffset(def->linearOffset), m_angularOffset(def->angularOffset),
		m_maxForce(def->maxForce), m_maxTorque(def->maxTorque), m_correctionFactor(def->correctionFactor) {}

	b2Vec2 m_linearOffset;
	float32 m_angularOffset, m_maxForce, m_maxTorque, m_correctionFactor;
};

#endif
