// Drop-in header: prismatic joint definition (reference: Box2D/Dynamics/Joints/b2PrismaticJoint.h:31-196).
// bodyB slides along an axis fixed in bodyA and keeps its relative angle; optional translation limits and a linear
// motor. Solved on the device (csrc/b2d_joint.h); this class only keeps the definition for the accessors.
#ifndef B2_PRISMATIC_JOINT_H
#define B2_PRISMATIC_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2PrismaticJointDef : public b2JointDef
{
	b2PrismaticJointDef()
	{
		type = e_prismaticJoint;
		localAnchorA.SetZero();
		localAnchorB.SetZero();
		localAxisA.Set(1.0f, 0.0f);
		referenceAngle = 0.0f;
		enableLimit = false;
		lowerTranslation = 0.0f;
		upperTranslation = 0.0f;
		enableMotor = false;
		maxMotorForce = 0.0f;
		motorSpeed = 0.0f;
	}
	// world anchor and world axis -> local anchors, local axis and reference angle
	void Initialize(b2Body* bodyA, b2Body* bodyB, const b2Vec2& anchor, const b2Vec2& axis);

	b2Vec2 localAnchorA;
	b2Vec2 localAnchorB;
	b2Vec2 localAxisA;
	float32 referenceAngle;
	bool enableLimit;
	float32 lowerTranslation;
	float32 upperTranslation;
	bool enableMotor;
	float32 maxMotorForce;
	float32 motorSpeed;
};

class b2PrismaticJoint : public b2Joint
{
public:
	const b2Vec2& GetLocalAnchorA() const { return m_localAnchorA; }
	const b2Vec2& GetLocalAnchorB() const { return m_localAnchorB; }
	const b2Vec2& GetLocalAxisA() const { return m_localAxisA; }
	float32 GetReferenceAngle() const { return m_referenceAngle; }
	bool IsLimitEnabled() const { return m_enableLimit; }
	float32 GetLowerLimit() const { return m_lowerTranslation; }
	float32 GetUpperLimit() const { return m_upperTranslation; }
	bool IsMotorEnabled() const { return m_enableMotor; }
	float32 GetMotorForce(float32 inv_dt) const { return MotorReaction(inv_dt); }
	float32 GetMotorSpeed() const { return m_motorSpeed; }
	float32 GetMaxMotorForce() const { return m_maxMotorForce; }
	// computed from the body states of the last step (reference: b2PrismaticJoint.cpp:512-542)
	float32 GetJointTranslation() const;
	float32 GetJointSpeed() const;
	// setters forward to the device record; like the reference's they wake both bodies when something changes
	void EnableLimit(bool flag);
	void SetLimits(float32 lower, float32 upper);
	void EnableMotor(bool flag);
	void SetMotorSpeed(float32 speed);
	void SetMaxMotorForce(float32 force);

protected:
	friend class b2World;
	void PushMotor();
	void PushLimits();
	b2PrismaticJoint(const b2PrismaticJointDef* def) : b2Joint(def), m_localAnchorA(def->localAnchorA),
		m_localAnchorB(def->localAnchorB), m_localAxisA(def->localAxisA), m_referenceAngle(def->referenceAngle),
		m_enableLimit(def->enableLimit), m_lowerTranslation(def->lowerTranslation), m_upperTranslation(def->upperTranslation),
		m_enableMotor(def->enableMotor), m_maxMotorForce(def->maxMotorForce), m_motorSpeed(def->motorSpeed)
	{
		m_localAxisA.Normalize();
	}

	b2Vec2 m_localAnchorA, m_localAnchorB, m_localAxisA;
	float32 m_referenceAngle;
	bool m_enableLimit;
	float32 m_lowerTranslation, m_upperTranslation;
	bool m_enableMotor;
	float32 m_maxMotorForce, m_motorSpeed;
};

#endif
