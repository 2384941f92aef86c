// Drop-in header: gear joint definition (reference: Box2D/Dynamics/Joints/b2GearJoint.h:28-122).
// Couples two revolute / prismatic joints: coordinate1 + ratio * coordinate2 = constant. Both joints must be created
// first and destroyed after the gear. Solved on the device (csrc/b2d_joint.h, GearRec).
#ifndef B2_GEAR_JOINT_H
#define B2_GEAR_JOINT_H

#include "Box2D/Dynamics/Joints/b2Joint.h"

struct b2GearJointDef : public b2JointDef
{
	b2GearJointDef()
	{
		type = e_gearJoint;
		joint1 = nullptr;
		joint2 = nullptr;
		ratio = 1.0f;
	}
	b2Joint* joint1; // revolute or prismatic
	b2Joint* joint2; // revolute or prismatic
	float32 ratio;
};

class b2GearJoint : public b2Joint
{
public:
	b2Joint* GetJoint1() { return m_joint1; }
	b2Joint* GetJoint2() { return m_joint2; }
	float32 GetRatio() const { return m_ratio; }
	void SetRatio(float32 ratio);      // b2GearJoint.cpp:402-406

protected:
	friend class b2World;
	// the gear acts between the second bodies of its two joints (b2GearJoint.cpp:62-63, :95-96)
	b2GearJoint(const b2GearJointDef* def) : b2Joint(def), m_joint1(def->joint1), m_joint2(def->joint2), m_ratio(def->ratio)
	{
		m_bodyA = m_joint1->GetBodyB();
		m_bodyB = m_joint2->GetBodyB();
	}

	b2Joint* m_joint1;
	b2Joint* m_joint2;
	float32 m_ratio;
};

#endif
