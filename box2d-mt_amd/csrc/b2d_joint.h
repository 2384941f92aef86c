// b2d_joint.h - joint constraints on the device path: revolute (Tumbler's motor), distance (rigid rods and
// soft springs), prismatic (MultithreadDemo's slider: axis, limits, motor), weld (rigid or soft), wheel (suspension
// spring + axle motor), rope (maximum distance), friction and motor (top-down drag / pose servo), pulley, mouse (drag a body to a target), gear (couples two revolute / prismatic joints: four bodies, own record). One fixed-size record per joint whatever its type (the type-specific members share storage), so
// the island kernels, the upload and the snapshot handle one array.
// Restates b2RevoluteJoint::{InitVelocityConstraints, SolveVelocityConstraints, SolvePositionConstraints}
// (Box2D/Dynamics/Joints/b2RevoluteJoint.cpp:65-376) and the same three of b2DistanceJoint
// (Joints/b2DistanceJoint.cpp:65-236), b2PrismaticJoint (Joints/b2PrismaticJoint.cpp:130-478) and b2WeldJoint
// (Joints/b2WeldJoint.cpp:58-303) in the reference's operand order; b2Mat33::Solve33 / Solve22 / GetInverse22 /
// GetSymInverse33 as in Box2D/Common/b2Math.cpp:25-94. Also b2WheelJoint (Joints/b2WheelJoint.cpp:79-292), b2RopeJoint
// (Joints/b2RopeJoint.cpp:48-182), b2FrictionJoint (Joints/b2FrictionJoint.cpp:58-185), b2MotorJoint
// (Joints/b2MotorJoint.cpp:62-203), b2PulleyJoint (Joints/b2PulleyJoint.cpp:81-253),
// b2MouseJoint (Joints/b2MouseJoint.cpp:99-198), b2GearJoint (Joints/b2GearJoint.cpp:131-390).
#ifndef B2D_JOINT_H
#define B2D_JOINT_H

#include "b2d_solver.h"

#define B2D_MAX_ANGULAR_CORRECTION (8.0f / 180.0f * B2D_PI)

enum
{
	B2D_LIMIT_INACTIVE = 0, // b2LimitState (b2Joint.h:58-64)
	B2D_LIMIT_AT_LOWER = 1,
	B2D_LIMIT_AT_UPPER = 2,
	B2D_LIMIT_EQUAL = 3
};

enum
{
	B2D_JOINT_DEAD = -1,    // destroyed (b2World::DestroyJoint): ids stay stable, every walk over the joints skips it
	B2D_JOINT_REVOLUTE = 0, // e_revoluteJoint
	B2D_JOINT_DISTANCE = 1, // e_distanceJoint
	B2D_JOINT_PRISMATIC = 2, // e_prismaticJoint
	B2D_JOINT_WELD = 3,     // e_weldJoint
	B2D_JOINT_WHEEL = 4,    // e_wheelJoint
	B2D_JOINT_ROPE = 5,     // e_ropeJoint
	B2D_JOINT_FRICTION = 6, // e_frictionJoint
	B2D_JOINT_MOTOR = 7,    // e_motorJoint
	B2D_JOINT_PULLEY = 8,   // e_pulleyJoint
	B2D_JOINT_MOUSE = 9,    // e_mouseJoint
	B2D_JOINT_GEAR = 10     // e_gearJoint: the JointRec carries bodyA / bodyB and, in enableLimit, the index of its GearRec
};

struct JointRec
{
	// definition (b2RevoluteJointDef / b2DistanceJointDef / b2PrismaticJointDef / b2WeldJointDef)
	int bodyA, bodyB;
	union { V2 localAnchorA; V2 linearOffset; V2 targetA; };   // motor joint: m_linearOffset ; mouse joint: m_targetA (world)
	V2 localAnchorB;
	union { float referenceAngle; float length; float maxLength; float angularOffset; float bodyMass; }; // mouse: bodyB's mass
	int enableLimit;
	union { float lowerAngle; float frequencyHz; float lowerTranslation; float correctionFactor; float ratio; };
	union { float upperAngle; float dampingRatio; float upperTranslation; float constant; };  // pulley: m_ratio, m_constant
	int enableMotor;
	union { float motorSpeed; float maxTorque; };  // friction / motor joint: m_maxTorque
	union { float maxMotorTorque; float maxMotorForce; float maxForce; };
	int collideConnected;
	// persistent solver state (b2RevoluteJoint.h:190-199, b2DistanceJoint.h:148-150)
	union { float impulseX; float impulse; };      // wheel: m_impulse ; friction / motor: m_linearImpulse.x
	union { float impulseY; float springImpulse; }; // friction / motor: m_linearImpulse.y
	union { float impulseZ; float angularImpulse; };
	float motorImpulse;
	int limitState;
	// per-step scratch written by init
	V2 rA, rB, localCenterA, localCenterB;
	float invMassA, invMassB, invIA, invIB;
	union { float m_exx; float mass; };  // revolute: m_mass (3x3) ; distance: m_mass, m_gamma, m_bias, m_u
	union { float m_exy; float gamma; float springMass; float curLength; }; // wheel: m_springMass, m_bias, m_gamma in exy, exz, eyx
	union { float m_exz; float bias; };                                     // rope: m_length in exy ; friction / motor: m_linearMass
	                                                                         //   in exx, exy (ex) and eyx, eyy (ey)
	union { float m_eyx; float ux; };
	union { float m_eyy; float uy; };
	float m_eyz, m_ezx, m_ezy, m_ezz;
	float motorMass;
	int islandFlag;
	int type;
	union { V2 localAxisA; V2 groundAnchorA; };  // prismatic / wheel: m_localXAxisA ; pulley: ground anchors here and in s1, s2
	union { float s1; float wGamma; float sAx; float linErrX; };  // prismatic: m_s1, m_s2, m_a1, m_a2 ; weld: m_gamma, m_bias
	union { float s2; float wBias; float sBx; float linErrY; };   // wheel: m_sAx, m_sBx, m_sAy, m_sBy ; motor: m_linearError,
	union { float a1; float sAy; float angErr; };                 //   m_angularError
	union { float a2; float sBy; };
};
typedef JointRec RevoluteJoint;

#include "b2d_mat33.h"

B2D_HD M33 b2dJointMatrix(const struct JointRec* j);

B2D_HD V3 b2dSolve33(const RevoluteJoint* j, V3 b)
{
	V3 ex, ey, ez;
	ex.x = j->m_exx; ex.y = j->m_exy; ex.z = j->m_exz;
	ey.x = j->m_eyx; ey.y = j->m_eyy; ey.z = j->m_eyz;
	ez.x = j->m_ezx; ez.y = j->m_ezy; ez.z = j->m_ezz;
	float det = b2dDot3(ex, b2dCross3(ey, ez));
	if (det != 0.0f) det = 1.0f / det;
	V3 x;
	x.x = det * b2dDot3(b, b2dCross3(ey, ez));
	x.y = det * b2dDot3(ex, b2dCross3(b, ez));
	x.z = det * b2dDot3(ex, b2dCross3(ey, b));
	return x;
}

B2D_HD V2 b2dSolve22(const RevoluteJoint* j, V2 b)
{
	float a11 = j->m_exx, a12 = j->m_eyx, a21 = j->m_exy, a22 = j->m_eyy;
	float det = a11 * a22 - a12 * a21;
	if (det != 0.0f) det = 1.0f / det;
	V2 x;
	x.x = det * (a22 * b.x - a12 * b.y);
	x.y = det * (a11 * b.y - a21 * b.x);
	return x;
}

// InitVelocityConstraints (b2RevoluteJoint.cpp:65-182), including the joint's own warm start.
B2D_HD void b2dRevoluteInit(RevoluteJoint* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	float aA, BodyVel* A, float aB, BodyVel* B, bool warmStarting, float dtRatio)
{
	j->localCenterA = lcA;
	j->localCenterB = lcB;
	j->invMassA = invMassA;
	j->invMassB = invMassB;
	j->invIA = invIA;
	j->invIB = invIB;
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	Rot qA = b2dRot(aA), qB = b2dRot(aB);
	j->rA = b2dMulRV(qA, j->localAnchorA - lcA);
	j->rB = b2dMulRV(qB, j->localAnchorB - lcB);
	float mA = invMassA, mB = invMassB, iA = invIA, iB = invIB;
	bool fixedRotation = (iA + iB == 0.0f);
	V2 rA = j->rA, rB = j->rB;
	j->m_exx = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
	j->m_eyx = -rA.y * rA.x * iA - rB.y * rB.x * iB;
	j->m_ezx = -rA.y * iA - rB.y * iB;
	j->m_exy = j->m_eyx;
	j->m_eyy = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
	j->m_ezy = rA.x * iA + rB.x * iB;
	j->m_exz = j->m_ezx;
	j->m_eyz = j->m_ezy;
	j->m_ezz = iA + iB;
	j->motorMass = iA + iB;
	if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
	if (j->enableMotor == 0 || fixedRotation) j->motorImpulse = 0.0f;
	if (j->enableLimit && fixedRotation == false)
	{
		float jointAngle = aB - aA - j->referenceAngle;
		if (b2dAbs(j->upperAngle - j->lowerAngle) < 2.0f * B2D_ANGULAR_SLOP)
		{
			j->limitState = B2D_LIMIT_EQUAL;
		}
		else if (jointAngle <= j->lowerAngle)
		{
			if (j->limitState != B2D_LIMIT_AT_LOWER) j->impulseZ = 0.0f;
			j->limitState = B2D_LIMIT_AT_LOWER;
		}
		else if (jointAngle >= j->upperAngle)
		{
			if (j->limitState != B2D_LIMIT_AT_UPPER) j->impulseZ = 0.0f;
			j->limitState = B2D_LIMIT_AT_UPPER;
		}
		else
		{
			j->limitState = B2D_LIMIT_INACTIVE;
			j->impulseZ = 0.0f;
		}
	}
	else
	{
		j->limitState = B2D_LIMIT_INACTIVE;
	}
	if (warmStarting)
	{
		j->impulseX *= dtRatio;
		j->impulseY *= dtRatio;
		j->impulseZ *= dtRatio;
		j->motorImpulse *= dtRatio;
		V2 P = v2(j->impulseX, j->impulseY);
		vA -= mA * P;
		wA -= iA * (b2dCross(rA, P) + j->motorImpulse + j->impulseZ);
		vB += mB * P;
		wB += iB * (b2dCross(rB, P) + j->motorImpulse + j->impulseZ);
	}
	else
	{
		j->impulseX = j->impulseY = j->impulseZ = 0.0f;
		j->motorImpulse = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2RevoluteJoint.cpp:184-290)
B2D_HD void b2dRevoluteSolveVelocity(RevoluteJoint* j, BodyVel* A, BodyVel* B, float dt)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	bool fixedRotation = (iA + iB == 0.0f);
	V2 rA = j->rA, rB = j->rB;
	if (j->enableMotor && j->limitState != B2D_LIMIT_EQUAL && fixedRotation == false)
	{
		float Cdot = wB - wA - j->motorSpeed;
		float impulse = -j->motorMass * Cdot;
		float oldImpulse = j->motorImpulse;
		float maxImpulse = dt * j->maxMotorTorque;
		j->motorImpulse = b2dClamp(j->motorImpulse + impulse, -maxImpulse, maxImpulse);
		impulse = j->motorImpulse - oldImpulse;
		wA -= iA * impulse;
		wB += iB * impulse;
	}
	if (j->enableLimit && j->limitState != B2D_LIMIT_INACTIVE && fixedRotation == false)
	{
		V2 Cdot1 = vB + b2dCrossSV(wB, rB) - vA - b2dCrossSV(wA, rA);
		float Cdot2 = wB - wA;
		V3 Cdot;
		Cdot.x = Cdot1.x; Cdot.y = Cdot1.y; Cdot.z = Cdot2;
		V3 s = b2dSolve33(j, Cdot);
		V3 impulse;
		impulse.x = -s.x; impulse.y = -s.y; impulse.z = -s.z;
		if (j->limitState == B2D_LIMIT_EQUAL)
		{
			j->impulseX += impulse.x; j->impulseY += impulse.y; j->impulseZ += impulse.z;
		}
		else if (j->limitState == B2D_LIMIT_AT_LOWER)
		{
			float newImpulse = j->impulseZ + impulse.z;
			if (newImpulse < 0.0f)
			{
				V2 rhs = -Cdot1 + j->impulseZ * v2(j->m_ezx, j->m_ezy);
				V2 reduced = b2dSolve22(j, rhs);
				impulse.x = reduced.x;
				impulse.y = reduced.y;
				impulse.z = -j->impulseZ;
				j->impulseX += reduced.x;
				j->impulseY += reduced.y;
				j->impulseZ = 0.0f;
			}
			else
			{
				j->impulseX += impulse.x; j->impulseY += impulse.y; j->impulseZ += impulse.z;
			}
		}
		else if (j->limitState == B2D_LIMIT_AT_UPPER)
		{
			float newImpulse = j->impulseZ + impulse.z;
			if (newImpulse > 0.0f)
			{
				V2 rhs = -Cdot1 + j->impulseZ * v2(j->m_ezx, j->m_ezy);
				V2 reduced = b2dSolve22(j, rhs);
				impulse.x = reduced.x;
				impulse.y = reduced.y;
				impulse.z = -j->impulseZ;
				j->impulseX += reduced.x;
				j->impulseY += reduced.y;
				j->impulseZ = 0.0f;
			}
			else
			{
				j->impulseX += impulse.x; j->impulseY += impulse.y; j->impulseZ += impulse.z;
			}
		}
		V2 P = v2(impulse.x, impulse.y);
		vA -= mA * P;
		wA -= iA * (b2dCross(rA, P) + impulse.z);
		vB += mB * P;
		wB += iB * (b2dCross(rB, P) + impulse.z);
	}
	else
	{
		V2 Cdot = vB + b2dCrossSV(wB, rB) - vA - b2dCrossSV(wA, rA);
		V2 impulse = b2dSolve22(j, -Cdot);
		j->impulseX += impulse.x;
		j->impulseY += impulse.y;
		vA -= mA * impulse;
		wA -= iA * b2dCross(rA, impulse);
		vB += mB * impulse;
		wB += iB * b2dCross(rB, impulse);
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolvePositionConstraints (b2RevoluteJoint.cpp:292-376); returns jointOkay.
B2D_HD bool b2dRevoluteSolvePosition(const RevoluteJoint* j, BodyPos* A, BodyPos* B)
{
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	float angularError = 0.0f, positionError = 0.0f;
	bool fixedRotation = (j->invIA + j->invIB == 0.0f);
	if (j->enableLimit && j->limitState != B2D_LIMIT_INACTIVE && fixedRotation == false)
	{
		float angle = aB - aA - j->referenceAngle;
		float limitImpulse = 0.0f;
		if (j->limitState == B2D_LIMIT_EQUAL)
		{
			float C = b2dClamp(angle - j->lowerAngle, -B2D_MAX_ANGULAR_CORRECTION, B2D_MAX_ANGULAR_CORRECTION);
			limitImpulse = -j->motorMass * C;
			angularError = b2dAbs(C);
		}
		else if (j->limitState == B2D_LIMIT_AT_LOWER)
		{
			float C = angle - j->lowerAngle;
			angularError = -C;
			C = b2dClamp(C + B2D_ANGULAR_SLOP, -B2D_MAX_ANGULAR_CORRECTION, 0.0f);
			limitImpulse = -j->motorMass * C;
		}
		else if (j->limitState == B2D_LIMIT_AT_UPPER)
		{
			float C = angle - j->upperAngle;
			angularError = C;
			C = b2dClamp(C - B2D_ANGULAR_SLOP, 0.0f, B2D_MAX_ANGULAR_CORRECTION);
			limitImpulse = -j->motorMass * C;
		}
		aA -= j->invIA * limitImpulse;
		aB += j->invIB * limitImpulse;
	}
	{
		Rot qA = b2dRot(aA), qB = b2dRot(aB);
		V2 rA = b2dMulRV(qA, j->localAnchorA - j->localCenterA);
		V2 rB = b2dMulRV(qB, j->localAnchorB - j->localCenterB);
		V2 C = cB + rB - cA - rA;
		positionError = b2dLength(C);
		float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
		float k_exx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
		float k_exy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
		float k_eyx = k_exy;
		float k_eyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
		// b2Mat22::Solve (b2Math.h:221-233)
		float a11 = k_exx, a12 = k_eyx, a21 = k_exy, a22 = k_eyy;
		float det = a11 * a22 - a12 * a21;
		if (det != 0.0f) det = 1.0f / det;
		V2 sol = v2(det * (a22 * C.x - a12 * C.y), det * (a11 * C.y - a21 * C.x));
		V2 impulse = -sol;
		cA -= mA * impulse;
		aA -= iA * b2dCross(rA, impulse);
		cB += mB * impulse;
		aB += iB * b2dCross(rB, impulse);
	}
	A->c = cA; A->a = aA;
	B->c = cB; B->a = aB;
	return positionError <= B2D_LINEAR_SLOP && angularError <= B2D_ANGULAR_SLOP;
}

// ---- distance joint ---------------------------------------------------------------------------------
// InitVelocityConstraints (b2DistanceJoint.cpp:65-157)
B2D_HD void b2dDistanceInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio, float dt)
{
	j->localCenterA = lcA;
	j->localCenterB = lcB;
	j->invMassA = invMassA;
	j->invMassB = invMassB;
	j->invIA = invIA;
	j->invIB = invIB;
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	V2 rA = b2dMulRV(qA, j->localAnchorA - lcA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - lcB);
	j->rA = rA;
	j->rB = rB;
	V2 u = pB.c + rB - pA.c - rA;
	float length = b2dLength(u);
	if (length > B2D_LINEAR_SLOP)
	{
		u = (1.0f / length) * u;
	}
	else
	{
		u = v2(0.0f, 0.0f);
	}
	float crAu = b2dCross(rA, u);
	float crBu = b2dCross(rB, u);
	float invMass = invMassA + invIA * crAu * crAu + invMassB + invIB * crBu * crBu;
	float mass = invMass != 0.0f ? 1.0f / invMass : 0.0f;
	if (j->frequencyHz > 0.0f)
	{
		float C = length - j->length;
		float omega = 2.0f * B2D_PI * j->frequencyHz;
		float d = 2.0f * mass * j->dampingRatio * omega;
		float k = mass * omega * omega;
		float gamma = dt * (d + dt * k);
		gamma = gamma != 0.0f ? 1.0f / gamma : 0.0f;
		j->gamma = gamma;
		j->bias = C * dt * k * gamma;
		invMass += gamma;
		mass = invMass != 0.0f ? 1.0f / invMass : 0.0f;
	}
	else
	{
		j->gamma = 0.0f;
		j->bias = 0.0f;
	}
	j->mass = mass;
	j->ux = u.x;
	j->uy = u.y;
	if (warmStarting)
	{
		j->impulse *= dtRatio;
		V2 P = j->impulse * u;
		vA -= invMassA * P;
		wA -= invIA * b2dCross(rA, P);
		vB += invMassB * P;
		wB += invIB * b2dCross(rB, P);
	}
	else
	{
		j->impulse = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2DistanceJoint.cpp:159-184)
B2D_HD void b2dDistanceSolveVelocity(JointRec* j, BodyVel* A, BodyVel* B)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	V2 rA = j->rA, rB = j->rB, u = v2(j->ux, j->uy);
	V2 vpA = vA + b2dCrossSV(wA, rA);
	V2 vpB = vB + b2dCrossSV(wB, rB);
	float Cdot = b2dDot(u, vpB - vpA);
	float impulse = -j->mass * (Cdot + j->bias + j->gamma * j->impulse);
	j->impulse += impulse;
	V2 P = impulse * u;
	vA -= j->invMassA * P;
	wA -= j->invIA * b2dCross(rA, P);
	vB += j->invMassB * P;
	wB += j->invIB * b2dCross(rB, P);
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolvePositionConstraints (b2DistanceJoint.cpp:186-225): soft joints have no position correction.
B2D_HD bool b2dDistanceSolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	if (j->frequencyHz > 0.0f) return true;
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	Rot qA = b2dRot(aA), qB = b2dRot(aB);
	V2 rA = b2dMulRV(qA, j->localAnchorA - j->localCenterA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - j->localCenterB);
	V2 u = cB + rB - cA - rA;
	float length = b2dNormalize(u);
	float C = length - j->length;
	C = b2dClamp(C, -B2D_MAX_LINEAR_CORRECTION, B2D_MAX_LINEAR_CORRECTION);
	float impulse = -j->mass * C;
	V2 P = impulse * u;
	cA -= j->invMassA * P;
	aA -= j->invIA * b2dCross(rA, P);
	cB += j->invMassB * P;
	aB += j->invIB * b2dCross(rB, P);
	A->c = cA; A->a = aA;
	B->c = cB; B->a = aB;
	return b2dAbs(C) < B2D_LINEAR_SLOP;
}

B2D_HD M33 b2dJointMatrix(const JointRec* j)
{
	M33 K;
	K.ex = v3(j->m_exx, j->m_exy, j->m_exz);
	K.ey = v3(j->m_eyx, j->m_eyy, j->m_eyz);
	K.ez = v3(j->m_ezx, j->m_ezy, j->m_ezz);
	return K;
}

B2D_HD void b2dJointSetMatrix(JointRec* j, const M33& K)
{
	j->m_exx = K.ex.x; j->m_exy = K.ex.y; j->m_exz = K.ex.z;
	j->m_eyx = K.ey.x; j->m_eyy = K.ey.y; j->m_eyz = K.ey.z;
	j->m_ezx = K.ez.x; j->m_ezy = K.ez.y; j->m_ezz = K.ez.z;
}

B2D_HD void b2dJointStoreBodies(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB)
{
	j->localCenterA = lcA;
	j->localCenterB = lcB;
	j->invMassA = invMassA;
	j->invMassB = invMassB;
	j->invIA = invIA;
	j->invIB = invIB;
}

// ---- prismatic joint --------------------------------------------------------------------------------
// The record keeps m_axis in rA and m_perp in rB (the lever arms themselves are not needed after init).
// InitVelocityConstraints (b2PrismaticJoint.cpp:130-257)
B2D_HD void b2dPrismaticInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio)
{
	b2dJointStoreBodies(j, invMassA, invIA, lcA, invMassB, invIB, lcB);
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	V2 rA = b2dMulRV(qA, j->localAnchorA - lcA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - lcB);
	V2 d = (pB.c - pA.c) + rB - rA;
	float mA = invMassA, mB = invMassB, iA = invIA, iB = invIB;
	V2 axis = b2dMulRV(qA, j->localAxisA);
	float a1 = b2dCross(d + rA, axis);
	float a2 = b2dCross(rB, axis);
	float motorMass = mA + mB + iA * a1 * a1 + iB * a2 * a2;
	if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
	V2 perp = b2dMulRV(qA, b2dCrossSV(1.0f, j->localAxisA));
	float s1 = b2dCross(d + rA, perp);
	float s2 = b2dCross(rB, perp);
	{
		float k11 = mA + mB + iA * s1 * s1 + iB * s2 * s2;
		float k12 = iA * s1 + iB * s2;
		float k13 = iA * s1 * a1 + iB * s2 * a2;
		float k22 = iA + iB;
		if (k22 == 0.0f) k22 = 1.0f; // bodies with fixed rotation
		float k23 = iA * a1 + iB * a2;
		float k33 = mA + mB + iA * a1 * a1 + iB * a2 * a2;
		M33 K;
		K.ex = v3(k11, k12, k13);
		K.ey = v3(k12, k22, k23);
		K.ez = v3(k13, k23, k33);
		b2dJointSetMatrix(j, K);
	}
	j->rA = axis;
	j->rB = perp;
	j->a1 = a1; j->a2 = a2; j->s1 = s1; j->s2 = s2;
	j->motorMass = motorMass;
	if (j->enableLimit)
	{
		float jointTranslation = b2dDot(axis, d);
		if (b2dAbs(j->upperTranslation - j->lowerTranslation) < 2.0f * B2D_LINEAR_SLOP)
		{
			j->limitState = B2D_LIMIT_EQUAL;
		}
		else if (jointTranslation <= j->lowerTranslation)
		{
			if (j->limitState != B2D_LIMIT_AT_LOWER)
			{
				j->limitState = B2D_LIMIT_AT_LOWER;
				j->impulseZ = 0.0f;
			}
		}
		else if (jointTranslation >= j->upperTranslation)
		{
			if (j->limitState != B2D_LIMIT_AT_UPPER)
			{
				j->limitState = B2D_LIMIT_AT_UPPER;
				j->impulseZ = 0.0f;
			}
		}
		else
		{
			j->limitState = B2D_LIMIT_INACTIVE;
			j->impulseZ = 0.0f;
		}
	}
	else
	{
		j->limitState = B2D_LIMIT_INACTIVE;
		j->impulseZ = 0.0f;
	}
	if (j->enableMotor == 0) j->motorImpulse = 0.0f;
	if (warmStarting)
	{
		j->impulseX *= dtRatio;
		j->impulseY *= dtRatio;
		j->impulseZ *= dtRatio;
		j->motorImpulse *= dtRatio;
		V2 P = j->impulseX * perp + (j->motorImpulse + j->impulseZ) * axis;
		float LA = j->impulseX * s1 + j->impulseY + (j->motorImpulse + j->impulseZ) * a1;
		float LB = j->impulseX * s2 + j->impulseY + (j->motorImpulse + j->impulseZ) * a2;
		vA -= mA * P;
		wA -= iA * LA;
		vB += mB * P;
		wB += iB * LB;
	}
	else
	{
		j->impulseX = j->impulseY = j->impulseZ = 0.0f;
		j->motorImpulse = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2PrismaticJoint.cpp:259-350)
B2D_HD void b2dPrismaticSolveVelocity(JointRec* j, BodyVel* A, BodyVel* B, float dt)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	const V2 axis = j->rA, perp = j->rB;
	const float a1 = j->a1, a2 = j->a2, s1 = j->s1, s2 = j->s2;
	if (j->enableMotor && j->limitState != B2D_LIMIT_EQUAL)
	{
		float Cdot = b2dDot(axis, vB - vA) + a2 * wB - a1 * wA;
		float impulse = j->motorMass * (j->motorSpeed - Cdot);
		float oldImpulse = j->motorImpulse;
		float maxImpulse = dt * j->maxMotorForce;
		j->motorImpulse = b2dClamp(j->motorImpulse + impulse, -maxImpulse, maxImpulse);
		impulse = j->motorImpulse - oldImpulse;
		V2 P = impulse * axis;
		float LA = impulse * a1;
		float LB = impulse * a2;
		vA -= mA * P;
		wA -= iA * LA;
		vB += mB * P;
		wB += iB * LB;
	}
	V2 Cdot1;
	Cdot1.x = b2dDot(perp, vB - vA) + s2 * wB - s1 * wA;
	Cdot1.y = wB - wA;
	const M33 K = b2dJointMatrix(j);
	if (j->enableLimit && j->limitState != B2D_LIMIT_INACTIVE)
	{
		float Cdot2 = b2dDot(axis, vB - vA) + a2 * wB - a1 * wA;
		V3 f1 = v3(j->impulseX, j->impulseY, j->impulseZ);
		V3 df = b2dM33Solve33(K, v3(-Cdot1.x, -Cdot1.y, -Cdot2));
		V3 imp = v3(f1.x + df.x, f1.y + df.y, f1.z + df.z);
		if (j->limitState == B2D_LIMIT_AT_LOWER)
		{
			imp.z = b2dMax(imp.z, 0.0f);
		}
		else if (j->limitState == B2D_LIMIT_AT_UPPER)
		{
			imp.z = b2dMin(imp.z, 0.0f);
		}
		// the two unlimited rows again with the limit impulse fixed
		V2 b = -Cdot1 - (imp.z - f1.z) * v2(K.ez.x, K.ez.y);
		V2 f2r = b2dM33Solve22(K, b) + v2(f1.x, f1.y);
		imp.x = f2r.x;
		imp.y = f2r.y;
		df = v3(imp.x - f1.x, imp.y - f1.y, imp.z - f1.z);
		j->impulseX = imp.x; j->impulseY = imp.y; j->impulseZ = imp.z;
		V2 P = df.x * perp + df.z * axis;
		float LA = df.x * s1 + df.y + df.z * a1;
		float LB = df.x * s2 + df.y + df.z * a2;
		vA -= mA * P;
		wA -= iA * LA;
		vB += mB * P;
		wB += iB * LB;
	}
	else
	{
		V2 df = b2dM33Solve22(K, -Cdot1);
		j->impulseX += df.x;
		j->impulseY += df.y;
		V2 P = df.x * perp;
		float LA = df.x * s1 + df.y;
		float LB = df.x * s2 + df.y;
		vA -= mA * P;
		wA -= iA * LA;
		vB += mB * P;
		wB += iB * LB;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolvePositionConstraints (b2PrismaticJoint.cpp:352-478)
B2D_HD bool b2dPrismaticSolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	Rot qA = b2dRot(aA), qB = b2dRot(aB);
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	V2 rA = b2dMulRV(qA, j->localAnchorA - j->localCenterA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - j->localCenterB);
	V2 d = cB + rB - cA - rA;
	V2 axis = b2dMulRV(qA, j->localAxisA);
	float a1 = b2dCross(d + rA, axis);
	float a2 = b2dCross(rB, axis);
	V2 perp = b2dMulRV(qA, b2dCrossSV(1.0f, j->localAxisA));
	float s1 = b2dCross(d + rA, perp);
	float s2 = b2dCross(rB, perp);
	V3 impulse;
	V2 C1;
	C1.x = b2dDot(perp, d);
	C1.y = aB - aA - j->referenceAngle;
	float linearError = b2dAbs(C1.x);
	float angularError = b2dAbs(C1.y);
	bool active = false;
	float C2 = 0.0f;
	if (j->enableLimit)
	{
		float translation = b2dDot(axis, d);
		if (b2dAbs(j->upperTranslation - j->lowerTranslation) < 2.0f * B2D_LINEAR_SLOP)
		{
			C2 = b2dClamp(translation, -B2D_MAX_LINEAR_CORRECTION, B2D_MAX_LINEAR_CORRECTION);
			linearError = b2dMax(linearError, b2dAbs(translation));
			active = true;
		}
		else if (translation <= j->lowerTranslation)
		{
			C2 = b2dClamp(translation - j->lowerTranslation + B2D_LINEAR_SLOP, -B2D_MAX_LINEAR_CORRECTION, 0.0f);
			linearError = b2dMax(linearError, j->lowerTranslation - translation);
			active = true;
		}
		else if (translation >= j->upperTranslation)
		{
			C2 = b2dClamp(translation - j->upperTranslation - B2D_LINEAR_SLOP, 0.0f, B2D_MAX_LINEAR_CORRECTION);
			linearError = b2dMax(linearError, translation - j->upperTranslation);
			active = true;
		}
	}
	float k11 = mA + mB + iA * s1 * s1 + iB * s2 * s2;
	float k12 = iA * s1 + iB * s2;
	float k22 = iA + iB;
	if (k22 == 0.0f) k22 = 1.0f;
	if (active)
	{
		float k13 = iA * s1 * a1 + iB * s2 * a2;
		float k23 = iA * a1 + iB * a2;
		float k33 = mA + mB + iA * a1 * a1 + iB * a2 * a2;
		M33 K;
		K.ex = v3(k11, k12, k13);
		K.ey = v3(k12, k22, k23);
		K.ez = v3(k13, k23, k33);
		impulse = b2dM33Solve33(K, v3(-C1.x, -C1.y, -C2));
	}
	else
	{
		// b2Mat22::Solve (b2Math.h:221-233)
		float a11 = k11, a12 = k12, a21 = k12, a22 = k22;
		float det = a11 * a22 - a12 * a21;
		if (det != 0.0f) det = 1.0f / det;
		V2 nb = -C1;
		impulse = v3(det * (a22 * nb.x - a12 * nb.y), det * (a11 * nb.y - a21 * nb.x), 0.0f);
	}
	V2 P = impulse.x * perp + impulse.z * axis;
	float LA = impulse.x * s1 + impulse.y + impulse.z * a1;
	float LB = impulse.x * s2 + impulse.y + impulse.z * a2;
	cA -= mA * P;
	aA -= iA * LA;
	cB += mB * P;
	aB += iB * LB;
	A->c = cA; A->a = aA;
	B->c = cB; B->a = aB;
	return linearError <= B2D_LINEAR_SLOP && angularError <= B2D_ANGULAR_SLOP;
}

// ---- weld joint -------------------------------------------------------------------------------------
B2D_HD M33 b2dWeldK(V2 rA, V2 rB, float mA, float mB, float iA, float iB)
{
	M33 K;
	K.ex.x = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
	K.ey.x = -rA.y * rA.x * iA - rB.y * rB.x * iB;
	K.ez.x = -rA.y * iA - rB.y * iB;
	K.ex.y = K.ey.x;
	K.ey.y = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
	K.ez.y = rA.x * iA + rB.x * iB;
	K.ex.z = K.ez.x;
	K.ey.z = K.ez.y;
	K.ez.z = iA + iB;
	return K;
}

// InitVelocityConstraints (b2WeldJoint.cpp:58-155); the record's matrix holds m_mass (the inverse)
B2D_HD void b2dWeldInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio, float dt)
{
	b2dJointStoreBodies(j, invMassA, invIA, lcA, invMassB, invIB, lcB);
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	V2 rA = b2dMulRV(qA, j->localAnchorA - lcA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - lcB);
	j->rA = rA;
	j->rB = rB;
	float mA = invMassA, mB = invMassB, iA = invIA, iB = invIB;
	M33 K = b2dWeldK(rA, rB, mA, mB, iA, iB);
	M33 mass;
	if (j->frequencyHz > 0.0f)
	{
		mass = b2dM33Inverse22(K);
		float invM = iA + iB;
		float m = invM > 0.0f ? 1.0f / invM : 0.0f;
		float C = pB.a - pA.a - j->referenceAngle;
		float omega = 2.0f * B2D_PI * j->frequencyHz;
		float d = 2.0f * m * j->dampingRatio * omega;
		float k = m * omega * omega;
		float gamma = dt * (d + dt * k);
		gamma = gamma != 0.0f ? 1.0f / gamma : 0.0f;
		j->wGamma = gamma;
		j->wBias = C * dt * k * gamma;
		invM += gamma;
		mass.ez.z = invM != 0.0f ? 1.0f / invM : 0.0f;
	}
	else if (K.ez.z == 0.0f)
	{
		mass = b2dM33Inverse22(K);
		j->wGamma = 0.0f;
		j->wBias = 0.0f;
	}
	else
	{
		mass = b2dM33SymInverse33(K);
		j->wGamma = 0.0f;
		j->wBias = 0.0f;
	}
	b2dJointSetMatrix(j, mass);
	if (warmStarting)
	{
		j->impulseX *= dtRatio;
		j->impulseY *= dtRatio;
		j->impulseZ *= dtRatio;
		V2 P = v2(j->impulseX, j->impulseY);
		vA -= mA * P;
		wA -= iA * (b2dCross(rA, P) + j->impulseZ);
		vB += mB * P;
		wB += iB * (b2dCross(rB, P) + j->impulseZ);
	}
	else
	{
		j->impulseX = j->impulseY = j->impulseZ = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2WeldJoint.cpp:157-216)
B2D_HD void b2dWeldSolveVelocity(JointRec* j, BodyVel* A, BodyVel* B)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	const V2 rA = j->rA, rB = j->rB;
	const M33 mass = b2dJointMatrix(j);
	if (j->frequencyHz > 0.0f)
	{
		float Cdot2 = wB - wA;
		float impulse2 = -mass.ez.z * (Cdot2 + j->wBias + j->wGamma * j->impulseZ);
		j->impulseZ += impulse2;
		wA -= iA * impulse2;
		wB += iB * impulse2;
		V2 Cdot1 = vB + b2dCrossSV(wB, rB) - vA - b2dCrossSV(wA, rA);
		V2 impulse1 = -b2dM33Mul22(mass, Cdot1);
		j->impulseX += impulse1.x;
		j->impulseY += impulse1.y;
		V2 P = impulse1;
		vA -= mA * P;
		wA -= iA * b2dCross(rA, P);
		vB += mB * P;
		wB += iB * b2dCross(rB, P);
	}
	else
	{
		V2 Cdot1 = vB + b2dCrossSV(wB, rB) - vA - b2dCrossSV(wA, rA);
		float Cdot2 = wB - wA;
		V3 m = b2dM33Mul(mass, v3(Cdot1.x, Cdot1.y, Cdot2));
		V3 impulse = v3(-m.x, -m.y, -m.z);
		j->impulseX += impulse.x;
		j->impulseY += impulse.y;
		j->impulseZ += impulse.z;
		V2 P = v2(impulse.x, impulse.y);
		vA -= mA * P;
		wA -= iA * (b2dCross(rA, P) + impulse.z);
		vB += mB * P;
		wB += iB * (b2dCross(rB, P) + impulse.z);
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolvePositionConstraints (b2WeldJoint.cpp:218-303)
B2D_HD bool b2dWeldSolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	Rot qA = b2dRot(aA), qB = b2dRot(aB);
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	V2 rA = b2dMulRV(qA, j->localAnchorA - j->localCenterA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - j->localCenterB);
	float positionError, angularError;
	M33 K = b2dWeldK(rA, rB, mA, mB, iA, iB);
	if (j->frequencyHz > 0.0f)
	{
		V2 C1 = cB + rB - cA - rA;
		positionError = b2dLength(C1);
		angularError = 0.0f;
		V2 P = -b2dM33Solve22(K, C1);
		cA -= mA * P;
		aA -= iA * b2dCross(rA, P);
		cB += mB * P;
		aB += iB * b2dCross(rB, P);
	}
	else
	{
		V2 C1 = cB + rB - cA - rA;
		float C2 = aB - aA - j->referenceAngle;
		positionError = b2dLength(C1);
		angularError = b2dAbs(C2);
		V3 impulse;
		if (K.ez.z > 0.0f)
		{
			V3 x = b2dM33Solve33(K, v3(C1.x, C1.y, C2));
			impulse = v3(-x.x, -x.y, -x.z);
		}
		else
		{
			V2 impulse2 = -b2dM33Solve22(K, C1);
			impulse = v3(impulse2.x, impulse2.y, 0.0f);
		}
		V2 P = v2(impulse.x, impulse.y);
		cA -= mA * P;
		aA -= iA * (b2dCross(rA, P) + impulse.z);
		cB += mB * P;
		aB += iB * (b2dCross(rB, P) + impulse.z);
	}
	A->c = cA; A->a = aA;
	B->c = cB; B->a = aB;
	return positionError <= B2D_LINEAR_SLOP && angularError <= B2D_ANGULAR_SLOP;
}

// ---- wheel joint ------------------------------------------------------------------------------------
// The record keeps m_ax in rA and m_ay in rB. InitVelocityConstraints (b2WheelJoint.cpp:79-198)
B2D_HD void b2dWheelInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio, float dt)
{
	b2dJointStoreBodies(j, invMassA, invIA, lcA, invMassB, invIB, lcB);
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	float mA = invMassA, mB = invMassB, iA = invIA, iB = invIB;
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	V2 rA = b2dMulRV(qA, j->localAnchorA - lcA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - lcB);
	V2 d = pB.c + rB - pA.c - rA;
	// point to line constraint
	V2 ay = b2dMulRV(qA, b2dCrossSV(1.0f, j->localAxisA));
	j->rB = ay;
	j->sAy = b2dCross(d + rA, ay);
	j->sBy = b2dCross(rB, ay);
	j->mass = mA + mB + iA * j->sAy * j->sAy + iB * j->sBy * j->sBy;
	if (j->mass > 0.0f) j->mass = 1.0f / j->mass;
	// spring constraint
	j->springMass = 0.0f;
	j->bias = 0.0f;
	float gamma = 0.0f;
	if (j->frequencyHz > 0.0f)
	{
		V2 ax = b2dMulRV(qA, j->localAxisA);
		j->rA = ax;
		j->sAx = b2dCross(d + rA, ax);
		j->sBx = b2dCross(rB, ax);
		float invMass = mA + mB + iA * j->sAx * j->sAx + iB * j->sBx * j->sBx;
		if (invMass > 0.0f)
		{
			j->springMass = 1.0f / invMass;
			float C = b2dDot(d, ax);
			float omega = 2.0f * B2D_PI * j->frequencyHz;
			float damp = 2.0f * j->springMass * j->dampingRatio * omega;
			float k = j->springMass * omega * omega;
			gamma = dt * (damp + dt * k);
			if (gamma > 0.0f) gamma = 1.0f / gamma;
			j->bias = C * dt * k * gamma;
			j->springMass = invMass + gamma;
			if (j->springMass > 0.0f) j->springMass = 1.0f / j->springMass;
		}
	}
	else
	{
		j->springImpulse = 0.0f;
	}
	j->m_eyx = gamma;
	if (j->enableMotor)
	{
		j->motorMass = iA + iB;
		if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
	}
	else
	{
		j->motorMass = 0.0f;
		j->motorImpulse = 0.0f;
	}
	if (warmStarting)
	{
		j->impulse *= dtRatio;
		j->springImpulse *= dtRatio;
		j->motorImpulse *= dtRatio;
		const V2 ax = j->rA;
		V2 P = j->impulse * ay + j->springImpulse * ax;
		float LA = j->impulse * j->sAy + j->springImpulse * j->sAx + j->motorImpulse;
		float LB = j->impulse * j->sBy + j->springImpulse * j->sBx + j->motorImpulse;
		vA -= mA * P;
		wA -= iA * LA;
		vB += mB * P;
		wB += iB * LB;
	}
	else
	{
		j->impulse = 0.0f;
		j->springImpulse = 0.0f;
		j->motorImpulse = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2WheelJoint.cpp:200-257): spring, motor, then the line constraint
B2D_HD void b2dWheelSolveVelocity(JointRec* j, BodyVel* A, BodyVel* B, float dt)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	const V2 ax = j->rA, ay = j->rB;
	{
		float Cdot = b2dDot(ax, vB - vA) + j->sBx * wB - j->sAx * wA;
		float impulse = -j->springMass * (Cdot + j->bias + j->m_eyx * j->springImpulse);
		j->springImpulse += impulse;
		V2 P = impulse * ax;
		float LA = impulse * j->sAx;
		float LB = impulse * j->sBx;
		vA -= mA * P;
		wA -= iA * LA;
		vB += mB * P;
		wB += iB * LB;
	}
	{
		float Cdot = wB - wA - j->motorSpeed;
		float impulse = -j->motorMass * Cdot;
		float oldImpulse = j->motorImpulse;
		float maxImpulse = dt * j->maxMotorTorque;
		j->motorImpulse = b2dClamp(j->motorImpulse + impulse, -maxImpulse, maxImpulse);
		impulse = j->motorImpulse - oldImpulse;
		wA -= iA * impulse;
		wB += iB * impulse;
	}
	{
		float Cdot = b2dDot(ay, vB - vA) + j->sBy * wB - j->sAy * wA;
		float impulse = -j->mass * Cdot;
		j->impulse += impulse;
		V2 P = impulse * ay;
		float LA = impulse * j->sAy;
		float LB = impulse * j->sBy;
		vA -= mA * P;
		wA -= iA * LA;
		vB += mB * P;
		wB += iB * LB;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolvePositionConstraints (b2WheelJoint.cpp:259-292). The effective mass there is built from the lever arms stored
// by InitVelocityConstraints (m_sAy, m_sBy), not from the ones of the current positions - kept.
B2D_HD bool b2dWheelSolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	Rot qA = b2dRot(aA), qB = b2dRot(aB);
	V2 rA = b2dMulRV(qA, j->localAnchorA - j->localCenterA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - j->localCenterB);
	V2 d = (cB - cA) + rB - rA;
	V2 ay = b2dMulRV(qA, b2dCrossSV(1.0f, j->localAxisA));
	float sAy = b2dCross(d + rA, ay);
	float sBy = b2dCross(rB, ay);
	float C = b2dDot(d, ay);
	float k = j->invMassA + j->invMassB + j->invIA * j->sAy * j->sAy + j->invIB * j->sBy * j->sBy;
	float impulse;
	if (k != 0.0f)
	{
		impulse = -C / k;
	}
	else
	{
		impulse = 0.0f;
	}
	V2 P = impulse * ay;
	float LA = impulse * sAy;
	float LB = impulse * sBy;
	cA -= j->invMassA * P;
	aA -= j->invIA * LA;
	cB += j->invMassB * P;
	aB += j->invIB * LB;
	A->c = cA; A->a = aA;
	B->c = cB; B->a = aB;
	return b2dAbs(C) <= B2D_LINEAR_SLOP;
}

// ---- rope joint -------------------------------------------------------------------------------------
// InitVelocityConstraints (b2RopeJoint.cpp:48-115); limitState holds m_state, curLength m_length
B2D_HD void b2dRopeInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio)
{
	b2dJointStoreBodies(j, invMassA, invIA, lcA, invMassB, invIB, lcB);
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	V2 rA = b2dMulRV(qA, j->localAnchorA - lcA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - lcB);
	j->rA = rA;
	j->rB = rB;
	V2 u = pB.c + rB - pA.c - rA;
	j->curLength = b2dLength(u);
	float C = j->curLength - j->maxLength;
	j->limitState = C > 0.0f ? B2D_LIMIT_AT_UPPER : B2D_LIMIT_INACTIVE;
	if (j->curLength > B2D_LINEAR_SLOP)
	{
		u = (1.0f / j->curLength) * u;
	}
	else
	{
		j->ux = 0.0f;
		j->uy = 0.0f;
		j->mass = 0.0f;
		j->impulse = 0.0f;
		return;
	}
	j->ux = u.x;
	j->uy = u.y;
	float crA = b2dCross(rA, u);
	float crB = b2dCross(rB, u);
	float invMass = invMassA + invIA * crA * crA + invMassB + invIB * crB * crB;
	j->mass = invMass != 0.0f ? 1.0f / invMass : 0.0f;
	if (warmStarting)
	{
		j->impulse *= dtRatio;
		V2 P = j->impulse * u;
		vA -= invMassA * P;
		wA -= invIA * b2dCross(rA, P);
		vB += invMassB * P;
		wB += invIB * b2dCross(rB, P);
	}
	else
	{
		j->impulse = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2RopeJoint.cpp:117-149): predictive when the rope is still slack
B2D_HD void b2dRopeSolveVelocity(JointRec* j, BodyVel* A, BodyVel* B, float inv_dt)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	const V2 rA = j->rA, rB = j->rB, u = v2(j->ux, j->uy);
	V2 vpA = vA + b2dCrossSV(wA, rA);
	V2 vpB = vB + b2dCrossSV(wB, rB);
	float C = j->curLength - j->maxLength;
	float Cdot = b2dDot(u, vpB - vpA);
	if (C < 0.0f) Cdot += inv_dt * C;
	float impulse = -j->mass * Cdot;
	float oldImpulse = j->impulse;
	j->impulse = b2dMin(0.0f, j->impulse + impulse);
	impulse = j->impulse - oldImpulse;
	V2 P = impulse * u;
	vA -= j->invMassA * P;
	wA -= j->invIA * b2dCross(rA, P);
	vB += j->invMassB * P;
	wB += j->invIB * b2dCross(rB, P);
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolvePositionConstraints (b2RopeJoint.cpp:151-182)
B2D_HD bool b2dRopeSolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	Rot qA = b2dRot(aA), qB = b2dRot(aB);
	V2 rA = b2dMulRV(qA, j->localAnchorA - j->localCenterA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - j->localCenterB);
	V2 u = cB + rB - cA - rA;
	float length = b2dNormalize(u);
	float C = length - j->maxLength;
	C = b2dClamp(C, 0.0f, B2D_MAX_LINEAR_CORRECTION);
	float impulse = -j->mass * C;
	V2 P = impulse * u;
	cA -= j->invMassA * P;
	aA -= j->invIA * b2dCross(rA, P);
	cB += j->invMassB * P;
	aB += j->invIB * b2dCross(rB, P);
	A->c = cA; A->a = aA;
	B->c = cB; B->a = aB;
	return length - j->maxLength < B2D_LINEAR_SLOP;
}

// ---- friction and motor joints: a clamped 2-D linear row pair + a clamped angular row ------------------
// m_linearMass = K.GetInverse() (b2Math.h:205-217), stored as ex = (exx, exy), ey = (eyx, eyy)
B2D_HD void b2dJointLinearMass(JointRec* j, V2 rA, V2 rB, float mA, float mB, float iA, float iB)
{
	float kxx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
	float kxy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
	float kyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
	float a = kxx, b = kxy, c = kxy, d = kyy;
	float det = a * d - b * c;
	if (det != 0.0f) det = 1.0f / det;
	j->m_exx = det * d; j->m_eyx = -det * b;
	j->m_exy = -det * c; j->m_eyy = det * a;
	j->motorMass = iA + iB; // m_angularMass
	if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
}

B2D_HD void b2dJointWarmStartLinearAngular(JointRec* j, V2 rA, V2 rB, BodyVel* A, BodyVel* B, bool warmStarting, float dtRatio)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	if (warmStarting)
	{
		j->impulseX *= dtRatio;
		j->impulseY *= dtRatio;
		j->angularImpulse *= dtRatio;
		V2 P = v2(j->impulseX, j->impulseY);
		vA -= j->invMassA * P;
		wA -= j->invIA * (b2dCross(rA, P) + j->angularImpulse);
		vB += j->invMassB * P;
		wB += j->invIB * (b2dCross(rB, P) + j->angularImpulse);
	}
	else
	{
		j->impulseX = j->impulseY = 0.0f;
		j->angularImpulse = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// angular row then linear rows, each clamped to dt * max (b2FrictionJoint.cpp:124-171, b2MotorJoint.cpp:136-189);
// angBias / linBias are zero for the friction joint
B2D_HD void b2dJointSolveLinearAngular(JointRec* j, BodyVel* A, BodyVel* B, float dt, float angBias, V2 linBias, bool biased)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	const V2 rA = j->rA, rB = j->rB;
	{
		float Cdot = wB - wA;
		if (biased) Cdot = Cdot + angBias;
		float impulse = -j->motorMass * Cdot;
		float oldImpulse = j->angularImpulse;
		float maxImpulse = dt * j->maxTorque;
		j->angularImpulse = b2dClamp(j->angularImpulse + impulse, -maxImpulse, maxImpulse);
		impulse = j->angularImpulse - oldImpulse;
		wA -= iA * impulse;
		wB += iB * impulse;
	}
	{
		V2 Cdot = vB + b2dCrossSV(wB, rB) - vA - b2dCrossSV(wA, rA);
		if (biased) Cdot = Cdot + linBias;
		V2 impulse = -v2(j->m_exx * Cdot.x + j->m_eyx * Cdot.y, j->m_exy * Cdot.x + j->m_eyy * Cdot.y);
		V2 oldImpulse = v2(j->impulseX, j->impulseY);
		V2 acc = oldImpulse + impulse;
		float maxImpulse = dt * j->maxForce;
		if (acc.x * acc.x + acc.y * acc.y > maxImpulse * maxImpulse)
		{
			b2dNormalize(acc);
			acc *= maxImpulse;
		}
		j->impulseX = acc.x;
		j->impulseY = acc.y;
		impulse = acc - oldImpulse;
		vA -= mA * impulse;
		wA -= iA * b2dCross(rA, impulse);
		vB += mB * impulse;
		wB += iB * b2dCross(rB, impulse);
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// b2FrictionJoint::InitVelocityConstraints (b2FrictionJoint.cpp:58-122)
B2D_HD void b2dFrictionInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio)
{
	b2dJointStoreBodies(j, invMassA, invIA, lcA, invMassB, invIB, lcB);
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	j->rA = b2dMulRV(qA, j->localAnchorA - lcA);
	j->rB = b2dMulRV(qB, j->localAnchorB - lcB);
	b2dJointLinearMass(j, j->rA, j->rB, invMassA, invMassB, invIA, invIB);
	b2dJointWarmStartLinearAngular(j, j->rA, j->rB, A, B, warmStarting, dtRatio);
}

// b2MotorJoint::InitVelocityConstraints (b2MotorJoint.cpp:62-134)
B2D_HD void b2dMotorInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio)
{
	b2dJointStoreBodies(j, invMassA, invIA, lcA, invMassB, invIB, lcB);
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	j->rA = b2dMulRV(qA, j->linearOffset - lcA);
	j->rB = b2dMulRV(qB, -lcB);
	b2dJointLinearMass(j, j->rA, j->rB, invMassA, invMassB, invIA, invIB);
	V2 linearError = pB.c + j->rB - pA.c - j->rA;
	j->linErrX = linearError.x;
	j->linErrY = linearError.y;
	j->angErr = pB.a - pA.a - j->angularOffset;
	b2dJointWarmStartLinearAngular(j, j->rA, j->rB, A, B, warmStarting, dtRatio);
}

// ---- pulley joint -----------------------------------------------------------------------------------
// length1 + ratio * length2 = constant. m_uA is kept in (m_eyx, m_eyy), m_uB in (m_eyz, m_ezx), ground anchor B in (s1, s2).
B2D_HD V2 b2dPulleyDir(V2 u, float* length)
{
	*length = b2dLength(u);
	if (*length > 10.0f * B2D_LINEAR_SLOP) return (1.0f / *length) * u;
	return v2(0.0f, 0.0f);
}

// InitVelocityConstraints (b2PulleyJoint.cpp:81-156)
B2D_HD void b2dPulleyInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio)
{
	b2dJointStoreBodies(j, invMassA, invIA, lcA, invMassB, invIB, lcB);
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	Rot qA = b2dRot(pA.a), qB = b2dRot(pB.a);
	V2 rA = b2dMulRV(qA, j->localAnchorA - lcA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - lcB);
	j->rA = rA;
	j->rB = rB;
	float lengthA, lengthB;
	V2 uA = b2dPulleyDir(pA.c + rA - j->groundAnchorA, &lengthA);
	V2 uB = b2dPulleyDir(pB.c + rB - v2(j->s1, j->s2), &lengthB);
	j->m_eyx = uA.x; j->m_eyy = uA.y;
	j->m_eyz = uB.x; j->m_ezx = uB.y;
	float ruA = b2dCross(rA, uA);
	float ruB = b2dCross(rB, uB);
	float mA = invMassA + invIA * ruA * ruA;
	float mB = invMassB + invIB * ruB * ruB;
	j->mass = mA + j->ratio * j->ratio * mB;
	if (j->mass > 0.0f) j->mass = 1.0f / j->mass;
	if (warmStarting)
	{
		j->impulse *= dtRatio;
		V2 PA = -(j->impulse) * uA;
		V2 PB = (-j->ratio * j->impulse) * uB;
		vA += invMassA * PA;
		wA += invIA * b2dCross(rA, PA);
		vB += invMassB * PB;
		wB += invIB * b2dCross(rB, PB);
	}
	else
	{
		j->impulse = 0.0f;
	}
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2PulleyJoint.cpp:158-183)
B2D_HD void b2dPulleySolveVelocity(JointRec* j, BodyVel* A, BodyVel* B)
{
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	const V2 rA = j->rA, rB = j->rB, uA = v2(j->m_eyx, j->m_eyy), uB = v2(j->m_eyz, j->m_ezx);
	V2 vpA = vA + b2dCrossSV(wA, rA);
	V2 vpB = vB + b2dCrossSV(wB, rB);
	float Cdot = -b2dDot(uA, vpA) - j->ratio * b2dDot(uB, vpB);
	float impulse = -j->mass * Cdot;
	j->impulse += impulse;
	V2 PA = -impulse * uA;
	V2 PB = -j->ratio * impulse * uB;
	vA += j->invMassA * PA;
	wA += j->invIA * b2dCross(rA, PA);
	vB += j->invMassB * PB;
	wB += j->invIB * b2dCross(rB, PB);
	A->v = vA; A->w = wA;
	B->v = vB; B->w = wB;
}

// SolvePositionConstraints (b2PulleyJoint.cpp:185-253)
B2D_HD bool b2dPulleySolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	Rot qA = b2dRot(aA), qB = b2dRot(aB);
	V2 rA = b2dMulRV(qA, j->localAnchorA - j->localCenterA);
	V2 rB = b2dMulRV(qB, j->localAnchorB - j->localCenterB);
	float lengthA, lengthB;
	V2 uA = b2dPulleyDir(cA + rA - j->groundAnchorA, &lengthA);
	V2 uB = b2dPulleyDir(cB + rB - v2(j->s1, j->s2), &lengthB);
	float ruA = b2dCross(rA, uA);
	float ruB = b2dCross(rB, uB);
	float mA = j->invMassA + j->invIA * ruA * ruA;
	float mB = j->invMassB + j->invIB * ruB * ruB;
	float mass = mA + j->ratio * j->ratio * mB;
	if (mass > 0.0f) mass = 1.0f / mass;
	float C = j->constant - lengthA - j->ratio * lengthB;
	float linearError = b2dAbs(C);
	float impulse = -mass * C;
	V2 PA = -impulse * uA;
	V2 PB = -j->ratio * impulse * uB;
	cA += j->invMassA * PA;
	aA += j->invIA * b2dCross(rA, PA);
	cB += j->invMassB * PB;
	aB += j->invIB * b2dCross(rB, PB);
	A->c = cA; A->a = aA;
	B->c = cB; B->a = aB;
	return linearError < B2D_LINEAR_SLOP;
}

// ---- mouse joint ------------------------------------------------------------------------------------
// Acts on bodyB only. m_mass (2x2) in exx, exy (ex) and eyx, eyy (ey); m_C in (s1, s2); m_beta in a1, m_gamma in a2.
// InitVelocityConstraints (b2MouseJoint.cpp:99-163)
B2D_HD void b2dMouseInit(JointRec* j, float invMassB, float invIB, V2 lcB, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio,
	float dt)
{
	j->localCenterB = lcB;
	j->invMassB = invMassB;
	j->invIB = invIB;
	V2 vB = B->v;
	float wB = B->w;
	Rot qB = b2dRot(pB.a);
	float mass = j->bodyMass;
	float omega = 2.0f * B2D_PI * j->frequencyHz;
	float d = 2.0f * mass * j->dampingRatio * omega;
	float k = mass * (omega * omega);
	float gamma = dt * (d + dt * k);
	if (gamma != 0.0f) gamma = 1.0f / gamma;
	float beta = dt * k * gamma;
	j->a1 = beta;
	j->a2 = gamma;
	V2 rB = b2dMulRV(qB, j->localAnchorB - lcB);
	j->rB = rB;
	float kxx = invMassB + invIB * rB.y * rB.y + gamma;
	float kxy = -invIB * rB.x * rB.y;
	float kyy = invMassB + invIB * rB.x * rB.x + gamma;
	float a = kxx, b = kxy, c = kxy, dd = kyy;
	float det = a * dd - b * c;
	if (det != 0.0f) det = 1.0f / det;
	j->m_exx = det * dd; j->m_eyx = -det * b;
	j->m_exy = -det * c; j->m_eyy = det * a;
	V2 C = pB.c + rB - j->targetA;
	C *= beta;
	j->s1 = C.x;
	j->s2 = C.y;
	wB *= 0.98f; // the reference's built-in damping
	if (warmStarting)
	{
		j->impulseX *= dtRatio;
		j->impulseY *= dtRatio;
		V2 P = v2(j->impulseX, j->impulseY);
		vB += invMassB * P;
		wB += invIB * b2dCross(rB, P);
	}
	else
	{
		j->impulseX = j->impulseY = 0.0f;
	}
	B->v = vB; B->w = wB;
}

// SolveVelocityConstraints (b2MouseJoint.cpp:165-192)
B2D_HD void b2dMouseSolveVelocity(JointRec* j, BodyVel* B, float dt)
{
	V2 vB = B->v;
	float wB = B->w;
	const V2 rB = j->rB;
	V2 Cdot = vB + b2dCrossSV(wB, rB);
	V2 oldImpulse = v2(j->impulseX, j->impulseY);
	V2 rhs = -(Cdot + v2(j->s1, j->s2) + j->a2 * oldImpulse);
	V2 impulse = v2(j->m_exx * rhs.x + j->m_eyx * rhs.y, j->m_exy * rhs.x + j->m_eyy * rhs.y);
	V2 acc = oldImpulse + impulse;
	float maxImpulse = dt * j->maxForce;
	if (acc.x * acc.x + acc.y * acc.y > maxImpulse * maxImpulse)
	{
		acc *= maxImpulse / b2dLength(acc);
	}
	j->impulseX = acc.x;
	j->impulseY = acc.y;
	impulse = acc - oldImpulse;
	vB += j->invMassB * impulse;
	wB += j->invIB * b2dCross(rB, impulse);
	B->v = vB; B->w = wB;
}

// ---- gear joint -------------------------------------------------------------------------------------
// coordinateA + ratio * coordinateB = constant, where a coordinate is the angle of a revolute joint or the translation of a
// prismatic joint. Bodies A and B are the moving bodies of the two joints, C and D the bodies they are attached to.
struct GearRec
{
	// definition, derived from the two joints at creation (b2GearJoint.cpp:50-129)
	int bodyC, bodyD;
	int typeA, typeB;                 // B2D_JOINT_REVOLUTE or B2D_JOINT_PRISMATIC
	V2 localAnchorA, localAnchorB, localAnchorC, localAnchorD;
	V2 localAxisC, localAxisD;
	float referenceAngleA, referenceAngleB;
	float ratio, constant;
	// persistent
	float impulse;
	// per-step scratch written by init
	V2 lcA, lcB, lcC, lcD;
	float mA, mB, mC, mD, iA, iB, iC, iD;
	V2 JvAC, JvBD;
	float JwA, JwB, JwC, JwD;
	float mass;
};

struct GearBodies
{
	BodyPos pA, pB, pC, pD;
	BodyVel vA, vB, vC, vD;
};

// the Jacobian of one side (b2GearJoint.cpp:159-195 and again :276-331); returns the side's share of the effective mass
B2D_HD float b2dGearSide(int type, float scale, V2 localAxis, V2 localAnchorFixed, V2 lcFixed, V2 localAnchorMoving, V2 lcMoving,
	float aFixed, float aMoving, float mFixed, float mMoving, float iFixed, float iMoving, V2* Jv, float* JwMoving, float* JwFixed,
	bool isB)
{
	if (type == B2D_JOINT_REVOLUTE)
	{
		*Jv = v2(0.0f, 0.0f);
		*JwMoving = scale;
		*JwFixed = scale;
		return isB ? scale * scale * (iMoving + iFixed) : iMoving + iFixed;
	}
	Rot qF = b2dRot(aFixed), qM = b2dRot(aMoving);
	V2 u = b2dMulRV(qF, localAxis);
	V2 rF = b2dMulRV(qF, localAnchorFixed - lcFixed);
	V2 rM = b2dMulRV(qM, localAnchorMoving - lcMoving);
	if (isB)
	{
		*Jv = scale * u;
		*JwFixed = scale * b2dCross(rF, u);
		*JwMoving = scale * b2dCross(rM, u);
		return scale * scale * (mFixed + mMoving) + iFixed * *JwFixed * *JwFixed + iMoving * *JwMoving * *JwMoving;
	}
	*Jv = u;
	*JwFixed = b2dCross(rF, u);
	*JwMoving = b2dCross(rM, u);
	return mFixed + mMoving + iFixed * *JwFixed * *JwFixed + iMoving * *JwMoving * *JwMoving;
}

// InitVelocityConstraints (b2GearJoint.cpp:131-222)
B2D_HD void b2dGearInit(GearRec* g, GearBodies* b, const float invMass[4], const float invI[4], const V2 lc[4], bool warmStarting)
{
	g->mA = invMass[0]; g->iA = invI[0]; g->lcA = lc[0];
	g->mB = invMass[1]; g->iB = invI[1]; g->lcB = lc[1];
	g->mC = invMass[2]; g->iC = invI[2]; g->lcC = lc[2];
	g->mD = invMass[3]; g->iD = invI[3]; g->lcD = lc[3];
	float mass = 0.0f;
	mass += b2dGearSide(g->typeA, 1.0f, g->localAxisC, g->localAnchorC, g->lcC, g->localAnchorA, g->lcA, b->pC.a, b->pA.a,
		g->mC, g->mA, g->iC, g->iA, &g->JvAC, &g->JwA, &g->JwC, false);
	mass += b2dGearSide(g->typeB, g->ratio, g->localAxisD, g->localAnchorD, g->lcD, g->localAnchorB, g->lcB, b->pD.a, b->pB.a,
		g->mD, g->mB, g->iD, g->iB, &g->JvBD, &g->JwB, &g->JwD, true);
	g->mass = mass > 0.0f ? 1.0f / mass : 0.0f;
	if (warmStarting)
	{
		b->vA.v += (g->mA * g->impulse) * g->JvAC;
		b->vA.w += g->iA * g->impulse * g->JwA;
		b->vB.v += (g->mB * g->impulse) * g->JvBD;
		b->vB.w += g->iB * g->impulse * g->JwB;
		b->vC.v -= (g->mC * g->impulse) * g->JvAC;
		b->vC.w -= g->iC * g->impulse * g->JwC;
		b->vD.v -= (g->mD * g->impulse) * g->JvBD;
		b->vD.w -= g->iD * g->impulse * g->JwD;
	}
	else
	{
		g->impulse = 0.0f;
	}
}

// SolveVelocityConstraints (b2GearJoint.cpp:224-258)
B2D_HD void b2dGearSolveVelocity(GearRec* g, GearBodies* b)
{
	float Cdot = b2dDot(g->JvAC, b->vA.v - b->vC.v) + b2dDot(g->JvBD, b->vB.v - b->vD.v);
	Cdot += (g->JwA * b->vA.w - g->JwC * b->vC.w) + (g->JwB * b->vB.w - g->JwD * b->vD.w);
	float impulse = -g->mass * Cdot;
	g->impulse += impulse;
	b->vA.v += (g->mA * impulse) * g->JvAC;
	b->vA.w += g->iA * impulse * g->JwA;
	b->vB.v += (g->mB * impulse) * g->JvBD;
	b->vB.w += g->iB * impulse * g->JwB;
	b->vC.v -= (g->mC * impulse) * g->JvAC;
	b->vC.w -= g->iC * impulse * g->JwC;
	b->vD.v -= (g->mD * impulse) * g->JvBD;
	b->vD.w -= g->iD * impulse * g->JwD;
}

// one side's coordinate at the current positions (b2GearJoint.cpp:284-305, :313-334)
B2D_HD float b2dGearCoordinate(int type, V2 localAxis, V2 localAnchorFixed, V2 lcFixed, V2 localAnchorMoving, V2 lcMoving, BodyPos pF,
	BodyPos pM, float referenceAngle)
{
	if (type == B2D_JOINT_REVOLUTE) return pM.a - pF.a - referenceAngle;
	Rot qF = b2dRot(pF.a), qM = b2dRot(pM.a);
	V2 rM = b2dMulRV(qM, localAnchorMoving - lcMoving);
	V2 pFix = localAnchorFixed - lcFixed;
	V2 pMov = b2dMulTRV(qF, rM + (pM.c - pF.c));
	return b2dDot(pMov - pFix, localAxis);
}

// SolvePositionConstraints (b2GearJoint.cpp:260-362); its linear error is never updated from zero, so it always reports success
B2D_HD bool b2dGearSolvePosition(const GearRec* g, GearBodies* b)
{
	V2 JvAC, JvBD;
	float JwA, JwB, JwC, JwD;
	float mass = 0.0f;
	mass += b2dGearSide(g->typeA, 1.0f, g->localAxisC, g->localAnchorC, g->lcC, g->localAnchorA, g->lcA, b->pC.a, b->pA.a,
		g->mC, g->mA, g->iC, g->iA, &JvAC, &JwA, &JwC, false);
	float coordinateA = b2dGearCoordinate(g->typeA, g->localAxisC, g->localAnchorC, g->lcC, g->localAnchorA, g->lcA, b->pC, b->pA,
		g->referenceAngleA);
	mass += b2dGearSide(g->typeB, g->ratio, g->localAxisD, g->localAnchorD, g->lcD, g->localAnchorB, g->lcB, b->pD.a, b->pB.a,
		g->mD, g->mB, g->iD, g->iB, &JvBD, &JwB, &JwD, true);
	float coordinateB = b2dGearCoordinate(g->typeB, g->localAxisD, g->localAnchorD, g->lcD, g->localAnchorB, g->lcB, b->pD, b->pB,
		g->referenceAngleB);
	float C = (coordinateA + g->ratio * coordinateB) - g->constant;
	float impulse = 0.0f;
	if (mass > 0.0f) impulse = -C / mass;
	b->pA.c += g->mA * impulse * JvAC;
	b->pA.a += g->iA * impulse * JwA;
	b->pB.c += g->mB * impulse * JvBD;
	b->pB.a += g->iB * impulse * JwB;
	b->pC.c -= g->mC * impulse * JvAC;
	b->pC.a -= g->iC * impulse * JwC;
	b->pD.c -= g->mD * impulse * JvBD;
	b->pD.a -= g->iD * impulse * JwD;
	return true;
}

// ---- dispatch on the joint type (b2Joint's virtual calls, b2Island.cpp:235-318) -------------------------
B2D_HD void b2dJointInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio, float dt)
{
	if (j->type == B2D_JOINT_DISTANCE)
		b2dDistanceInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio, dt);
	else if (j->type == B2D_JOINT_PRISMATIC)
		b2dPrismaticInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio);
	else if (j->type == B2D_JOINT_WELD)
		b2dWeldInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio, dt);
	else if (j->type == B2D_JOINT_WHEEL)
		b2dWheelInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio, dt);
	else if (j->type == B2D_JOINT_ROPE)
		b2dRopeInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio);
	else if (j->type == B2D_JOINT_FRICTION)
		b2dFrictionInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio);
	else if (j->type == B2D_JOINT_MOTOR)
		b2dMotorInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio);
	else if (j->type == B2D_JOINT_PULLEY)
		b2dPulleyInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio);
	else if (j->type == B2D_JOINT_MOUSE)
		b2dMouseInit(j, invMassB, invIB, lcB, pB, B, warmStarting, dtRatio, dt);
	else
		b2dRevoluteInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA.a, A, pB.a, B, warmStarting, dtRatio);
}

B2D_HD void b2dJointSolveVelocity(JointRec* j, BodyVel* A, BodyVel* B, float dt, float inv_dt)
{
	if (j->type == B2D_JOINT_DISTANCE)
		b2dDistanceSolveVelocity(j, A, B);
	else if (j->type == B2D_JOINT_PRISMATIC)
		b2dPrismaticSolveVelocity(j, A, B, dt);
	else if (j->type == B2D_JOINT_WELD)
		b2dWeldSolveVelocity(j, A, B);
	else if (j->type == B2D_JOINT_WHEEL)
		b2dWheelSolveVelocity(j, A, B, dt);
	else if (j->type == B2D_JOINT_ROPE)
		b2dRopeSolveVelocity(j, A, B, inv_dt);
	else if (j->type == B2D_JOINT_PULLEY)
		b2dPulleySolveVelocity(j, A, B);
	else if (j->type == B2D_JOINT_MOUSE)
		b2dMouseSolveVelocity(j, B, dt);
	else if (j->type == B2D_JOINT_FRICTION)
		b2dJointSolveLinearAngular(j, A, B, dt, 0.0f, v2(0.0f, 0.0f), false);
	else if (j->type == B2D_JOINT_MOTOR)
		b2dJointSolveLinearAngular(j, A, B, dt, inv_dt * j->correctionFactor * j->angErr,
			(inv_dt * j->correctionFactor) * v2(j->linErrX, j->linErrY), true);
	else
		b2dRevoluteSolveVelocity(j, A, B, dt);
}

B2D_HD bool b2dJointSolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	if (j->type == B2D_JOINT_DISTANCE) return b2dDistanceSolvePosition(j, A, B);
	if (j->type == B2D_JOINT_PRISMATIC) return b2dPrismaticSolvePosition(j, A, B);
	if (j->type == B2D_JOINT_WELD) return b2dWeldSolvePosition(j, A, B);
	if (j->type == B2D_JOINT_WHEEL) return b2dWheelSolvePosition(j, A, B);
	if (j->type == B2D_JOINT_ROPE) return b2dRopeSolvePosition(j, A, B);
	if (j->type == B2D_JOINT_PULLEY) return b2dPulleySolvePosition(j, A, B);
	if (j->type == B2D_JOINT_FRICTION || j->type == B2D_JOINT_MOTOR || j->type == B2D_JOINT_MOUSE) return true;
	return b2dRevoluteSolvePosition(j, A, B);
}

// ---- what a joint did in the last step: b2Joint::GetReactionForce / GetReactionTorque and the motor's share
// (GetMotorTorque / GetMotorForce), from the record's accumulated impulses and the directions its init stored. Reference:
// b2RevoluteJoint.cpp:439-456, b2DistanceJoint.cpp:236-247, b2PrismaticJoint.cpp:502-510,618-621, b2WeldJoint.cpp:316-325,
// b2WheelJoint.cpp:340-348,434-437, b2RopeJoint.cpp:207-217, b2FrictionJoint.cpp:222-230, b2MotorJoint.cpp:236-244,
// b2PulleyJoint.cpp:273-283, b2MouseJoint.cpp:210-218, b2GearJoint.cpp:381-391.
struct JointReaction
{
	V2 force;     // on bodyB at the anchor, newtons
	float torque; // on bodyB, N m
	float motor;  // motor torque / force (revolute, prismatic, wheel), else 0
};

B2D_HD JointReaction b2dJointReaction(const JointRec* j, const GearRec* gear, float inv_dt)
{
	JointReaction r;
	r.force = v2(0.0f, 0.0f);
	r.torque = 0.0f;
	r.motor = 0.0f;
	switch (j->type)
	{
	case B2D_JOINT_REVOLUTE:
	case B2D_JOINT_WELD:
		r.force = inv_dt * v2(j->impulseX, j->impulseY);
		r.torque = inv_dt * j->impulseZ;
		if (j->type == B2D_JOINT_REVOLUTE) r.motor = inv_dt * j->motorImpulse;
		break;
	case B2D_JOINT_DISTANCE:
	case B2D_JOINT_ROPE:
		r.force = (inv_dt * j->impulse) * v2(j->ux, j->uy);
		break;
	case B2D_JOINT_PRISMATIC:
		// (init keeps m_axis in rA and m_perp in rB)
		r.force = inv_dt * (j->impulseX * j->rB + (j->motorImpulse + j->impulseZ) * j->rA);
		r.torque = inv_dt * j->impulseY;
		r.motor = inv_dt * j->motorImpulse;
		break;
	case B2D_JOINT_WHEEL:
		// (m_ax in rA, m_ay in rB)
		r.force = inv_dt * (j->impulse * j->rB + j->springImpulse * j->rA);
		r.torque = inv_dt * j->motorImpulse;
		r.motor = inv_dt * j->motorImpulse;
		break;
	case B2D_JOINT_FRICTION:
	case B2D_JOINT_MOTOR:
		r.force = inv_dt * v2(j->impulseX, j->impulseY);
		r.torque = inv_dt * j->angularImpulse;
		break;
	case B2D_JOINT_PULLEY:
	{
		const V2 P = j->impulse * v2(j->m_eyz, j->m_ezx); // m_uB
		r.force = inv_dt * P;
		break;
	}
	case B2D_JOINT_MOUSE:
		r.force = inv_dt * v2(j->impulseX, j->impulseY);
		r.torque = inv_dt * 0.0f;
		break;
	case B2D_JOINT_GEAR:
		if (gear != nullptr)
		{
			const V2 P = gear->impulse * gear->JvAC;
			r.force = inv_dt * P;
			const float L = gear->impulse * gear->JwA;
			r.torque = inv_dt * L;
		}
		break;
	default:
		break;
	}
	return r;
}

#endif
