// b2d_joint.h - joint constraints on the device path: revolute (Tumbler's motor) and distance (rigid rods and
// soft springs). One fixed-size record per joint whatever its type (the type-specific members share storage), so
// the island kernels, the upload and the snapshot handle one array.
// Restates b2RevoluteJoint::{InitVelocityConstraints, SolveVelocityConstraints, SolvePositionConstraints}
// (Box2D/Dynamics/Joints/b2RevoluteJoint.cpp:65-376) and the same three of b2DistanceJoint
// (Joints/b2DistanceJoint.cpp:65-236) in the reference's operand order; b2Mat33::Solve33 / Solve22 as in
// Box2D/Common/b2Math.cpp:25-53.
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
	B2D_JOINT_REVOLUTE = 0, // e_revoluteJoint
	B2D_JOINT_DISTANCE = 1  // e_distanceJoint
};

struct JointRec
{
	// definition (b2RevoluteJointDef / b2DistanceJointDef)
	int bodyA, bodyB;
	V2 localAnchorA, localAnchorB;
	union { float referenceAngle; float length; };
	int enableLimit;
	union { float lowerAngle; float frequencyHz; };
	union { float upperAngle; float dampingRatio; };
	int enableMotor;
	float motorSpeed, maxMotorTorque;
	int collideConnected;
	// persistent solver state (b2RevoluteJoint.h:190-199, b2DistanceJoint.h:148-150)
	union { float impulseX; float impulse; };
	float impulseY, impulseZ;
	float motorImpulse;
	int limitState;
	// per-step scratch written by init
	V2 rA, rB, localCenterA, localCenterB;
	float invMassA, invMassB, invIA, invIB;
	union { float m_exx; float mass; };  // revolute: m_mass (3x3) ; distance: m_mass, m_gamma, m_bias, m_u
	union { float m_exy; float gamma; };
	union { float m_exz; float bias; };
	union { float m_eyx; float ux; };
	union { float m_eyy; float uy; };
	float m_eyz, m_ezx, m_ezy, m_ezz;
	float motorMass;
	int islandFlag;
	int type;
};
typedef JointRec RevoluteJoint;

struct V3
{
	float x, y, z;
};

B2D_HD V3 b2dCross3(V3 a, V3 b)
{
	V3 r;
	r.x = a.y * b.z - a.z * b.y;
	r.y = a.z * b.x - a.x * b.z;
	r.z = a.x * b.y - a.y * b.x;
	return r;
}
B2D_HD float b2dDot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

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

// ---- dispatch on the joint type (b2Joint's virtual calls, b2Island.cpp:235-318) -------------------------
B2D_HD void b2dJointInit(JointRec* j, float invMassA, float invIA, V2 lcA, float invMassB, float invIB, V2 lcB,
	BodyPos pA, BodyVel* A, BodyPos pB, BodyVel* B, bool warmStarting, float dtRatio, float dt)
{
	if (j->type == B2D_JOINT_DISTANCE)
		b2dDistanceInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA, A, pB, B, warmStarting, dtRatio, dt);
	else
		b2dRevoluteInit(j, invMassA, invIA, lcA, invMassB, invIB, lcB, pA.a, A, pB.a, B, warmStarting, dtRatio);
}

B2D_HD void b2dJointSolveVelocity(JointRec* j, BodyVel* A, BodyVel* B, float dt)
{
	if (j->type == B2D_JOINT_DISTANCE)
		b2dDistanceSolveVelocity(j, A, B);
	else
		b2dRevoluteSolveVelocity(j, A, B, dt);
}

B2D_HD bool b2dJointSolvePosition(const JointRec* j, BodyPos* A, BodyPos* B)
{
	if (j->type == B2D_JOINT_DISTANCE) return b2dDistanceSolvePosition(j, A, B);
	return b2dRevoluteSolvePosition(j, A, B);
}

#endif
