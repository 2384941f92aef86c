// b2d_solver.h - sequential-impulse contact constraint, one constraint per lane.
//
// The constraint lives in the owning lane's registers (small islands) or in SoA rows in HBM
// (coloured large islands); the arithmetic is the same and follows b2ContactSolver.cpp line by
// line in operand order (cited per function) so that a lane walking the reference's constraint
// order reproduces its floats exactly.
#ifndef B2D_SOLVER_H
#define B2D_SOLVER_H

#include "b2d_collide.h"

// A manifold has at most two points: loops over them are written with a constant trip count and a predicate, so that after
// unrolling every index into the constraint's small arrays is a constant and the constraint can live in registers (a
// data-dependent trip count made the compiler keep it in scratch memory: 240-300 bytes per lane, re-read at every solve).
// Same operations in the same order: the floats do not change.
#define B2D_FOR_POINTS(j, count) _Pragma("unroll") for (int j = 0; j < 2; ++j) if (j < (count))

struct BodyVel
{
	V2 v;
	float w;
};

struct BodyPos
{
	V2 c;
	float a;
};

// b2ContactVelocityConstraint + b2ContactPositionConstraint (b2ContactSolver.h:31-57, .cpp:32-45)
struct ContactConstraint
{
	// velocity part
	V2 normal;
	V2 rA[2], rB[2];
	float normalImpulse[2], tangentImpulse[2];
	float normalMass[2], tangentMass[2], velocityBias[2];
	float nm_exx, nm_exy, nm_eyx, nm_eyy; // normalMass matrix (inverse of K)
	float k_exx, k_exy, k_eyx, k_eyy;     // K
	float invMassA, invMassB, invIA, invIB;
	float friction, restitution, tangentSpeed;
	int pointCount; // may be reduced to 1 by the conditioning guard
	// position part
	V2 localNormal, localPoint;
	V2 localPoints[2];
	V2 localCenterA, localCenterB;
	float radiusA, radiusB;
	int type;
	int pcPointCount;
};

// b2ContactSolver::b2ContactSolver (b2ContactSolver.cpp:47-133) + InitializeVelocityConstraints (:142-251)
// posA/posB, velA/velB are the island state AFTER velocity integration and BEFORE warm starting.
template <bool INLINE_ROT = false>
B2D_HD void b2dInitConstraint(ContactConstraint* cc, const Manifold* mf,
	float friction, float restitution, float tangentSpeed,
	float invMassA, float invIA, V2 localCenterA, float radiusA,
	float invMassB, float invIB, V2 localCenterB, float radiusB,
	BodyPos posA, BodyVel velA, BodyPos posB, BodyVel velB,
	bool warmStarting, float dtRatio)
{
	int pointCount = mf->pointCount;
	cc->friction = friction;
	cc->restitution = restitution;
	cc->tangentSpeed = tangentSpeed;
	cc->invMassA = invMassA;
	cc->invMassB = invMassB;
	cc->invIA = invIA;
	cc->invIB = invIB;
	cc->pointCount = pointCount;
	cc->k_exx = cc->k_exy = cc->k_eyx = cc->k_eyy = 0.0f;
	cc->nm_exx = cc->nm_exy = cc->nm_eyx = cc->nm_eyy = 0.0f;
	cc->localCenterA = localCenterA;
	cc->localCenterB = localCenterB;
	cc->localNormal = mf->localNormal;
	cc->localPoint = mf->localPoint;
	cc->pcPointCount = pointCount;
	cc->radiusA = radiusA;
	cc->radiusB = radiusB;
	cc->type = mf->type;
	for (int j = 0; j < 2; ++j)
	{
		cc->normalImpulse[j] = 0.0f;
		cc->tangentImpulse[j] = 0.0f;
		cc->rA[j] = v2(0.0f, 0.0f);
		cc->rB[j] = v2(0.0f, 0.0f);
		cc->normalMass[j] = 0.0f;
		cc->tangentMass[j] = 0.0f;
		cc->velocityBias[j] = 0.0f;
		cc->localPoints[j] = v2(0.0f, 0.0f);
	}
	B2D_FOR_POINTS(j, pointCount)
	{
		if (warmStarting)
		{
			cc->normalImpulse[j] = dtRatio * mf->ni[j];
			cc->tangentImpulse[j] = dtRatio * mf->ti[j];
		}
		cc->localPoints[j] = mf->p[j];
	}

	float mA = invMassA, mB = invMassB, iA = invIA, iB = invIB;
	V2 cA = posA.c, cB = posB.c;
	float aA = posA.a, aB = posB.a;
	V2 vA = velA.v, vB = velB.v;
	float wA = velA.w, wB = velB.w;

	Xf xfA, xfB;
	xfA.q = INLINE_ROT ? b2dRotInline(aA) : b2dRot(aA);
	xfB.q = INLINE_ROT ? b2dRotInline(aB) : b2dRot(aB);
	xfA.p = cA - b2dMulRV(xfA.q, localCenterA);
	xfB.p = cB - b2dMulRV(xfB.q, localCenterB);

	// b2WorldManifold::Initialize (b2Collision.cpp:22-86)
	V2 wmNormal = v2(0.0f, 0.0f);
	V2 wmPoints[2];
	wmPoints[0] = v2(0.0f, 0.0f);
	wmPoints[1] = v2(0.0f, 0.0f);
	if (mf->type == B2D_MANIFOLD_CIRCLES)
	{
		wmNormal = v2(1.0f, 0.0f);
		V2 pointA = b2dMulXV(xfA, mf->localPoint);
		V2 pointB = b2dMulXV(xfB, mf->p[0]);
		if (b2dDistanceSquared(pointA, pointB) > B2D_EPSILON * B2D_EPSILON)
		{
			wmNormal = pointB - pointA;
			b2dNormalize(wmNormal);
		}
		V2 ca = pointA + radiusA * wmNormal;
		V2 cb = pointB - radiusB * wmNormal;
		wmPoints[0] = 0.5f * (ca + cb);
	}
	else if (mf->type == B2D_MANIFOLD_FACE_A)
	{
		wmNormal = b2dMulRV(xfA.q, mf->localNormal);
		V2 planePoint = b2dMulXV(xfA, mf->localPoint);
		B2D_FOR_POINTS(i, pointCount)
		{
			V2 clipPoint = b2dMulXV(xfB, mf->p[i]);
			V2 ca = clipPoint + (radiusA - b2dDot(clipPoint - planePoint, wmNormal)) * wmNormal;
			V2 cb = clipPoint - radiusB * wmNormal;
			wmPoints[i] = 0.5f * (ca + cb);
		}
	}
	else
	{
		wmNormal = b2dMulRV(xfB.q, mf->localNormal);
		V2 planePoint = b2dMulXV(xfB, mf->localPoint);
		B2D_FOR_POINTS(i, pointCount)
		{
			V2 clipPoint = b2dMulXV(xfA, mf->p[i]);
			V2 cb = clipPoint + (radiusB - b2dDot(clipPoint - planePoint, wmNormal)) * wmNormal;
			V2 ca = clipPoint - radiusA * wmNormal;
			wmPoints[i] = 0.5f * (ca + cb);
		}
		wmNormal = -wmNormal;
	}

	cc->normal = wmNormal;

	B2D_FOR_POINTS(j, pointCount)
	{
		cc->rA[j] = wmPoints[j] - cA;
		cc->rB[j] = wmPoints[j] - cB;

		float rnA = b2dCross(cc->rA[j], cc->normal);
		float rnB = b2dCross(cc->rB[j], cc->normal);
		float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
		cc->normalMass[j] = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;

		V2 tangent = b2dCrossVS(cc->normal, 1.0f);
		float rtA = b2dCross(cc->rA[j], tangent);
		float rtB = b2dCross(cc->rB[j], tangent);
		float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
		cc->tangentMass[j] = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;

		cc->velocityBias[j] = 0.0f;
		float vRel = b2dDot(cc->normal, vB + b2dCrossSV(wB, cc->rB[j]) - vA - b2dCrossSV(wA, cc->rA[j]));
		if (vRel < -B2D_VELOCITY_THRESHOLD)
		{
			cc->velocityBias[j] = -cc->restitution * vRel;
		}
	}

	if (cc->pointCount == 2)
	{
		float rn1A = b2dCross(cc->rA[0], cc->normal);
		float rn1B = b2dCross(cc->rB[0], cc->normal);
		float rn2A = b2dCross(cc->rA[1], cc->normal);
		float rn2B = b2dCross(cc->rB[1], cc->normal);

		float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
		float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
		float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;

		const float k_maxConditionNumber = 1000.0f;
		if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12))
		{
			cc->k_exx = k11;
			cc->k_exy = k12;
			cc->k_eyx = k12;
			cc->k_eyy = k22;
			// b2Mat22::GetInverse (b2Math.h:205-217)
			float a = k11, b = k12, c = k12, d = k22;
			float det = a * d - b * c;
			if (det != 0.0f)
			{
				det = 1.0f / det;
			}
			cc->nm_exx = det * d;
			cc->nm_eyx = -det * b;
			cc->nm_exy = -det * c;
			cc->nm_eyy = det * a;
		}
		else
		{
			cc->pointCount = 1;
		}
	}
}

// b2ContactSolver::WarmStart (b2ContactSolver.cpp:253-291), one constraint
B2D_HD void b2dWarmStart(const ContactConstraint* cc, BodyVel* A, BodyVel* B)
{
	float mA = cc->invMassA, iA = cc->invIA, mB = cc->invMassB, iB = cc->invIB;
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	V2 normal = cc->normal;
	V2 tangent = b2dCrossVS(normal, 1.0f);
	B2D_FOR_POINTS(j, cc->pointCount)
	{
		V2 P = cc->normalImpulse[j] * normal + cc->tangentImpulse[j] * tangent;
		wA -= iA * b2dCross(cc->rA[j], P);
		vA -= mA * P;
		wB += iB * b2dCross(cc->rB[j], P);
		vB += mB * P;
	}
	A->v = vA;
	A->w = wA;
	B->v = vB;
	B->w = wB;
}

// b2ContactSolver::SolveVelocityConstraints (b2ContactSolver.cpp:293-603), one constraint
B2D_HD void b2dSolveVelocity(ContactConstraint* cc, BodyVel* A, BodyVel* B)
{
	float mA = cc->invMassA, iA = cc->invIA, mB = cc->invMassB, iB = cc->invIB;
	int pointCount = cc->pointCount;
	V2 vA = A->v, vB = B->v;
	float wA = A->w, wB = B->w;
	V2 normal = cc->normal;
	V2 tangent = b2dCrossVS(normal, 1.0f);
	float friction = cc->friction;

	B2D_FOR_POINTS(j, pointCount)
	{
		V2 dv = vB + b2dCrossSV(wB, cc->rB[j]) - vA - b2dCrossSV(wA, cc->rA[j]);
		float vt = b2dDot(dv, tangent) - cc->tangentSpeed;
		float lambda = cc->tangentMass[j] * (-vt);
		float maxFriction = friction * cc->normalImpulse[j];
		float newImpulse = b2dClamp(cc->tangentImpulse[j] + lambda, -maxFriction, maxFriction);
		lambda = newImpulse - cc->tangentImpulse[j];
		cc->tangentImpulse[j] = newImpulse;
		V2 P = lambda * tangent;
		vA -= mA * P;
		wA -= iA * b2dCross(cc->rA[j], P);
		vB += mB * P;
		wB += iB * b2dCross(cc->rB[j], P);
	}

	if (pointCount == 1)
	{
		V2 dv = vB + b2dCrossSV(wB, cc->rB[0]) - vA - b2dCrossSV(wA, cc->rA[0]);
		float vn = b2dDot(dv, normal);
		float lambda = -cc->normalMass[0] * (vn - cc->velocityBias[0]);
		float newImpulse = b2dMax(cc->normalImpulse[0] + lambda, 0.0f);
		lambda = newImpulse - cc->normalImpulse[0];
		cc->normalImpulse[0] = newImpulse;
		V2 P = lambda * normal;
		vA -= mA * P;
		wA -= iA * b2dCross(cc->rA[0], P);
		vB += mB * P;
		wB += iB * b2dCross(cc->rB[0], P);
	}
	else
	{
		// Block solver, total enumeration (b2ContactSolver.cpp:410-595)
		V2 a = v2(cc->normalImpulse[0], cc->normalImpulse[1]);
		V2 dv1 = vB + b2dCrossSV(wB, cc->rB[0]) - vA - b2dCrossSV(wA, cc->rA[0]);
		V2 dv2 = vB + b2dCrossSV(wB, cc->rB[1]) - vA - b2dCrossSV(wA, cc->rA[1]);
		float vn1 = b2dDot(dv1, normal);
		float vn2 = b2dDot(dv2, normal);
		V2 b;
		b.x = vn1 - cc->velocityBias[0];
		b.y = vn2 - cc->velocityBias[1];
		// b -= K * a   (b2Mul(Mat22, Vec2): ex.x*v.x + ey.x*v.y, ex.y*v.x + ey.y*v.y)
		b.x -= cc->k_exx * a.x + cc->k_eyx * a.y;
		b.y -= cc->k_exy * a.x + cc->k_eyy * a.y;

		V2 x;
		bool found = false;
		// Case 1: vn = 0
		{
			float mx = cc->nm_exx * b.x + cc->nm_eyx * b.y;
			float my = cc->nm_exy * b.x + cc->nm_eyy * b.y;
			x = v2(-mx, -my);
			if (x.x >= 0.0f && x.y >= 0.0f) found = true;
		}
		// Case 2: vn1 = 0 and x2 = 0
		if (!found)
		{
			x.x = -cc->normalMass[0] * b.x;
			x.y = 0.0f;
			vn2 = cc->k_exy * x.x + b.y;
			if (x.x >= 0.0f && vn2 >= 0.0f) found = true;
		}
		// Case 3: vn2 = 0 and x1 = 0
		if (!found)
		{
			x.x = 0.0f;
			x.y = -cc->normalMass[1] * b.y;
			vn1 = cc->k_eyx * x.y + b.x;
			if (x.y >= 0.0f && vn1 >= 0.0f) found = true;
		}
		// Case 4: x1 = 0 and x2 = 0
		if (!found)
		{
			x.x = 0.0f;
			x.y = 0.0f;
			vn1 = b.x;
			vn2 = b.y;
			if (vn1 >= 0.0f && vn2 >= 0.0f) found = true;
		}
		if (found)
		{
			V2 d = x - a;
			V2 P1 = d.x * normal;
			V2 P2 = d.y * normal;
			vA -= mA * (P1 + P2);
			wA -= iA * (b2dCross(cc->rA[0], P1) + b2dCross(cc->rA[1], P2));
			vB += mB * (P1 + P2);
			wB += iB * (b2dCross(cc->rB[0], P1) + b2dCross(cc->rB[1], P2));
			cc->normalImpulse[0] = x.x;
			cc->normalImpulse[1] = x.y;
		}
	}

	A->v = vA;
	A->w = wA;
	B->v = vB;
	B->w = wB;
}

// b2ContactSolver::SolvePositionConstraints (b2ContactSolver.cpp:676-752) with
// b2PositionSolverManifold::Initialize (:620-673), one constraint. Returns its min separation
// (starting from 0, as the reference's running minimum does).
// INLINE_ROT: expand the sin/cos pair in place instead of calling the out-of-line copy (see b2dRot).
template <bool INLINE_ROT = false>
B2D_HD float b2dSolvePosition(const ContactConstraint* cc, BodyPos* A, BodyPos* B, float baumgarte, float* minSepInOut)
{
	V2 localCenterA = cc->localCenterA, localCenterB = cc->localCenterB;
	float mA = cc->invMassA, iA = cc->invIA, mB = cc->invMassB, iB = cc->invIB;
	int pointCount = cc->pcPointCount;
	V2 cA = A->c, cB = B->c;
	float aA = A->a, aB = B->a;
	float minSeparation = *minSepInOut;

	B2D_FOR_POINTS(j, pointCount)
	{
		Xf xfA, xfB;
		xfA.q = INLINE_ROT ? b2dRotInline(aA) : b2dRot(aA);
		xfB.q = INLINE_ROT ? b2dRotInline(aB) : b2dRot(aB);
		xfA.p = cA - b2dMulRV(xfA.q, localCenterA);
		xfB.p = cB - b2dMulRV(xfB.q, localCenterB);

		V2 normal, point;
		float separation;
		if (cc->type == B2D_MANIFOLD_CIRCLES)
		{
			V2 pointA = b2dMulXV(xfA, cc->localPoint);
			V2 pointB = b2dMulXV(xfB, cc->localPoints[0]);
			normal = pointB - pointA;
			b2dNormalize(normal);
			point = 0.5f * (pointA + pointB);
			separation = b2dDot(pointB - pointA, normal) - cc->radiusA - cc->radiusB;
		}
		else
		{
			// e_faceA and e_faceB (b2ContactSolver.cpp:645-671) are one computation with the two bodies' roles exchanged:
			// selects instead of a branch (the rows of a wave are a mix of both)
			const bool faceA = cc->type == B2D_MANIFOLD_FACE_A;
			const Xf xfRef = faceA ? xfA : xfB;
			const Xf xfInc = faceA ? xfB : xfA;
			normal = b2dMulRV(xfRef.q, cc->localNormal);
			V2 planePoint = b2dMulXV(xfRef, cc->localPoint);
			V2 clipPoint = b2dMulXV(xfInc, cc->localPoints[j]);
			separation = b2dDot(clipPoint - planePoint, normal) - cc->radiusA - cc->radiusB;
			point = clipPoint;
			if (!faceA) normal = -normal;
		}

		V2 rA = point - cA;
		V2 rB = point - cB;

		minSeparation = b2dMin(minSeparation, separation);

		float C = b2dClamp(baumgarte * (separation + B2D_LINEAR_SLOP), -B2D_MAX_LINEAR_CORRECTION, 0.0f);

		float rnA = b2dCross(rA, normal);
		float rnB = b2dCross(rB, normal);
		float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;

		float impulse = K > 0.0f ? -C / K : 0.0f;
		V2 P = impulse * normal;

		cA -= mA * P;
		aA -= iA * b2dCross(rA, P);
		cB += mB * P;
		aB += iB * b2dCross(rB, P);
	}

	A->c = cA;
	A->a = aA;
	B->c = cB;
	B->a = aB;
	*minSepInOut = minSeparation;
	return minSeparation;
}

// b2Island::Solve step 1 (b2Island.cpp:209-224): dynamic bodies only.
B2D_HD void b2dIntegrateVelocity(V2* v, float* w, float h, V2 gravity, float gravityScale, float invMass, float invI,
	V2 force, float torque, float linearDamping, float angularDamping)
{
	V2 acc = gravityScale * gravity + invMass * force;
	*v += h * acc;
	*w += h * invI * torque;
	*v *= 1.0f / (1.0f + h * linearDamping);
	*w *= 1.0f / (1.0f + h * angularDamping);
}

// b2Island::Solve position integration (b2Island.cpp:283-313)
B2D_HD void b2dIntegratePosition(V2* c, float* a, V2* v, float* w, float h)
{
	V2 translation = h * (*v);
	if (b2dDot(translation, translation) > B2D_MAX_TRANSLATION_SQ)
	{
		float ratio = B2D_MAX_TRANSLATION / b2dLength(translation);
		*v *= ratio;
	}
	float rotation = h * (*w);
	if (rotation * rotation > B2D_MAX_ROTATION_SQ)
	{
		float ratio = B2D_MAX_ROTATION / b2dAbs(rotation);
		*w *= ratio;
	}
	*c += h * (*v);
	*a += h * (*w);
}

#endif
