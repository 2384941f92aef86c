/* b2o_joint.c - CPU oracle, joints: plain-C restatement of b2RevoluteJoint
 * (Box2D/Dynamics/Joints/b2RevoluteJoint.cpp:65-376; b2Mat33::Solve33/Solve22 b2Math.cpp:25-53) and of
 * b2DistanceJoint (Joints/b2DistanceJoint.cpp:65-225).
 * TEST INFRASTRUCTURE (see b2o.h). */
#include "b2o_joint.h"

static float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const float a[3], const float b[3], float out[3])
{
	out[0] = a[1] * b[2] - a[2] * b[1];
	out[1] = a[2] * b[0] - a[0] * b[2];
	out[2] = a[0] * b[1] - a[1] * b[0];
}

static void solve33(const revolute_t* j, const float b[3], float x[3])
{
	float c[3];
	cross3(j->ey, j->ez, c);
	float det = dot3(j->ex, c);
	if (det != 0.0f) det = 1.0f / det;
	x[0] = det * dot3(b, c);
	cross3(b, j->ez, c);
	x[1] = det * dot3(j->ex, c);
	cross3(j->ey, b, c);
	x[2] = det * dot3(j->ex, c);
}

static vec2 solve22(const revolute_t* j, vec2 b)
{
	float a11 = j->ex[0], a12 = j->ey[0], a21 = j->ex[1], a22 = j->ey[1];
	float det = a11 * a22 - a12 * a21;
	if (det != 0.0f) det = 1.0f / det;
	return v_make(det * (a22 * b.x - a12 * b.y), det * (a11 * b.y - a21 * b.x));
}

/* InitVelocityConstraints :65-182 */
void b2o_revolute_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	float aA, vec2* vA, float* wA, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	j->rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	j->rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	vec2 rA = j->rA, rB = j->rB;
	int fixedRotation = (iA + iB == 0.0f);
	j->ex[0] = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
	j->ey[0] = -rA.y * rA.x * iA - rB.y * rB.x * iB;
	j->ez[0] = -rA.y * iA - rB.y * iB;
	j->ex[1] = j->ey[0];
	j->ey[1] = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
	j->ez[1] = rA.x * iA + rB.x * iB;
	j->ex[2] = j->ez[0];
	j->ey[2] = j->ez[1];
	j->ez[2] = iA + iB;
	j->motorMass = iA + iB;
	if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
	if (!j->enableMotor || fixedRotation) j->motorImpulse = 0.0f;
	if (j->enableLimit && !fixedRotation)
	{
		float jointAngle = aB - aA - j->referenceAngle;
		if (f_abs(j->upperAngle - j->lowerAngle) < 2.0f * B2O_ANGULAR_SLOP) j->limitState = 3;
		else if (jointAngle <= j->lowerAngle)
		{
			if (j->limitState != 1) j->impulse[2] = 0.0f;
			j->limitState = 1;
		}
		else if (jointAngle >= j->upperAngle)
		{
			if (j->limitState != 2) j->impulse[2] = 0.0f;
			j->limitState = 2;
		}
		else
		{
			j->limitState = 0;
			j->impulse[2] = 0.0f;
		}
	}
	else
	{
		j->limitState = 0;
	}
	if (warmStarting)
	{
		j->impulse[0] *= dtRatio; j->impulse[1] *= dtRatio; j->impulse[2] *= dtRatio;
		j->motorImpulse *= dtRatio;
		vec2 P = v_make(j->impulse[0], j->impulse[1]);
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * (v_cross(rA, P) + j->motorImpulse + j->impulse[2]);
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * (v_cross(rB, P) + j->motorImpulse + j->impulse[2]);
	}
	else
	{
		j->impulse[0] = j->impulse[1] = j->impulse[2] = 0.0f;
		j->motorImpulse = 0.0f;
	}
}

/* SolveVelocityConstraints :184-290 */
void b2o_revolute_velocity(revolute_t* j, vec2* pvA, float* pwA, vec2* pvB, float* pwB, float dt)
{
	vec2 vA = *pvA, vB = *pvB;
	float wA = *pwA, wB = *pwB;
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	int fixedRotation = (iA + iB == 0.0f);
	vec2 rA = j->rA, rB = j->rB;
	if (j->enableMotor && j->limitState != 3 && !fixedRotation)
	{
		float Cdot = wB - wA - j->motorSpeed;
		float impulse = -j->motorMass * Cdot;
		float oldImpulse = j->motorImpulse;
		float maxImpulse = dt * j->maxMotorTorque;
		j->motorImpulse = f_clamp(j->motorImpulse + impulse, -maxImpulse, maxImpulse);
		impulse = j->motorImpulse - oldImpulse;
		wA -= iA * impulse;
		wB += iB * impulse;
	}
	if (j->enableLimit && j->limitState != 0 && !fixedRotation)
	{
		vec2 Cdot1 = v_sub(v_sub(v_add(vB, v_cross_sv(wB, rB)), vA), v_cross_sv(wA, rA));
		float Cdot2 = wB - wA;
		float Cdot[3] = { Cdot1.x, Cdot1.y, Cdot2 };
		float sol[3], impulse[3];
		solve33(j, Cdot, sol);
		impulse[0] = -sol[0]; impulse[1] = -sol[1]; impulse[2] = -sol[2];
		if (j->limitState == 3)
		{
			j->impulse[0] += impulse[0]; j->impulse[1] += impulse[1]; j->impulse[2] += impulse[2];
		}
		else
		{
			float newImpulse = j->impulse[2] + impulse[2];
			int clampIt = (j->limitState == 1) ? (newImpulse < 0.0f) : (newImpulse > 0.0f);
			if (clampIt)
			{
				vec2 rhs = v_add(v_neg(Cdot1), v_scale(j->impulse[2], v_make(j->ez[0], j->ez[1])));
				vec2 reduced = solve22(j, rhs);
				impulse[0] = reduced.x;
				impulse[1] = reduced.y;
				impulse[2] = -j->impulse[2];
				j->impulse[0] += reduced.x;
				j->impulse[1] += reduced.y;
				j->impulse[2] = 0.0f;
			}
			else
			{
				j->impulse[0] += impulse[0]; j->impulse[1] += impulse[1]; j->impulse[2] += impulse[2];
			}
		}
		vec2 P = v_make(impulse[0], impulse[1]);
		vA = v_sub(vA, v_scale(mA, P));
		wA -= iA * (v_cross(rA, P) + impulse[2]);
		vB = v_add(vB, v_scale(mB, P));
		wB += iB * (v_cross(rB, P) + impulse[2]);
	}
	else
	{
		vec2 Cdot = v_sub(v_sub(v_add(vB, v_cross_sv(wB, rB)), vA), v_cross_sv(wA, rA));
		vec2 impulse = solve22(j, v_neg(Cdot));
		j->impulse[0] += impulse.x;
		j->impulse[1] += impulse.y;
		vA = v_sub(vA, v_scale(mA, impulse));
		wA -= iA * v_cross(rA, impulse);
		vB = v_add(vB, v_scale(mB, impulse));
		wB += iB * v_cross(rB, impulse);
	}
	*pvA = vA; *pwA = wA; *pvB = vB; *pwB = wB;
}

/* SolvePositionConstraints :292-376 */
int b2o_revolute_position(const revolute_t* j, vec2* pcA, float* paA, vec2* pcB, float* paB)
{
	vec2 cA = *pcA, cB = *pcB;
	float aA = *paA, aB = *paB;
	float angularError = 0.0f, positionError = 0.0f;
	int fixedRotation = (j->invIA + j->invIB == 0.0f);
	const float maxAng = 8.0f / 180.0f * B2O_PI;
	if (j->enableLimit && j->limitState != 0 && !fixedRotation)
	{
		float angle = aB - aA - j->referenceAngle;
		float limitImpulse = 0.0f;
		if (j->limitState == 3)
		{
			float C = f_clamp(angle - j->lowerAngle, -maxAng, maxAng);
			limitImpulse = -j->motorMass * C;
			angularError = f_abs(C);
		}
		else if (j->limitState == 1)
		{
			float C = angle - j->lowerAngle;
			angularError = -C;
			C = f_clamp(C + B2O_ANGULAR_SLOP, -maxAng, 0.0f);
			limitImpulse = -j->motorMass * C;
		}
		else if (j->limitState == 2)
		{
			float C = angle - j->upperAngle;
			angularError = C;
			C = f_clamp(C - B2O_ANGULAR_SLOP, 0.0f, maxAng);
			limitImpulse = -j->motorMass * C;
		}
		aA -= j->invIA * limitImpulse;
		aB += j->invIB * limitImpulse;
	}
	{
		rot qA = r_make(aA), qB = r_make(aB);
		vec2 rA = r_mul(qA, v_sub(j->localAnchorA, j->localCenterA));
		vec2 rB = r_mul(qB, v_sub(j->localAnchorB, j->localCenterB));
		vec2 C = v_sub(v_sub(v_add(cB, rB), cA), rA);
		positionError = v_length(C);
		float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
		float kexx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
		float kexy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
		float keyx = kexy;
		float keyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
		float a11 = kexx, a12 = keyx, a21 = kexy, a22 = keyy;
		float det = a11 * a22 - a12 * a21;
		if (det != 0.0f) det = 1.0f / det;
		vec2 sol = v_make(det * (a22 * C.x - a12 * C.y), det * (a11 * C.y - a21 * C.x));
		vec2 impulse = v_neg(sol);
		cA = v_sub(cA, v_scale(mA, impulse));
		aA -= iA * v_cross(rA, impulse);
		cB = v_add(cB, v_scale(mB, impulse));
		aB += iB * v_cross(rB, impulse);
	}
	*pcA = cA; *paA = aA; *pcB = cB; *paB = aB;
	return positionError <= B2O_LINEAR_SLOP && angularError <= B2O_ANGULAR_SLOP;
}

/* ---- distance joint ---------------------------------------------------------------------------- */
/* InitVelocityConstraints b2DistanceJoint.cpp:65-157 */
void b2o_distance_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio, float dt)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	j->rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	j->rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	j->u = v_sub(v_sub(v_add(cB, j->rB), cA), j->rA);
	float length = v_length(j->u);
	if (length > B2O_LINEAR_SLOP) j->u = v_scale(1.0f / length, j->u);
	else j->u = v_make(0.0f, 0.0f);
	float crAu = v_cross(j->rA, j->u);
	float crBu = v_cross(j->rB, j->u);
	float invMass = mA + iA * crAu * crAu + mB + iB * crBu * crBu;
	j->mass = invMass != 0.0f ? 1.0f / invMass : 0.0f;
	if (j->frequencyHz > 0.0f)
	{
		float C = length - j->length;
		float omega = 2.0f * B2O_PI * j->frequencyHz;
		float d = 2.0f * j->mass * j->dampingRatio * omega;
		float k = j->mass * omega * omega;
		j->gamma = dt * (d + dt * k);
		j->gamma = j->gamma != 0.0f ? 1.0f / j->gamma : 0.0f;
		j->bias = C * dt * k * j->gamma;
		invMass += j->gamma;
		j->mass = invMass != 0.0f ? 1.0f / invMass : 0.0f;
	}
	else
	{
		j->gamma = 0.0f;
		j->bias = 0.0f;
	}
	if (warmStarting)
	{
		j->impulse[0] *= dtRatio;
		vec2 P = v_scale(j->impulse[0], j->u);
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * v_cross(j->rA, P);
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * v_cross(j->rB, P);
	}
	else
	{
		j->impulse[0] = 0.0f;
	}
}

/* SolveVelocityConstraints :159-184 */
void b2o_distance_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB)
{
	vec2 vpA = v_add(*vA, v_cross_sv(*wA, j->rA));
	vec2 vpB = v_add(*vB, v_cross_sv(*wB, j->rB));
	float Cdot = v_dot(j->u, v_sub(vpB, vpA));
	float impulse = -j->mass * (Cdot + j->bias + j->gamma * j->impulse[0]);
	j->impulse[0] += impulse;
	vec2 P = v_scale(impulse, j->u);
	*vA = v_sub(*vA, v_scale(j->invMassA, P));
	*wA -= j->invIA * v_cross(j->rA, P);
	*vB = v_add(*vB, v_scale(j->invMassB, P));
	*wB += j->invIB * v_cross(j->rB, P);
}

/* SolvePositionConstraints :186-225 */
int b2o_distance_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB)
{
	if (j->frequencyHz > 0.0f) return 1;
	rot qA = r_make(*aA), qB = r_make(*aB);
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, j->localCenterA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, j->localCenterB));
	vec2 u = v_sub(v_sub(v_add(*cB, rB), *cA), rA);
	float length = v_normalize(&u);
	float C = length - j->length;
	C = f_clamp(C, -B2O_MAX_LINEAR_CORRECTION, B2O_MAX_LINEAR_CORRECTION);
	float impulse = -j->mass * C;
	vec2 P = v_scale(impulse, u);
	*cA = v_sub(*cA, v_scale(j->invMassA, P));
	*aA -= j->invIA * v_cross(rA, P);
	*cB = v_add(*cB, v_scale(j->invMassB, P));
	*aB += j->invIB * v_cross(rB, P);
	return f_abs(C) < B2O_LINEAR_SLOP;
}
