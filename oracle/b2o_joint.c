/* b2o_joint.c - CPU oracle, joints: plain-C restatement of b2RevoluteJoint
 * (Box2D/Dynamics/Joints/b2RevoluteJoint.cpp:65-376; b2Mat33::Solve33/Solve22 b2Math.cpp:25-53) and of
 * b2DistanceJoint (Joints/b2DistanceJoint.cpp:65-225), b2PrismaticJoint (Joints/b2PrismaticJoint.cpp:130-478),
 * b2WeldJoint (Joints/b2WeldJoint.cpp:58-303), b2WheelJoint (Joints/b2WheelJoint.cpp:79-292), b2RopeJoint
 * (Joints/b2RopeJoint.cpp:48-182), b2FrictionJoint (Joints/b2FrictionJoint.cpp:58-185), b2MotorJoint (Joints/b2MotorJoint.cpp:62-203),
 * b2PulleyJoint (Joints/b2PulleyJoint.cpp:81-253), b2MouseJoint (Joints/b2MouseJoint.cpp:99-198),
 * b2GearJoint (Joints/b2GearJoint.cpp:131-390); b2Mat33::GetInverse22 / GetSymInverse33 b2Math.cpp:56-94.
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

/* ---- 3x3 helpers on column arrays (b2Mat33, b2Math.cpp:25-94) ------------------------------------ */
static void m_solve33(const float ex[3], const float ey[3], const float ez[3], const float b[3], float x[3])
{
	float c[3];
	cross3(ey, ez, c);
	float det = dot3(ex, c);
	if (det != 0.0f) det = 1.0f / det;
	x[0] = det * dot3(b, c);
	cross3(b, ez, c);
	x[1] = det * dot3(ex, c);
	cross3(ey, b, c);
	x[2] = det * dot3(ex, c);
}

static vec2 m_solve22(const float ex[3], const float ey[3], vec2 b)
{
	float a11 = ex[0], a12 = ey[0], a21 = ex[1], a22 = ey[1];
	float det = a11 * a22 - a12 * a21;
	if (det != 0.0f) det = 1.0f / det;
	return v_make(det * (a22 * b.x - a12 * b.y), det * (a11 * b.y - a21 * b.x));
}

static void m_inverse22(const float ex[3], const float ey[3], float ox[3], float oy[3], float oz[3])
{
	float a = ex[0], b = ey[0], c = ex[1], d = ey[1];
	float det = a * d - b * c;
	if (det != 0.0f) det = 1.0f / det;
	ox[0] = det * d; oy[0] = -det * b; ox[2] = 0.0f;
	ox[1] = -det * c; oy[1] = det * a; oy[2] = 0.0f;
	oz[0] = 0.0f; oz[1] = 0.0f; oz[2] = 0.0f;
}

static void m_sym_inverse33(const float ex[3], const float ey[3], const float ez[3], float ox[3], float oy[3], float oz[3])
{
	float c[3];
	cross3(ey, ez, c);
	float det = dot3(ex, c);
	if (det != 0.0f) det = 1.0f / det;
	float a11 = ex[0], a12 = ey[0], a13 = ez[0];
	float a22 = ey[1], a23 = ez[1];
	float a33 = ez[2];
	ox[0] = det * (a22 * a33 - a23 * a23);
	ox[1] = det * (a13 * a23 - a12 * a33);
	ox[2] = det * (a12 * a23 - a13 * a22);
	oy[0] = ox[1];
	oy[1] = det * (a11 * a33 - a13 * a13);
	oy[2] = det * (a13 * a12 - a11 * a23);
	oz[0] = ox[2];
	oz[1] = oy[2];
	oz[2] = det * (a11 * a22 - a12 * a12);
}

/* ---- prismatic joint ---------------------------------------------------------------------------- */
static void prismatic_k(float mA, float mB, float iA, float iB, float s1, float s2, float a1, float a2,
	float ex[3], float ey[3], float ez[3])
{
	float k11 = mA + mB + iA * s1 * s1 + iB * s2 * s2;
	float k12 = iA * s1 + iB * s2;
	float k13 = iA * s1 * a1 + iB * s2 * a2;
	float k22 = iA + iB;
	if (k22 == 0.0f) k22 = 1.0f;
	float k23 = iA * a1 + iB * a2;
	float k33 = mA + mB + iA * a1 * a1 + iB * a2 * a2;
	ex[0] = k11; ex[1] = k12; ex[2] = k13;
	ey[0] = k12; ey[1] = k22; ey[2] = k23;
	ez[0] = k13; ez[1] = k23; ez[2] = k33;
}

/* InitVelocityConstraints b2PrismaticJoint.cpp:130-257 */
void b2o_prismatic_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	vec2 d = v_sub(v_add(v_sub(cB, cA), rB), rA);
	j->axis = r_mul(qA, j->localXAxisA);
	j->a1 = v_cross(v_add(d, rA), j->axis);
	j->a2 = v_cross(rB, j->axis);
	j->motorMass = mA + mB + iA * j->a1 * j->a1 + iB * j->a2 * j->a2;
	if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
	j->perp = r_mul(qA, j->localYAxisA);
	j->s1 = v_cross(v_add(d, rA), j->perp);
	j->s2 = v_cross(rB, j->perp);
	prismatic_k(mA, mB, iA, iB, j->s1, j->s2, j->a1, j->a2, j->ex, j->ey, j->ez);
	if (j->enableLimit)
	{
		float jointTranslation = v_dot(j->axis, d);
		if (f_abs(j->upperAngle - j->lowerAngle) < 2.0f * B2O_LINEAR_SLOP)
		{
			j->limitState = 3;
		}
		else if (jointTranslation <= j->lowerAngle)
		{
			if (j->limitState != 1) { j->limitState = 1; j->impulse[2] = 0.0f; }
		}
		else if (jointTranslation >= j->upperAngle)
		{
			if (j->limitState != 2) { j->limitState = 2; j->impulse[2] = 0.0f; }
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
		j->impulse[2] = 0.0f;
	}
	if (!j->enableMotor) j->motorImpulse = 0.0f;
	if (warmStarting)
	{
		j->impulse[0] *= dtRatio; j->impulse[1] *= dtRatio; j->impulse[2] *= dtRatio;
		j->motorImpulse *= dtRatio;
		vec2 P = v_add(v_scale(j->impulse[0], j->perp), v_scale(j->motorImpulse + j->impulse[2], j->axis));
		float LA = j->impulse[0] * j->s1 + j->impulse[1] + (j->motorImpulse + j->impulse[2]) * j->a1;
		float LB = j->impulse[0] * j->s2 + j->impulse[1] + (j->motorImpulse + j->impulse[2]) * j->a2;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * LA;
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * LB;
	}
	else
	{
		j->impulse[0] = j->impulse[1] = j->impulse[2] = 0.0f;
		j->motorImpulse = 0.0f;
	}
}

/* SolveVelocityConstraints :259-350 */
void b2o_prismatic_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt)
{
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	if (j->enableMotor && j->limitState != 3)
	{
		float Cdot = v_dot(j->axis, v_sub(*vB, *vA)) + j->a2 * *wB - j->a1 * *wA;
		float impulse = j->motorMass * (j->motorSpeed - Cdot);
		float oldImpulse = j->motorImpulse;
		float maxImpulse = dt * j->maxMotorTorque;
		j->motorImpulse = f_clamp(j->motorImpulse + impulse, -maxImpulse, maxImpulse);
		impulse = j->motorImpulse - oldImpulse;
		vec2 P = v_scale(impulse, j->axis);
		float LA = impulse * j->a1;
		float LB = impulse * j->a2;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * LA;
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * LB;
	}
	vec2 Cdot1;
	Cdot1.x = v_dot(j->perp, v_sub(*vB, *vA)) + j->s2 * *wB - j->s1 * *wA;
	Cdot1.y = *wB - *wA;
	if (j->enableLimit && j->limitState != 0)
	{
		float Cdot2 = v_dot(j->axis, v_sub(*vB, *vA)) + j->a2 * *wB - j->a1 * *wA;
		float nC[3] = { -Cdot1.x, -Cdot1.y, -Cdot2 };
		float f1[3] = { j->impulse[0], j->impulse[1], j->impulse[2] };
		float df[3];
		m_solve33(j->ex, j->ey, j->ez, nC, df);
		j->impulse[0] += df[0]; j->impulse[1] += df[1]; j->impulse[2] += df[2];
		if (j->limitState == 1) j->impulse[2] = f_max(j->impulse[2], 0.0f);
		else if (j->limitState == 2) j->impulse[2] = f_min(j->impulse[2], 0.0f);
		vec2 b = v_sub(v_neg(Cdot1), v_scale(j->impulse[2] - f1[2], v_make(j->ez[0], j->ez[1])));
		vec2 f2r = v_add(m_solve22(j->ex, j->ey, b), v_make(f1[0], f1[1]));
		j->impulse[0] = f2r.x;
		j->impulse[1] = f2r.y;
		df[0] = j->impulse[0] - f1[0]; df[1] = j->impulse[1] - f1[1]; df[2] = j->impulse[2] - f1[2];
		vec2 P = v_add(v_scale(df[0], j->perp), v_scale(df[2], j->axis));
		float LA = df[0] * j->s1 + df[1] + df[2] * j->a1;
		float LB = df[0] * j->s2 + df[1] + df[2] * j->a2;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * LA;
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * LB;
	}
	else
	{
		vec2 df = m_solve22(j->ex, j->ey, v_neg(Cdot1));
		j->impulse[0] += df.x;
		j->impulse[1] += df.y;
		vec2 P = v_scale(df.x, j->perp);
		float LA = df.x * j->s1 + df.y;
		float LB = df.x * j->s2 + df.y;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * LA;
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * LB;
	}
}

/* SolvePositionConstraints :352-478 */
int b2o_prismatic_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB)
{
	rot qA = r_make(*aA), qB = r_make(*aB);
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, j->localCenterA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, j->localCenterB));
	vec2 d = v_sub(v_sub(v_add(*cB, rB), *cA), rA);
	vec2 axis = r_mul(qA, j->localXAxisA);
	float a1 = v_cross(v_add(d, rA), axis);
	float a2 = v_cross(rB, axis);
	vec2 perp = r_mul(qA, j->localYAxisA);
	float s1 = v_cross(v_add(d, rA), perp);
	float s2 = v_cross(rB, perp);
	float impulse[3];
	vec2 C1;
	C1.x = v_dot(perp, d);
	C1.y = *aB - *aA - j->referenceAngle;
	float linearError = f_abs(C1.x);
	float angularError = f_abs(C1.y);
	int active = 0;
	float C2 = 0.0f;
	if (j->enableLimit)
	{
		float translation = v_dot(axis, d);
		if (f_abs(j->upperAngle - j->lowerAngle) < 2.0f * B2O_LINEAR_SLOP)
		{
			C2 = f_clamp(translation, -B2O_MAX_LINEAR_CORRECTION, B2O_MAX_LINEAR_CORRECTION);
			linearError = f_max(linearError, f_abs(translation));
			active = 1;
		}
		else if (translation <= j->lowerAngle)
		{
			C2 = f_clamp(translation - j->lowerAngle + B2O_LINEAR_SLOP, -B2O_MAX_LINEAR_CORRECTION, 0.0f);
			linearError = f_max(linearError, j->lowerAngle - translation);
			active = 1;
		}
		else if (translation >= j->upperAngle)
		{
			C2 = f_clamp(translation - j->upperAngle - B2O_LINEAR_SLOP, 0.0f, B2O_MAX_LINEAR_CORRECTION);
			linearError = f_max(linearError, translation - j->upperAngle);
			active = 1;
		}
	}
	float ex[3], ey[3], ez[3];
	prismatic_k(mA, mB, iA, iB, s1, s2, a1, a2, ex, ey, ez);
	if (active)
	{
		float nC[3] = { -C1.x, -C1.y, -C2 };
		m_solve33(ex, ey, ez, nC, impulse);
	}
	else
	{
		vec2 i1 = m_solve22(ex, ey, v_neg(C1)); /* b2Mat22::Solve has the same form (b2Math.h:221-233) */
		impulse[0] = i1.x; impulse[1] = i1.y; impulse[2] = 0.0f;
	}
	vec2 P = v_add(v_scale(impulse[0], perp), v_scale(impulse[2], axis));
	float LA = impulse[0] * s1 + impulse[1] + impulse[2] * a1;
	float LB = impulse[0] * s2 + impulse[1] + impulse[2] * a2;
	*cA = v_sub(*cA, v_scale(mA, P));
	*aA -= iA * LA;
	*cB = v_add(*cB, v_scale(mB, P));
	*aB += iB * LB;
	return linearError <= B2O_LINEAR_SLOP && angularError <= B2O_ANGULAR_SLOP;
}

/* ---- weld joint --------------------------------------------------------------------------------- */
static void weld_k(vec2 rA, vec2 rB, float mA, float mB, float iA, float iB, float ex[3], float ey[3], float ez[3])
{
	ex[0] = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
	ey[0] = -rA.y * rA.x * iA - rB.y * rB.x * iB;
	ez[0] = -rA.y * iA - rB.y * iB;
	ex[1] = ey[0];
	ey[1] = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
	ez[1] = rA.x * iA + rB.x * iB;
	ex[2] = ez[0];
	ey[2] = ez[1];
	ez[2] = iA + iB;
}

/* InitVelocityConstraints b2WeldJoint.cpp:58-155 */
void b2o_weld_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	float aA, vec2* vA, float* wA, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio, float dt)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	j->rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	j->rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	float kx[3], ky[3], kz[3];
	weld_k(j->rA, j->rB, mA, mB, iA, iB, kx, ky, kz);
	if (j->frequencyHz > 0.0f)
	{
		m_inverse22(kx, ky, j->ex, j->ey, j->ez);
		float invM = iA + iB;
		float m = invM > 0.0f ? 1.0f / invM : 0.0f;
		float C = aB - aA - j->referenceAngle;
		float omega = 2.0f * B2O_PI * j->frequencyHz;
		float d = 2.0f * m * j->dampingRatio * omega;
		float k = m * omega * omega;
		j->gamma = dt * (d + dt * k);
		j->gamma = j->gamma != 0.0f ? 1.0f / j->gamma : 0.0f;
		j->bias = C * dt * k * j->gamma;
		invM += j->gamma;
		j->ez[2] = invM != 0.0f ? 1.0f / invM : 0.0f;
	}
	else if (kz[2] == 0.0f)
	{
		m_inverse22(kx, ky, j->ex, j->ey, j->ez);
		j->gamma = 0.0f;
		j->bias = 0.0f;
	}
	else
	{
		m_sym_inverse33(kx, ky, kz, j->ex, j->ey, j->ez);
		j->gamma = 0.0f;
		j->bias = 0.0f;
	}
	if (warmStarting)
	{
		j->impulse[0] *= dtRatio; j->impulse[1] *= dtRatio; j->impulse[2] *= dtRatio;
		vec2 P = v_make(j->impulse[0], j->impulse[1]);
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * (v_cross(j->rA, P) + j->impulse[2]);
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * (v_cross(j->rB, P) + j->impulse[2]);
	}
	else
	{
		j->impulse[0] = j->impulse[1] = j->impulse[2] = 0.0f;
	}
}

/* SolveVelocityConstraints :157-216 */
void b2o_weld_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB)
{
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	if (j->frequencyHz > 0.0f)
	{
		float Cdot2 = *wB - *wA;
		float impulse2 = -j->ez[2] * (Cdot2 + j->bias + j->gamma * j->impulse[2]);
		j->impulse[2] += impulse2;
		*wA -= iA * impulse2;
		*wB += iB * impulse2;
		vec2 Cdot1 = v_sub(v_sub(v_add(*vB, v_cross_sv(*wB, j->rB)), *vA), v_cross_sv(*wA, j->rA));
		/* b2Mul22 b2Math.h:521-524 */
		vec2 impulse1 = v_neg(v_make(j->ex[0] * Cdot1.x + j->ey[0] * Cdot1.y, j->ex[1] * Cdot1.x + j->ey[1] * Cdot1.y));
		j->impulse[0] += impulse1.x;
		j->impulse[1] += impulse1.y;
		vec2 P = impulse1;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * v_cross(j->rA, P);
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * v_cross(j->rB, P);
	}
	else
	{
		vec2 Cdot1 = v_sub(v_sub(v_add(*vB, v_cross_sv(*wB, j->rB)), *vA), v_cross_sv(*wA, j->rA));
		float Cdot2 = *wB - *wA;
		/* b2Mul(b2Mat33, b2Vec3) b2Math.h:515-518 */
		float impulse[3];
		for (int k = 0; k < 3; ++k) impulse[k] = -(Cdot1.x * j->ex[k] + Cdot1.y * j->ey[k] + Cdot2 * j->ez[k]);
		j->impulse[0] += impulse[0]; j->impulse[1] += impulse[1]; j->impulse[2] += impulse[2];
		vec2 P = v_make(impulse[0], impulse[1]);
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * (v_cross(j->rA, P) + impulse[2]);
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * (v_cross(j->rB, P) + impulse[2]);
	}
}

/* SolvePositionConstraints :218-303 */
int b2o_weld_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB)
{
	rot qA = r_make(*aA), qB = r_make(*aB);
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, j->localCenterA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, j->localCenterB));
	float positionError, angularError;
	float kx[3], ky[3], kz[3];
	weld_k(rA, rB, mA, mB, iA, iB, kx, ky, kz);
	vec2 C1 = v_sub(v_sub(v_add(*cB, rB), *cA), rA);
	if (j->frequencyHz > 0.0f)
	{
		positionError = v_length(C1);
		angularError = 0.0f;
		vec2 P = v_neg(m_solve22(kx, ky, C1));
		*cA = v_sub(*cA, v_scale(mA, P));
		*aA -= iA * v_cross(rA, P);
		*cB = v_add(*cB, v_scale(mB, P));
		*aB += iB * v_cross(rB, P);
	}
	else
	{
		float C2 = *aB - *aA - j->referenceAngle;
		positionError = v_length(C1);
		angularError = f_abs(C2);
		float impulse[3];
		if (kz[2] > 0.0f)
		{
			float C[3] = { C1.x, C1.y, C2 };
			m_solve33(kx, ky, kz, C, impulse);
			impulse[0] = -impulse[0]; impulse[1] = -impulse[1]; impulse[2] = -impulse[2];
		}
		else
		{
			vec2 i2 = v_neg(m_solve22(kx, ky, C1));
			impulse[0] = i2.x; impulse[1] = i2.y; impulse[2] = 0.0f;
		}
		vec2 P = v_make(impulse[0], impulse[1]);
		*cA = v_sub(*cA, v_scale(mA, P));
		*aA -= iA * (v_cross(rA, P) + impulse[2]);
		*cB = v_add(*cB, v_scale(mB, P));
		*aB += iB * (v_cross(rB, P) + impulse[2]);
	}
	return positionError <= B2O_LINEAR_SLOP && angularError <= B2O_ANGULAR_SLOP;
}

/* ---- wheel joint -------------------------------------------------------------------------------- */
/* InitVelocityConstraints b2WheelJoint.cpp:79-198 */
void b2o_wheel_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio, float dt)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	vec2 d = v_sub(v_sub(v_add(cB, rB), cA), rA);
	j->perp = r_mul(qA, j->localYAxisA);
	j->a1 = v_cross(v_add(d, rA), j->perp);
	j->a2 = v_cross(rB, j->perp);
	j->mass = mA + mB + iA * j->a1 * j->a1 + iB * j->a2 * j->a2;
	if (j->mass > 0.0f) j->mass = 1.0f / j->mass;
	j->springMass = 0.0f;
	j->bias = 0.0f;
	j->gamma = 0.0f;
	if (j->frequencyHz > 0.0f)
	{
		j->axis = r_mul(qA, j->localXAxisA);
		j->s1 = v_cross(v_add(d, rA), j->axis);
		j->s2 = v_cross(rB, j->axis);
		float invMass = mA + mB + iA * j->s1 * j->s1 + iB * j->s2 * j->s2;
		if (invMass > 0.0f)
		{
			j->springMass = 1.0f / invMass;
			float C = v_dot(d, j->axis);
			float omega = 2.0f * B2O_PI * j->frequencyHz;
			float damp = 2.0f * j->springMass * j->dampingRatio * omega;
			float k = j->springMass * omega * omega;
			j->gamma = dt * (damp + dt * k);
			if (j->gamma > 0.0f) j->gamma = 1.0f / j->gamma;
			j->bias = C * dt * k * j->gamma;
			j->springMass = invMass + j->gamma;
			if (j->springMass > 0.0f) j->springMass = 1.0f / j->springMass;
		}
	}
	else
	{
		j->springImpulse = 0.0f;
	}
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
		j->impulse[0] *= dtRatio;
		j->springImpulse *= dtRatio;
		j->motorImpulse *= dtRatio;
		vec2 P = v_add(v_scale(j->impulse[0], j->perp), v_scale(j->springImpulse, j->axis));
		float LA = j->impulse[0] * j->a1 + j->springImpulse * j->s1 + j->motorImpulse;
		float LB = j->impulse[0] * j->a2 + j->springImpulse * j->s2 + j->motorImpulse;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * LA;
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * LB;
	}
	else
	{
		j->impulse[0] = 0.0f;
		j->springImpulse = 0.0f;
		j->motorImpulse = 0.0f;
	}
}

/* SolveVelocityConstraints :200-257 */
void b2o_wheel_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt)
{
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	{
		float Cdot = v_dot(j->axis, v_sub(*vB, *vA)) + j->s2 * *wB - j->s1 * *wA;
		float impulse = -j->springMass * (Cdot + j->bias + j->gamma * j->springImpulse);
		j->springImpulse += impulse;
		vec2 P = v_scale(impulse, j->axis);
		float LA = impulse * j->s1;
		float LB = impulse * j->s2;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * LA;
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * LB;
	}
	{
		float Cdot = *wB - *wA - j->motorSpeed;
		float impulse = -j->motorMass * Cdot;
		float oldImpulse = j->motorImpulse;
		float maxImpulse = dt * j->maxMotorTorque;
		j->motorImpulse = f_clamp(j->motorImpulse + impulse, -maxImpulse, maxImpulse);
		impulse = j->motorImpulse - oldImpulse;
		*wA -= iA * impulse;
		*wB += iB * impulse;
	}
	{
		float Cdot = v_dot(j->perp, v_sub(*vB, *vA)) + j->a2 * *wB - j->a1 * *wA;
		float impulse = -j->mass * Cdot;
		j->impulse[0] += impulse;
		vec2 P = v_scale(impulse, j->perp);
		float LA = impulse * j->a1;
		float LB = impulse * j->a2;
		*vA = v_sub(*vA, v_scale(mA, P));
		*wA -= iA * LA;
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * LB;
	}
}

/* SolvePositionConstraints :259-292 (k is built from the stored m_sAy / m_sBy, as there) */
int b2o_wheel_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB)
{
	rot qA = r_make(*aA), qB = r_make(*aB);
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, j->localCenterA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, j->localCenterB));
	vec2 d = v_sub(v_add(v_sub(*cB, *cA), rB), rA);
	vec2 ay = r_mul(qA, j->localYAxisA);
	float sAy = v_cross(v_add(d, rA), ay);
	float sBy = v_cross(rB, ay);
	float C = v_dot(d, ay);
	float k = j->invMassA + j->invMassB + j->invIA * j->a1 * j->a1 + j->invIB * j->a2 * j->a2;
	float impulse = k != 0.0f ? -C / k : 0.0f;
	vec2 P = v_scale(impulse, ay);
	float LA = impulse * sAy;
	float LB = impulse * sBy;
	*cA = v_sub(*cA, v_scale(j->invMassA, P));
	*aA -= j->invIA * LA;
	*cB = v_add(*cB, v_scale(j->invMassB, P));
	*aB += j->invIB * LB;
	return f_abs(C) <= B2O_LINEAR_SLOP;
}

/* ---- rope joint --------------------------------------------------------------------------------- */
/* InitVelocityConstraints b2RopeJoint.cpp:48-115 */
void b2o_rope_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	j->rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	j->rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	j->u = v_sub(v_sub(v_add(cB, j->rB), cA), j->rA);
	j->curLength = v_length(j->u);
	float C = j->curLength - j->length;
	j->limitState = C > 0.0f ? 2 : 0;
	if (j->curLength > B2O_LINEAR_SLOP)
	{
		j->u = v_scale(1.0f / j->curLength, j->u);
	}
	else
	{
		j->u = v_make(0.0f, 0.0f);
		j->mass = 0.0f;
		j->impulse[0] = 0.0f;
		return;
	}
	float crA = v_cross(j->rA, j->u);
	float crB = v_cross(j->rB, j->u);
	float invMass = mA + iA * crA * crA + mB + iB * crB * crB;
	j->mass = invMass != 0.0f ? 1.0f / invMass : 0.0f;
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

/* SolveVelocityConstraints :117-149 */
void b2o_rope_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float inv_dt)
{
	vec2 vpA = v_add(*vA, v_cross_sv(*wA, j->rA));
	vec2 vpB = v_add(*vB, v_cross_sv(*wB, j->rB));
	float C = j->curLength - j->length;
	float Cdot = v_dot(j->u, v_sub(vpB, vpA));
	if (C < 0.0f) Cdot += inv_dt * C;
	float impulse = -j->mass * Cdot;
	float oldImpulse = j->impulse[0];
	j->impulse[0] = f_min(0.0f, j->impulse[0] + impulse);
	impulse = j->impulse[0] - oldImpulse;
	vec2 P = v_scale(impulse, j->u);
	*vA = v_sub(*vA, v_scale(j->invMassA, P));
	*wA -= j->invIA * v_cross(j->rA, P);
	*vB = v_add(*vB, v_scale(j->invMassB, P));
	*wB += j->invIB * v_cross(j->rB, P);
}

/* SolvePositionConstraints :151-182 */
int b2o_rope_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB)
{
	rot qA = r_make(*aA), qB = r_make(*aB);
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, j->localCenterA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, j->localCenterB));
	vec2 u = v_sub(v_sub(v_add(*cB, rB), *cA), rA);
	float length = v_normalize(&u);
	float C = length - j->length;
	C = f_clamp(C, 0.0f, B2O_MAX_LINEAR_CORRECTION);
	float impulse = -j->mass * C;
	vec2 P = v_scale(impulse, u);
	*cA = v_sub(*cA, v_scale(j->invMassA, P));
	*aA -= j->invIA * v_cross(rA, P);
	*cB = v_add(*cB, v_scale(j->invMassB, P));
	*aB += j->invIB * v_cross(rB, P);
	return length - j->length < B2O_LINEAR_SLOP;
}

/* ---- friction / motor joints -------------------------------------------------------------------- */
/* K and its inverse (b2FrictionJoint.cpp:86-100 = b2MotorJoint.cpp:90-104; b2Mat22::GetInverse b2Math.h:205-217) */
static void linear_angular_mass(revolute_t* j)
{
	float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
	float kxx = mA + mB + iA * j->rA.y * j->rA.y + iB * j->rB.y * j->rB.y;
	float kxy = -iA * j->rA.x * j->rA.y - iB * j->rB.x * j->rB.y;
	float kyy = mA + mB + iA * j->rA.x * j->rA.x + iB * j->rB.x * j->rB.x;
	float a = kxx, b = kxy, c = kxy, d = kyy;
	float det = a * d - b * c;
	if (det != 0.0f) det = 1.0f / det;
	j->linearMass[0] = det * d; j->linearMass[2] = -det * b;
	j->linearMass[1] = -det * c; j->linearMass[3] = det * a;
	j->angularMass = iA + iB;
	if (j->angularMass > 0.0f) j->angularMass = 1.0f / j->angularMass;
}

static void linear_angular_warm_start(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, int warmStarting, float dtRatio)
{
	if (warmStarting)
	{
		j->impulse[0] *= dtRatio; j->impulse[1] *= dtRatio;
		j->impulse[2] *= dtRatio;
		vec2 P = v_make(j->impulse[0], j->impulse[1]);
		*vA = v_sub(*vA, v_scale(j->invMassA, P));
		*wA -= j->invIA * (v_cross(j->rA, P) + j->impulse[2]);
		*vB = v_add(*vB, v_scale(j->invMassB, P));
		*wB += j->invIB * (v_cross(j->rB, P) + j->impulse[2]);
	}
	else
	{
		j->impulse[0] = j->impulse[1] = 0.0f;
		j->impulse[2] = 0.0f;
	}
}

/* the clamped linear rows shared by both (b2FrictionJoint.cpp:145-170, b2MotorJoint.cpp:163-188) */
static void linear_rows(revolute_t* j, vec2 Cdot, vec2* vA, float* wA, vec2* vB, float* wB, float dt)
{
	vec2 impulse = v_neg(v_make(j->linearMass[0] * Cdot.x + j->linearMass[2] * Cdot.y, j->linearMass[1] * Cdot.x + j->linearMass[3] * Cdot.y));
	vec2 oldImpulse = v_make(j->impulse[0], j->impulse[1]);
	vec2 acc = v_add(oldImpulse, impulse);
	float maxImpulse = dt * j->maxForce;
	if (v_dot(acc, acc) > maxImpulse * maxImpulse)
	{
		v_normalize(&acc);
		acc.x *= maxImpulse;
		acc.y *= maxImpulse;
	}
	j->impulse[0] = acc.x;
	j->impulse[1] = acc.y;
	impulse = v_sub(acc, oldImpulse);
	*vA = v_sub(*vA, v_scale(j->invMassA, impulse));
	*wA -= j->invIA * v_cross(j->rA, impulse);
	*vB = v_add(*vB, v_scale(j->invMassB, impulse));
	*wB += j->invIB * v_cross(j->rB, impulse);
}

static void angular_row(revolute_t* j, float Cdot, float* wA, float* wB, float dt)
{
	float impulse = -j->angularMass * Cdot;
	float oldImpulse = j->impulse[2];
	float maxImpulse = dt * j->maxTorque;
	j->impulse[2] = f_clamp(j->impulse[2] + impulse, -maxImpulse, maxImpulse);
	impulse = j->impulse[2] - oldImpulse;
	*wA -= j->invIA * impulse;
	*wB += j->invIB * impulse;
}

/* b2FrictionJoint::InitVelocityConstraints b2FrictionJoint.cpp:58-122 */
void b2o_friction_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	float aA, vec2* vA, float* wA, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	j->rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	j->rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	linear_angular_mass(j);
	linear_angular_warm_start(j, vA, wA, vB, wB, warmStarting, dtRatio);
}

/* b2FrictionJoint::SolveVelocityConstraints :124-171 */
void b2o_friction_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt)
{
	angular_row(j, *wB - *wA, wA, wB, dt);
	vec2 Cdot = v_sub(v_sub(v_add(*vB, v_cross_sv(*wB, j->rB)), *vA), v_cross_sv(*wA, j->rA));
	linear_rows(j, Cdot, vA, wA, vB, wB, dt);
}

/* b2MotorJoint::InitVelocityConstraints b2MotorJoint.cpp:62-134 */
void b2o_motor_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	j->rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	j->rB = r_mul(qB, v_neg(lcB));
	linear_angular_mass(j);
	j->linearError = v_sub(v_sub(v_add(cB, j->rB), cA), j->rA);
	j->angularError = aB - aA - j->referenceAngle;
	linear_angular_warm_start(j, vA, wA, vB, wB, warmStarting, dtRatio);
}

/* b2MotorJoint::SolveVelocityConstraints :136-189 */
void b2o_motor_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt, float inv_dt)
{
	angular_row(j, *wB - *wA + inv_dt * j->correctionFactor * j->angularError, wA, wB, dt);
	vec2 Cdot = v_add(v_sub(v_sub(v_add(*vB, v_cross_sv(*wB, j->rB)), *vA), v_cross_sv(*wA, j->rA)),
		v_scale(inv_dt * j->correctionFactor, j->linearError));
	linear_rows(j, Cdot, vA, wA, vB, wB, dt);
}

/* ---- pulley joint ------------------------------------------------------------------------------- */
static vec2 pulley_dir(vec2 u, float* length)
{
	*length = v_length(u);
	if (*length > 10.0f * B2O_LINEAR_SLOP) return v_scale(1.0f / *length, u);
	return v_make(0.0f, 0.0f);
}

/* InitVelocityConstraints b2PulleyJoint.cpp:81-156 */
void b2o_pulley_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio)
{
	j->localCenterA = lcA; j->localCenterB = lcB;
	j->invMassA = mA; j->invMassB = mB; j->invIA = iA; j->invIB = iB;
	rot qA = r_make(aA), qB = r_make(aB);
	j->rA = r_mul(qA, v_sub(j->localAnchorA, lcA));
	j->rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	float lengthA, lengthB;
	j->uA = pulley_dir(v_sub(v_add(cA, j->rA), j->groundAnchorA), &lengthA);
	j->uB = pulley_dir(v_sub(v_add(cB, j->rB), j->groundAnchorB), &lengthB);
	float ruA = v_cross(j->rA, j->uA);
	float ruB = v_cross(j->rB, j->uB);
	float effA = mA + iA * ruA * ruA;
	float effB = mB + iB * ruB * ruB;
	j->mass = effA + j->ratio * j->ratio * effB;
	if (j->mass > 0.0f) j->mass = 1.0f / j->mass;
	if (warmStarting)
	{
		j->impulse[0] *= dtRatio;
		vec2 PA = v_scale(-(j->impulse[0]), j->uA);
		vec2 PB = v_scale(-j->ratio * j->impulse[0], j->uB);
		*vA = v_add(*vA, v_scale(mA, PA));
		*wA += iA * v_cross(j->rA, PA);
		*vB = v_add(*vB, v_scale(mB, PB));
		*wB += iB * v_cross(j->rB, PB);
	}
	else
	{
		j->impulse[0] = 0.0f;
	}
}

/* SolveVelocityConstraints :158-183 */
void b2o_pulley_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB)
{
	vec2 vpA = v_add(*vA, v_cross_sv(*wA, j->rA));
	vec2 vpB = v_add(*vB, v_cross_sv(*wB, j->rB));
	float Cdot = -v_dot(j->uA, vpA) - j->ratio * v_dot(j->uB, vpB);
	float impulse = -j->mass * Cdot;
	j->impulse[0] += impulse;
	vec2 PA = v_scale(-impulse, j->uA);
	vec2 PB = v_scale(-j->ratio * impulse, j->uB);
	*vA = v_add(*vA, v_scale(j->invMassA, PA));
	*wA += j->invIA * v_cross(j->rA, PA);
	*vB = v_add(*vB, v_scale(j->invMassB, PB));
	*wB += j->invIB * v_cross(j->rB, PB);
}

/* SolvePositionConstraints :185-253 */
int b2o_pulley_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB)
{
	rot qA = r_make(*aA), qB = r_make(*aB);
	vec2 rA = r_mul(qA, v_sub(j->localAnchorA, j->localCenterA));
	vec2 rB = r_mul(qB, v_sub(j->localAnchorB, j->localCenterB));
	float lengthA, lengthB;
	vec2 uA = pulley_dir(v_sub(v_add(*cA, rA), j->groundAnchorA), &lengthA);
	vec2 uB = pulley_dir(v_sub(v_add(*cB, rB), j->groundAnchorB), &lengthB);
	float ruA = v_cross(rA, uA);
	float ruB = v_cross(rB, uB);
	float effA = j->invMassA + j->invIA * ruA * ruA;
	float effB = j->invMassB + j->invIB * ruB * ruB;
	float mass = effA + j->ratio * j->ratio * effB;
	if (mass > 0.0f) mass = 1.0f / mass;
	float C = j->constant - lengthA - j->ratio * lengthB;
	float linearError = f_abs(C);
	float impulse = -mass * C;
	vec2 PA = v_scale(-impulse, uA);
	vec2 PB = v_scale(-j->ratio * impulse, uB);
	*cA = v_add(*cA, v_scale(j->invMassA, PA));
	*aA += j->invIA * v_cross(rA, PA);
	*cB = v_add(*cB, v_scale(j->invMassB, PB));
	*aB += j->invIB * v_cross(rB, PB);
	return linearError < B2O_LINEAR_SLOP;
}

/* ---- mouse joint -------------------------------------------------------------------------------- */
/* InitVelocityConstraints b2MouseJoint.cpp:99-163 */
void b2o_mouse_init(revolute_t* j, float massB, float mB, float iB, vec2 lcB, vec2 cB, float aB, vec2* vB, float* wB,
	int warmStarting, float dtRatio, float dt)
{
	j->localCenterB = lcB;
	j->invMassB = mB; j->invIB = iB;
	rot qB = r_make(aB);
	float omega = 2.0f * B2O_PI * j->frequencyHz;
	float d = 2.0f * massB * j->dampingRatio * omega;
	float k = massB * (omega * omega);
	j->gamma = dt * (d + dt * k);
	if (j->gamma != 0.0f) j->gamma = 1.0f / j->gamma;
	j->beta = dt * k * j->gamma;
	j->rB = r_mul(qB, v_sub(j->localAnchorB, lcB));
	float kxx = mB + iB * j->rB.y * j->rB.y + j->gamma;
	float kxy = -iB * j->rB.x * j->rB.y;
	float kyy = mB + iB * j->rB.x * j->rB.x + j->gamma;
	float a = kxx, b = kxy, c = kxy, dd = kyy;
	float det = a * dd - b * c;
	if (det != 0.0f) det = 1.0f / det;
	j->linearMass[0] = det * dd; j->linearMass[2] = -det * b;
	j->linearMass[1] = -det * c; j->linearMass[3] = det * a;
	j->mouseC = v_sub(v_add(cB, j->rB), j->localAnchorA);
	j->mouseC.x *= j->beta;
	j->mouseC.y *= j->beta;
	*wB *= 0.98f;
	if (warmStarting)
	{
		j->impulse[0] *= dtRatio; j->impulse[1] *= dtRatio;
		vec2 P = v_make(j->impulse[0], j->impulse[1]);
		*vB = v_add(*vB, v_scale(mB, P));
		*wB += iB * v_cross(j->rB, P);
	}
	else
	{
		j->impulse[0] = j->impulse[1] = 0.0f;
	}
}

/* SolveVelocityConstraints :165-192 */
void b2o_mouse_velocity(revolute_t* j, vec2* vB, float* wB, float dt)
{
	vec2 Cdot = v_add(*vB, v_cross_sv(*wB, j->rB));
	vec2 old = v_make(j->impulse[0], j->impulse[1]);
	vec2 rhs = v_neg(v_add(v_add(Cdot, j->mouseC), v_scale(j->gamma, old)));
	vec2 impulse = v_make(j->linearMass[0] * rhs.x + j->linearMass[2] * rhs.y, j->linearMass[1] * rhs.x + j->linearMass[3] * rhs.y);
	vec2 acc = v_add(old, impulse);
	float maxImpulse = dt * j->maxForce;
	if (v_dot(acc, acc) > maxImpulse * maxImpulse)
	{
		float s = maxImpulse / v_length(acc);
		acc.x *= s;
		acc.y *= s;
	}
	j->impulse[0] = acc.x;
	j->impulse[1] = acc.y;
	impulse = v_sub(acc, old);
	*vB = v_add(*vB, v_scale(j->invMassB, impulse));
	*wB += j->invIB * v_cross(j->rB, impulse);
}

/* ---- gear joint --------------------------------------------------------------------------------- */
enum { GA = 0, GB = 1, GC = 2, GD = 3 };

/* Jacobian rows of the two sides at the given angles (b2GearJoint.cpp:159-195, :276-331); returns the effective inverse mass */
static float gear_jacobian(const revolute_t* j, const float a[4], vec2* JvAC, vec2* JvBD, float* JwA, float* JwB, float* JwC, float* JwD)
{
	float mass = 0.0f;
	if (j->typeA == B2O_JOINT_REVOLUTE)
	{
		*JvAC = v_make(0.0f, 0.0f);
		*JwA = 1.0f;
		*JwC = 1.0f;
		mass += j->giA + j->giC;
	}
	else
	{
		rot qA = r_make(a[GA]), qC = r_make(a[GC]);
		vec2 u = r_mul(qC, j->localAxisC);
		vec2 rC = r_mul(qC, v_sub(j->gLocalAnchorC, j->lcC));
		vec2 rA = r_mul(qA, v_sub(j->gLocalAnchorA, j->lcA));
		*JvAC = u;
		*JwC = v_cross(rC, u);
		*JwA = v_cross(rA, u);
		mass += j->gmC + j->gmA + j->giC * *JwC * *JwC + j->giA * *JwA * *JwA;
	}
	if (j->typeB == B2O_JOINT_REVOLUTE)
	{
		*JvBD = v_make(0.0f, 0.0f);
		*JwB = j->ratio;
		*JwD = j->ratio;
		mass += j->ratio * j->ratio * (j->giB + j->giD);
	}
	else
	{
		rot qB = r_make(a[GB]), qD = r_make(a[GD]);
		vec2 u = r_mul(qD, j->localAxisD);
		vec2 rD = r_mul(qD, v_sub(j->gLocalAnchorD, j->lcD));
		vec2 rB = r_mul(qB, v_sub(j->gLocalAnchorB, j->lcB));
		*JvBD = v_scale(j->ratio, u);
		*JwD = j->ratio * v_cross(rD, u);
		*JwB = j->ratio * v_cross(rB, u);
		mass += j->ratio * j->ratio * (j->gmD + j->gmB) + j->giD * *JwD * *JwD + j->giB * *JwB * *JwB;
	}
	return mass;
}

static void gear_apply(const revolute_t* j, gear_bodies* b, float impulse)
{
	b->v[GA] = v_add(b->v[GA], v_scale(j->gmA * impulse, j->JvAC));
	b->w[GA] += j->giA * impulse * j->JwA;
	b->v[GB] = v_add(b->v[GB], v_scale(j->gmB * impulse, j->JvBD));
	b->w[GB] += j->giB * impulse * j->JwB;
	b->v[GC] = v_sub(b->v[GC], v_scale(j->gmC * impulse, j->JvAC));
	b->w[GC] -= j->giC * impulse * j->JwC;
	b->v[GD] = v_sub(b->v[GD], v_scale(j->gmD * impulse, j->JvBD));
	b->w[GD] -= j->giD * impulse * j->JwD;
}

/* InitVelocityConstraints b2GearJoint.cpp:131-222 */
void b2o_gear_init(revolute_t* j, gear_bodies* b, const float invMass[4], const float invI[4], const vec2 lc[4], int warmStarting)
{
	j->lcA = lc[GA]; j->lcB = lc[GB]; j->lcC = lc[GC]; j->lcD = lc[GD];
	j->gmA = invMass[GA]; j->gmB = invMass[GB]; j->gmC = invMass[GC]; j->gmD = invMass[GD];
	j->giA = invI[GA]; j->giB = invI[GB]; j->giC = invI[GC]; j->giD = invI[GD];
	float mass = gear_jacobian(j, b->a, &j->JvAC, &j->JvBD, &j->JwA, &j->JwB, &j->JwC, &j->JwD);
	j->mass = mass > 0.0f ? 1.0f / mass : 0.0f;
	if (warmStarting) gear_apply(j, b, j->impulse[0]);
	else j->impulse[0] = 0.0f;
}

/* SolveVelocityConstraints :224-258 */
void b2o_gear_velocity(revolute_t* j, gear_bodies* b)
{
	float Cdot = v_dot(j->JvAC, v_sub(b->v[GA], b->v[GC])) + v_dot(j->JvBD, v_sub(b->v[GB], b->v[GD]));
	Cdot += (j->JwA * b->w[GA] - j->JwC * b->w[GC]) + (j->JwB * b->w[GB] - j->JwD * b->w[GD]);
	float impulse = -j->mass * Cdot;
	j->impulse[0] += impulse;
	gear_apply(j, b, impulse);
}

/* SolvePositionConstraints :260-362 (its linearError stays zero: always "solved") */
int b2o_gear_position(const revolute_t* j, gear_bodies* b)
{
	vec2 JvAC, JvBD;
	float JwA, JwB, JwC, JwD;
	float mass = gear_jacobian(j, b->a, &JvAC, &JvBD, &JwA, &JwB, &JwC, &JwD);
	float coordinateA, coordinateB;
	if (j->typeA == B2O_JOINT_REVOLUTE)
	{
		coordinateA = b->a[GA] - b->a[GC] - j->referenceAngleA;
	}
	else
	{
		rot qA = r_make(b->a[GA]), qC = r_make(b->a[GC]);
		vec2 rA = r_mul(qA, v_sub(j->gLocalAnchorA, j->lcA));
		vec2 pC = v_sub(j->gLocalAnchorC, j->lcC);
		vec2 pA = r_mul_t(qC, v_add(rA, v_sub(b->c[GA], b->c[GC])));
		coordinateA = v_dot(v_sub(pA, pC), j->localAxisC);
	}
	if (j->typeB == B2O_JOINT_REVOLUTE)
	{
		coordinateB = b->a[GB] - b->a[GD] - j->referenceAngleB;
	}
	else
	{
		rot qB = r_make(b->a[GB]), qD = r_make(b->a[GD]);
		vec2 rB = r_mul(qB, v_sub(j->gLocalAnchorB, j->lcB));
		vec2 pD = v_sub(j->gLocalAnchorD, j->lcD);
		vec2 pB = r_mul_t(qD, v_add(rB, v_sub(b->c[GB], b->c[GD])));
		coordinateB = v_dot(v_sub(pB, pD), j->localAxisD);
	}
	float C = (coordinateA + j->ratio * coordinateB) - j->constant;
	float impulse = 0.0f;
	if (mass > 0.0f) impulse = -C / mass;
	b->c[GA] = v_add(b->c[GA], v_scale(j->gmA * impulse, JvAC));
	b->a[GA] += j->giA * impulse * JwA;
	b->c[GB] = v_add(b->c[GB], v_scale(j->gmB * impulse, JvBD));
	b->a[GB] += j->giB * impulse * JwB;
	b->c[GC] = v_sub(b->c[GC], v_scale(j->gmC * impulse, JvAC));
	b->a[GC] -= j->giC * impulse * JwC;
	b->c[GD] = v_sub(b->c[GD], v_scale(j->gmD * impulse, JvBD));
	b->a[GD] -= j->giD * impulse * JwD;
	return 1;
}

/* ---- what a joint did in the last step: b2Joint::GetReactionForce / GetReactionTorque and the motor's share
 * (b2RevoluteJoint.cpp:439-456, b2DistanceJoint.cpp:236-247, b2PrismaticJoint.cpp:502-510,618-621, b2WeldJoint.cpp:316-325,
 * b2WheelJoint.cpp:340-348,434-437, b2RopeJoint.cpp:207-217, b2FrictionJoint.cpp:222-230, b2MotorJoint.cpp:236-244,
 * b2PulleyJoint.cpp:273-283, b2MouseJoint.cpp:210-218, b2GearJoint.cpp:381-391). out4 = force.x, force.y, torque, motor ---- */
void b2o_joint_reaction(const revolute_t* j, float inv_dt, float out4[4])
{
	vec2 F = v_make(0.0f, 0.0f);
	float T = 0.0f, M = 0.0f;
	switch (j->type)
	{
	case B2O_JOINT_REVOLUTE:
	case B2O_JOINT_WELD:
		F = v_scale(inv_dt, v_make(j->impulse[0], j->impulse[1]));
		T = inv_dt * j->impulse[2];
		if (j->type == B2O_JOINT_REVOLUTE) M = inv_dt * j->motorImpulse;
		break;
	case B2O_JOINT_DISTANCE:
	case B2O_JOINT_ROPE:
		F = v_scale(inv_dt * j->impulse[0], j->u);
		break;
	case B2O_JOINT_PRISMATIC:
		F = v_scale(inv_dt, v_add(v_scale(j->impulse[0], j->perp), v_scale(j->motorImpulse + j->impulse[2], j->axis)));
		T = inv_dt * j->impulse[1];
		M = inv_dt * j->motorImpulse;
		break;
	case B2O_JOINT_WHEEL:
		/* m_ax in axis, m_ay in perp */
		F = v_scale(inv_dt, v_add(v_scale(j->impulse[0], j->perp), v_scale(j->springImpulse, j->axis)));
		T = inv_dt * j->motorImpulse;
		M = inv_dt * j->motorImpulse;
		break;
	case B2O_JOINT_FRICTION:
	case B2O_JOINT_MOTOR:
		F = v_scale(inv_dt, v_make(j->impulse[0], j->impulse[1]));
		T = inv_dt * j->impulse[2];
		break;
	case B2O_JOINT_PULLEY:
		F = v_scale(inv_dt, v_scale(j->impulse[0], j->uB));
		break;
	case B2O_JOINT_MOUSE:
		F = v_scale(inv_dt, v_make(j->impulse[0], j->impulse[1]));
		T = inv_dt * 0.0f;
		break;
	case B2O_JOINT_GEAR:
		F = v_scale(inv_dt, v_scale(j->impulse[0], j->JvAC));
		T = inv_dt * (j->impulse[0] * j->JwA);
		break;
	default:
		break;
	}
	out4[0] = F.x; out4[1] = F.y; out4[2] = T; out4[3] = M;
}
