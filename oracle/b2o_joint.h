/* b2o_joint.h - CPU oracle, joint state: revolute, distance, prismatic, weld, wheel, rope, friction, motor, pulley, mouse, gear (TEST INFRASTRUCTURE, see b2o.h). */
#ifndef B2O_JOINT_H
#define B2O_JOINT_H

#include "b2o_internal.h"

enum { B2O_JOINT_REVOLUTE = 0, B2O_JOINT_DISTANCE = 1, B2O_JOINT_PRISMATIC = 2, B2O_JOINT_WELD = 3,
	B2O_JOINT_WHEEL = 4, B2O_JOINT_ROPE = 5, B2O_JOINT_FRICTION = 6, B2O_JOINT_MOTOR = 7, B2O_JOINT_PULLEY = 8, B2O_JOINT_MOUSE = 9, B2O_JOINT_GEAR = 10 };

typedef struct
{
	int type;
	int bodyA, bodyB;
	vec2 localAnchorA, localAnchorB;
	float referenceAngle;
	int enableLimit;
	float lowerAngle, upperAngle;
	int enableMotor;
	float motorSpeed, maxMotorTorque;
	int collideConnected;
	float impulse[3], motorImpulse;
	int limitState; /* 0 inactive, 1 at lower, 2 at upper, 3 equal */
	vec2 rA, rB, localCenterA, localCenterB;
	float invMassA, invMassB, invIA, invIB;
	float ex[3], ey[3], ez[3], motorMass;
	/* distance joint (b2DistanceJoint.h:137-163); its accumulated impulse is impulse[0] */
	float length, frequencyHz, dampingRatio;
	float gamma, bias, mass;
	vec2 u;
	/* prismatic joint (b2PrismaticJoint.h:170-196): limits in lowerAngle / upperAngle, motor force in maxMotorTorque,
	 * m_K in ex / ey / ez ; weld joint (b2WeldJoint.h:97-123): m_mass in ex / ey / ez, gamma / bias above */
	vec2 localXAxisA, localYAxisA, axis, perp;
	float s1, s2, a1, a2;
	/* wheel joint (b2WheelJoint.h:170-211): m_ax / m_ay in axis / perp, m_sAx, m_sBx, m_sAy, m_sBy in s1, s2, a1, a2,
	 * m_impulse = impulse[0], m_mass = mass, spring state below ; rope joint (b2RopeJoint.h:97-117): m_maxLength = length,
	 * m_state = limitState ; friction / motor joints (b2FrictionJoint.h:93-117, b2MotorJoint.h:107-133): linear impulse
	 * impulse[0..1], angular impulse impulse[2], linearOffset = localAnchorA, angularOffset = referenceAngle */
	float springImpulse, springMass;
	float curLength;
	float maxForce, maxTorque, correctionFactor;
	float linearMass[4]; /* ex.x, ex.y, ey.x, ey.y */
	float angularMass;
	vec2 linearError;
	float angularError;
	/* pulley joint (b2PulleyJoint.h:118-150): impulse[0], mass */
	vec2 groundAnchorA, groundAnchorB, uA, uB;
	float ratio, constant;
	/* mouse joint (b2MouseJoint.h:101-126): m_targetA = localAnchorA, impulse[0..1], m_mass in linearMass, m_C, m_beta; gamma above */
	vec2 mouseC;
	float beta;
	/* gear joint (b2GearJoint.h:84-122): impulse[0], mass, ratio, constant above */
	int bodyC, bodyD, typeA, typeB;
	vec2 gLocalAnchorA, gLocalAnchorB, gLocalAnchorC, gLocalAnchorD, localAxisC, localAxisD;
	float referenceAngleA, referenceAngleB;
	vec2 lcA, lcB, lcC, lcD;
	float gmA, gmB, gmC, gmD, giA, giB, giC, giD;
	vec2 JvAC, JvBD;
	float JwA, JwB, JwC, JwD;
	int islandFlag;
	int nextA, nextB; /* per-body joint lists, newest first: edge id = joint * 2 + side */
} revolute_t;

void b2o_revolute_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	float aA, vec2* vA, float* wA, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio);
void b2o_revolute_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt);
int b2o_revolute_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB);

void b2o_distance_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio, float dt);
void b2o_distance_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB);
int b2o_distance_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB);

void b2o_prismatic_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio);
void b2o_prismatic_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt);
int b2o_prismatic_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB);

void b2o_weld_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	float aA, vec2* vA, float* wA, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio, float dt);
void b2o_weld_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB);
int b2o_weld_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB);

void b2o_wheel_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio, float dt);
void b2o_wheel_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt);
int b2o_wheel_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB);

void b2o_rope_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio);
void b2o_rope_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float inv_dt);
int b2o_rope_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB);

/* friction and motor joints have no position step (b2FrictionJoint.cpp:173-178, b2MotorJoint.cpp:191-196) */
void b2o_friction_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	float aA, vec2* vA, float* wA, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio);
void b2o_friction_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt);
void b2o_motor_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio);
void b2o_motor_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB, float dt, float inv_dt);

void b2o_pulley_init(revolute_t* j, float mA, float iA, vec2 lcA, float mB, float iB, vec2 lcB,
	vec2 cA, float aA, vec2* vA, float* wA, vec2 cB, float aB, vec2* vB, float* wB, int warmStarting, float dtRatio);
void b2o_pulley_velocity(revolute_t* j, vec2* vA, float* wA, vec2* vB, float* wB);
int b2o_pulley_position(const revolute_t* j, vec2* cA, float* aA, vec2* cB, float* aB);

/* four bodies: positions / velocities as arrays in the order A, B, C, D */
typedef struct { vec2 c[4]; float a[4]; vec2 v[4]; float w[4]; } gear_bodies;
void b2o_gear_init(revolute_t* j, gear_bodies* b, const float invMass[4], const float invI[4], const vec2 lc[4], int warmStarting);
void b2o_gear_velocity(revolute_t* j, gear_bodies* b);
int b2o_gear_position(const revolute_t* j, gear_bodies* b);

/* acts on bodyB only; no position step (b2MouseJoint.cpp:194-198) */
void b2o_mouse_init(revolute_t* j, float massB, float mB, float iB, vec2 lcB, vec2 cB, float aB, vec2* vB, float* wB,
	int warmStarting, float dtRatio, float dt);
void b2o_mouse_velocity(revolute_t* j, vec2* vB, float* wB, float dt);

void b2o_joint_reaction(const revolute_t* j, float inv_dt, float out4[4]);

#endif
