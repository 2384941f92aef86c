/* b2o_step.c - CPU oracle: serial, plain-C restatement of b2World::Step() (TEST INFRASTRUCTURE, see b2o.h).
 *
 * Follows the reference's data flow and ORDER literally: per-body contact lists with new contacts at
 * the front (b2ContactManager.cpp:531-553), DFS islands seeded in creation order
 * (b2World.cpp:1207-1371), sequential-impulse sweeps in island order (b2ContactSolver.cpp), fat-AABB
 * broad-phase semantics (b2DynamicTree.cpp:130-174) with a brute-force overlap query in place of the
 * tree (the pair set does not depend on the index structure), creation sorted by proxy ids
 * (b2ContactManager.cpp:366-386). Joints: revolute, distance, prismatic, weld, wheel, rope, friction, motor, pulley, mouse, gear (b2o_joint.c). Not covered (same as the device
 * path): chain shapes. Continuous collision: b2o_toi.c
 * (GJK + time of impact) and the TOI event loop at the end of this file.
 */
#include "b2o_internal.h"
#include "b2o_joint.h"
#include "b2o_toi.h"

#include <stdlib.h>
#include <string.h>

#define BF_AWAKE 0x4u
#define BF_AUTOSLEEP 0x8u
#define BF_BULLET 0x10u
#define BF_FIXEDROT 0x20u
#define BF_ACTIVE 0x40u
#define BF_ISLAND 0x80u

#define CF_TOUCHING 0x1u
#define CF_ENABLED 0x2u
#define CF_FILTER 0x4u
#define CF_ISLAND 0x8u
#define CF_TOI 0x10u           /* e_toiFlag: m_toi is valid */
#define CF_TOI_CANDIDATE 0x20u /* e_toiCandidateFlag */

#define B2O_MAX_SUB_STEPS 8
#define B2O_TOI_CONTACT_CAP 32
#define B2O_TOI_BODY_CAP 64
#define B2O_TOI_BAUMGARTE 0.75f

typedef struct
{
	int type;
	uint32_t flags;
	xform xf;
	vec2 localCenter, c0, c;
	float a0, a, alpha0;
	vec2 v;
	float w;
	vec2 force;
	float torque;
	float mass, invMass, I, invI;
	float linearDamping, angularDamping, gravityScale, sleepTime;
	int fixtureHead;   /* newest fixture first */
	int jointHead;     /* joint edge id = joint * 2 + side, newest first */
	int contactHead;   /* edge id = contact * 2 + side, newest first */
	int islandIndex;
	int label;
	int worldIndex;    /* slot in b2o_world::nonStatic (m_nonStaticBodies), -1 for static bodies */
	int dead;          /* destroyed: the id stays */
} body_t;

typedef struct
{
	int body;
	b2o_shape shape;
	float density, friction, restitution;
	uint16_t categoryBits, maskBits;
	int16_t groupIndex;
	int isSensor, thick;
	int proxyId;
	float fat[4];
	int nextInBody;
	int dead;          /* destroyed: the id stays, the proxy is gone */
} fixture_t;

typedef struct
{
	int alive;
	int fixtureA, fixtureB;
	int bodyA, bodyB;
	uint32_t flags;
	int proxyLo, proxyHi;
	manifold m;
	float friction, restitution, tangentSpeed;
	int prev[2], next[2]; /* edge links: side 0 lives in bodyA's list, side 1 in bodyB's list */
	uint64_t seq;
	float toi;
	int toiCount;
	int managerIndex; /* position in b2o_world::carray (b2Contact::m_managerIndex) */
	int reported;     /* contact events: the host has been told that this contact touches */
	manifold oldm;    /* the manifold before this step's Update (what b2ContactListener::PreSolve is handed) */
	int preSolveDue;  /* updated by this step's Collide, touching, not a sensor (b2Contact.cpp:283) */
} contact_t;

/* b2VelocityConstraintPoint / b2ContactVelocityConstraint / b2ContactPositionConstraint
 * (b2ContactSolver.h:31-57, b2ContactSolver.cpp:32-45) */
typedef struct
{
	vec2 rA[2], rB[2];
	float normalImpulse[2], tangentImpulse[2], normalMass[2], tangentMass[2], velocityBias[2];
	vec2 normal;
	float nm[4], K[4]; /* ex.x, ex.y, ey.x, ey.y */
	int indexA, indexB;
	float invMassA, invMassB, invIA, invIB;
	float friction, restitution, tangentSpeed;
	int pointCount, contact;
	/* position part */
	vec2 localPoints[2], localNormal, localPoint, localCenterA, localCenterB;
	float radiusA, radiusB;
	int type, pcPointCount;
} constraint_t;

struct b2o_world
{
	vec2 gravity;
	int allowSleep, warmStarting, continuous;
	float inv_dt0;
	int newFixture;
	body_t* bodies; int nBodies, capBodies;
	fixture_t* fixtures; int nFixtures, capFixtures;
	contact_t* contacts; int nContactSlots, capContacts, liveContacts;
	int* freeContacts; int nFreeContacts, capFreeContacts;
	int* moves; int nMoves, capMoves;
	uint64_t nextSeq;
	/* proxy id allocator (b2DynamicTree free list) */
	int nextNode, leafCount;
	int* freeLeaves; int nFreeLeaves, capFreeLeaves;
	revolute_t* joints; int nJoints, capJoints;
	/* b2ContactManager::m_contacts: TOI candidates first ([0, toiCount)), maintained by swaps */
	int* carray; int nArr, capArr, toiCount;
	int toiEvents, toiCalls; /* TOI sub-steps solved / b2TimeOfImpact calls so far (diagnostics) */
	/* contact events of the last step (b2ContactListener::BeginContact / EndContact, one net event per contact and step) */
	int eventsOn;
	b2o_contact_event* events; int nEvents, capEvents;
	uint64_t* eventKeys; int capEventKeys;
	/* island sharding (same ownership rule and exchange format as box2d-mt_amd/csrc/b2d_kernels_shard.h) */
	int shardRank, shardCount;
	int shardBodies[8], shardContacts[8], shardJoints[8]; /* what every rank's slab of the exchange holds this step (every rank counts all) */
	unsigned char* bodyOwned; int capBodyOwned;       /* per body: its island was solved HERE this step */
	unsigned char* contactOwned; int capContactOwned; /* per contact slot */
	unsigned char* jointOwned; int capJointOwned;
	float stepDt; int stepVelIters, stepPosIters;     /* parameters of the running step (phase entry points) */
	/* spatial ownership (include/b2hip.h: b2hip_shard_spatial; the protocol of box2d-mt_amd/csrc/b2d_kernels_spatial.h restated
	 * serially: every rank keeps the whole structure - bodies, proxies, contact slots - and evaluates, solves and moves the
	 * bodies it owns; rows / fat AABBs of moved bodies and new pairs are all-gathered, components that a new contact joins
	 * over an ownership boundary migrate) */
	int spatial;
	unsigned char* owner; int capOwner;
	int (*gatherFn)(void* user, const void* send, size_t bytes, void* recv); void* gatherUser;
	float spBounds[9];
	long long spMigrated, spResolves, spBytes;
	int spOwnersDirty, spFailed;
	/* m_nonStaticBodies (b2World.cpp:573, 662-667): island seeds are taken in this order */
	int* nonStatic; int nNonStatic, capNonStatic;
	/* the other listener callbacks and the user contact filter (same protocol as include/b2hip.h) */
	b2o_should_collide_fn filterFn; void* filterUser;
	b2o_pre_solve_fn preSolveFn; void* preSolveUser;
	b2o_pre_solve_batch_fn preSolveBatchFn;
	int subStepping, stepComplete; /* b2World::m_subStepping, m_stepComplete (b2World.cpp:462-464) */
	b2o_toi_callback* toiLog; int nToiLog, capToiLog; /* listener calls of this step's TOI sub-steps (contact_index = slot until read) */
	int postSolveOn;
	b2o_contact_impulse* postSolve; int nPostSolve, capPostSolve;
	int* postSolveSlot; int capPostSolveSlot; /* contact slot of each record until the end of the step */
};

#define GROW(ptr, cap, need, type)                                            \
	do                                                                        \
	{                                                                         \
		if ((need) > (cap))                                                   \
		{                                                                     \
			int ncap = (cap) ? (cap) * 2 : 64;                                \
			while (ncap < (need)) ncap *= 2;                                  \
			(ptr) = (type*)realloc((ptr), (size_t)ncap * sizeof(type));       \
			(cap) = ncap;                                                     \
		}                                                                     \
	} while (0)

b2o_world* b2o_world_create(float gx, float gy, int allow_sleep, int warm_starting, int continuous)
{
	b2o_world* w = (b2o_world*)calloc(1, sizeof(b2o_world));
	w->gravity = v_make(gx, gy);
	w->allowSleep = allow_sleep;
	w->warmStarting = warm_starting;
	w->continuous = continuous;
	w->stepComplete = 1;
	return w;
}

/* b2World::SetSubStepping (b2World.h:183) */
void b2o_set_sub_stepping(b2o_world* w, int flag) { w->subStepping = flag != 0; }
int b2o_step_complete(const b2o_world* w) { return w->stepComplete; }

void b2o_world_destroy(b2o_world* w)
{
	if (!w) return;
	free(w->bodies);
	free(w->fixtures);
	free(w->contacts);
	free(w->freeContacts);
	free(w->moves);
	free(w->freeLeaves);
	free(w->joints);
	free(w->carray);
	free(w->nonStatic);
	free(w->events);
	free(w->eventKeys);
	free(w->postSolve);
	free(w->postSolveSlot);
	free(w->toiLog);
	free(w->bodyOwned);
	free(w->contactOwned);
	free(w->jointOwned);
	free(w);
}

void b2o_set_gravity(b2o_world* w, float gx, float gy) { w->gravity = v_make(gx, gy); }

void b2o_set_flags(b2o_world* w, int allow_sleep, int warm_starting, int continuous)
{
	w->allowSleep = allow_sleep;
	w->warmStarting = warm_starting;
	w->continuous = continuous;
}

/* b2Body::b2Body  Box2D/Dynamics/b2Body.cpp:26-112 */
int b2o_create_body(b2o_world* w, const b2o_body_def* d)
{
	GROW(w->bodies, w->capBodies, w->nBodies + 1, body_t);
	body_t* b = &w->bodies[w->nBodies];
	memset(b, 0, sizeof(*b));
	b->type = d->type;
	if (d->bullet) b->flags |= BF_BULLET;
	if (d->fixed_rotation) b->flags |= BF_FIXEDROT;
	if (d->allow_sleep) b->flags |= BF_AUTOSLEEP;
	if (d->awake) b->flags |= BF_AWAKE;
	if (d->active) b->flags |= BF_ACTIVE;
	b->xf.p = v_make(d->px, d->py);
	b->xf.q = r_make(d->angle);
	b->c0 = b->c = b->xf.p;
	b->a0 = b->a = d->angle;
	b->v = v_make(d->vx, d->vy);
	b->w = d->w;
	b->linearDamping = d->linear_damping;
	b->angularDamping = d->angular_damping;
	b->gravityScale = d->gravity_scale;
	if (d->type == 2) { b->mass = 1.0f; b->invMass = 1.0f; }
	b->fixtureHead = -1;
	b->jointHead = -1;
	b->contactHead = -1;
	b->label = -1;
	b->worldIndex = -1;
	if (d->type != 0)
	{
		/* b2World::CreateBody (b2World.cpp:571-575) */
		GROW(w->nonStatic, w->capNonStatic, w->nNonStatic + 1, int);
		b->worldIndex = w->nNonStatic;
		w->nonStatic[w->nNonStatic++] = w->nBodies;
	}
	return w->nBodies++;
}

/* b2PolygonShape::ComputeMass b2PolygonShape.cpp:359-440; b2CircleShape::ComputeMass b2CircleShape.cpp:92-100;
 * b2EdgeShape::ComputeMass b2EdgeShape.cpp:131-138 */
static void shape_mass(const b2o_shape* s, float density, float* mass, vec2* center, float* I)
{
	if (s->type == SHAPE_CIRCLE)
	{
		vec2 p = shape_vert(s, 0);
		*mass = density * B2O_PI * s->radius * s->radius;
		*center = p;
		*I = (*mass) * (0.5f * s->radius * s->radius + v_dot(p, p));
		return;
	}
	if (SHAPE_IS_SEGMENT(s->type))
	{
		*mass = 0.0f;
		*center = v_scale(0.5f, v_add(shape_vert(s, 0), shape_vert(s, 1)));
		*I = 0.0f;
		return;
	}
	vec2 c = v_make(0.0f, 0.0f);
	float area = 0.0f, inertia = 0.0f;
	vec2 ref = v_make(0.0f, 0.0f);
	for (int i = 0; i < s->count; ++i) ref = v_add(ref, shape_vert(s, i));
	ref = v_scale(1.0f / s->count, ref);
	const float k_inv3 = 1.0f / 3.0f;
	for (int i = 0; i < s->count; ++i)
	{
		vec2 e1 = v_sub(shape_vert(s, i), ref);
		vec2 e2 = i + 1 < s->count ? v_sub(shape_vert(s, i + 1), ref) : v_sub(shape_vert(s, 0), ref);
		float D = v_cross(e1, e2);
		float tri = 0.5f * D;
		area += tri;
		c = v_add(c, v_scale(tri * k_inv3, v_add(e1, e2)));
		float intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x;
		float inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y;
		inertia += (0.25f * k_inv3 * D) * (intx2 + inty2);
	}
	*mass = density * area;
	c = v_scale(1.0f / area, c);
	*center = v_add(c, ref);
	*I = density * inertia;
	*I += (*mass) * (v_dot(*center, *center) - v_dot(c, c));
}

/* shape AABBs: b2PolygonShape.cpp:340-357, b2CircleShape.cpp:83-90, b2EdgeShape.cpp:116-129 */
static void shape_aabb(const b2o_shape* s, xform xf, float out[4])
{
	vec2 lower, upper;
	if (s->type == SHAPE_CIRCLE)
	{
		vec2 q = r_mul(xf.q, shape_vert(s, 0));
		vec2 p = v_make(xf.p.x + q.x, xf.p.y + q.y);
		out[0] = p.x - s->radius; out[1] = p.y - s->radius;
		out[2] = p.x + s->radius; out[3] = p.y + s->radius;
		return;
	}
	if (s->type == SHAPE_CHAIN)
	{
		/* b2ChainShape::ComputeAABB  b2ChainShape.cpp:174-189 : no radius */
		vec2 a = xf_mul(xf, shape_vert(s, 0)), b = xf_mul(xf, shape_vert(s, 1));
		lower = v_min(a, b);
		upper = v_max(a, b);
		out[0] = lower.x; out[1] = lower.y;
		out[2] = upper.x; out[3] = upper.y;
		return;
	}
	if (s->type == SHAPE_EDGE)
	{
		vec2 a = xf_mul(xf, shape_vert(s, 0)), b = xf_mul(xf, shape_vert(s, 1));
		lower = v_min(a, b);
		upper = v_max(a, b);
	}
	else
	{
		lower = xf_mul(xf, shape_vert(s, 0));
		upper = lower;
		for (int i = 1; i < s->count; ++i)
		{
			vec2 v = xf_mul(xf, shape_vert(s, i));
			lower = v_min(lower, v);
			upper = v_max(upper, v);
		}
	}
	out[0] = lower.x - s->radius; out[1] = lower.y - s->radius;
	out[2] = upper.x + s->radius; out[3] = upper.y + s->radius;
}

/* b2Body::ResetMassData  b2Body.cpp:310-385 */
static void reset_mass(b2o_world* w, body_t* b)
{
	b->mass = b->invMass = b->I = b->invI = 0.0f;
	b->localCenter = v_make(0.0f, 0.0f);
	if (b->type != 2)
	{
		b->c0 = b->c = b->xf.p;
		b->a0 = b->a;
		return;
	}
	vec2 lc = v_make(0.0f, 0.0f);
	for (int f = b->fixtureHead; f >= 0; f = w->fixtures[f].nextInBody)
	{
		fixture_t* fx = &w->fixtures[f];
		if (fx->density == 0.0f) continue;
		float mass, I;
		vec2 center;
		shape_mass(&fx->shape, fx->density, &mass, &center, &I);
		b->mass += mass;
		lc = v_add(lc, v_scale(mass, center));
		b->I += I;
	}
	if (b->mass > 0.0f)
	{
		b->invMass = 1.0f / b->mass;
		lc = v_scale(b->invMass, lc);
	}
	else
	{
		b->mass = 1.0f;
		b->invMass = 1.0f;
	}
	if (b->I > 0.0f && (b->flags & BF_FIXEDROT) == 0)
	{
		b->I -= b->mass * v_dot(lc, lc);
		b->invI = 1.0f / b->I;
	}
	else
	{
		b->I = 0.0f;
		b->invI = 0.0f;
	}
	vec2 oldCenter = b->c;
	b->localCenter = lc;
	b->c0 = b->c = xf_mul(b->xf, lc);
	b->v = v_add(b->v, v_cross_sv(b->w, v_sub(b->c, oldCenter)));
}

/* b2DynamicTree::AllocateNode / FreeNode (b2DynamicTree.cpp:53-99): leaf ids depend only on the
 * create / destroy sequence; an internal parent node is consumed whenever the tree is not empty. */
static int alloc_proxy_id(b2o_world* w)
{
	int id;
	if (w->nFreeLeaves > 0)
	{
		/* the free list is LIFO; -1 marks the internal node RemoveLeaf gave back under a freed leaf */
		id = w->freeLeaves[--w->nFreeLeaves];
		if (w->leafCount > 0)
		{
			if (w->nFreeLeaves > 0) --w->nFreeLeaves; /* InsertLeaf's parent node */
			else w->nextNode++;
		}
	}
	else
	{
		id = w->nextNode++;
		if (w->leafCount > 0) w->nextNode++;
	}
	w->leafCount++;
	return id;
}

/* b2DynamicTree::DestroyProxy (b2DynamicTree.cpp:121-128): the parent RemoveLeaf drops, then the leaf */
static void free_proxy_id(b2o_world* w, int id)
{
	GROW(w->freeLeaves, w->capFreeLeaves, w->nFreeLeaves + 2, int);
	if (w->leafCount > 1) w->freeLeaves[w->nFreeLeaves++] = -1;
	w->freeLeaves[w->nFreeLeaves++] = id;
	w->leafCount--;
}

/* b2Body::CreateFixture b2Body.cpp:182-226, b2Fixture::Create/CreateProxies b2Fixture.cpp:42-141,
 * b2DynamicTree::CreateProxy b2DynamicTree.cpp:105-119, b2BroadPhase::CreateProxy b2BroadPhase.cpp:46-52 */
int b2o_create_fixture(b2o_world* w, int body, const b2o_fixture_def* d, const b2o_shape* shape)
{
	GROW(w->fixtures, w->capFixtures, w->nFixtures + 1, fixture_t);
	fixture_t* f = &w->fixtures[w->nFixtures];
	memset(f, 0, sizeof(*f));
	body_t* b = &w->bodies[body];
	f->body = body;
	f->shape = *shape;
	f->density = d->density;
	f->friction = d->friction;
	f->restitution = d->restitution;
	f->categoryBits = d->category_bits;
	f->maskBits = d->mask_bits;
	f->groupIndex = d->group_index;
	f->isSensor = d->is_sensor;
	f->thick = d->thick_shape;
	float aabb[4];
	shape_aabb(shape, b->xf, aabb);
	f->fat[0] = aabb[0] - B2O_AABB_EXTENSION;
	f->fat[1] = aabb[1] - B2O_AABB_EXTENSION;
	f->fat[2] = aabb[2] + B2O_AABB_EXTENSION;
	f->fat[3] = aabb[3] + B2O_AABB_EXTENSION;
	f->nextInBody = b->fixtureHead;
	int id = w->nFixtures++;
	b->fixtureHead = id;
	if (b->flags & BF_ACTIVE)
	{
		/* (b2Body.cpp:199-203: an inactive body's fixtures get their proxies when it is activated) */
		f->proxyId = alloc_proxy_id(w);
		GROW(w->moves, w->capMoves, w->nMoves + 1, int);
		w->moves[w->nMoves++] = id;
	}
	else
	{
		f->proxyId = -1;
		f->fat[0] = f->fat[1] = 1e30f;
		f->fat[2] = f->fat[3] = -1e30f;
	}
	if (f->density > 0.0f) reset_mass(w, b);
	w->newFixture = 1;
	return id;
}

/* b2World::CreateJoint (b2World.cpp:679-735) + b2RevoluteJoint::b2RevoluteJoint (b2RevoluteJoint.cpp:46-63) */
int b2o_create_revolute_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, float referenceAngle,
	int enableLimit, float lower, float upper, int enableMotor, float motorSpeed, float maxMotorTorque, int collideConnected)
{
	GROW(w->joints, w->capJoints, w->nJoints + 1, revolute_t);
	revolute_t* j = &w->joints[w->nJoints];
	memset(j, 0, sizeof(*j));
	j->bodyA = bodyA;
	j->bodyB = bodyB;
	j->localAnchorA = v_make(anchors4[0], anchors4[1]);
	j->localAnchorB = v_make(anchors4[2], anchors4[3]);
	j->referenceAngle = referenceAngle;
	j->enableLimit = enableLimit;
	j->lowerAngle = lower;
	j->upperAngle = upper;
	j->enableMotor = enableMotor;
	j->motorSpeed = motorSpeed;
	j->maxMotorTorque = maxMotorTorque;
	j->collideConnected = collideConnected;
	int id = w->nJoints++;
	j->nextA = w->bodies[bodyA].jointHead;
	w->bodies[bodyA].jointHead = id * 2;
	j->nextB = w->bodies[bodyB].jointHead;
	w->bodies[bodyB].jointHead = id * 2 + 1;
	if (!collideConnected)
	{
		for (int e = w->bodies[bodyB].contactHead; e >= 0; e = w->contacts[e >> 1].next[e & 1])
		{
			contact_t* c = &w->contacts[e >> 1];
			int other = (e & 1) == 0 ? c->bodyB : c->bodyA;
			if (other == bodyA) c->flags |= CF_FILTER;
		}
	}
	return id;
}

/* b2World::CreateJoint + b2DistanceJoint::b2DistanceJoint (b2DistanceJoint.cpp:51-63): same list linking and
 * re-filtering as above, distance-specific definition members. */
int b2o_create_distance_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, float length,
	float frequencyHz, float dampingRatio, int collideConnected)
{
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors4, 0.0f, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_DISTANCE;
	j->length = length;
	j->frequencyHz = frequencyHz;
	j->dampingRatio = dampingRatio;
	return id;
}

/* b2PrismaticJoint::b2PrismaticJoint (b2PrismaticJoint.cpp:98-128): the axis is normalised, the y axis is cross(1, x) */
int b2o_create_prismatic_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, const float* axis2, float referenceAngle,
	int enableLimit, float lower, float upper, int enableMotor, float motorSpeed, float maxMotorForce, int collideConnected)
{
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors4, referenceAngle, enableLimit, lower, upper, enableMotor,
		motorSpeed, maxMotorForce, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_PRISMATIC;
	j->localXAxisA = v_make(axis2[0], axis2[1]);
	v_normalize(&j->localXAxisA);
	j->localYAxisA = v_cross_sv(1.0f, j->localXAxisA);
	return id;
}

/* b2WeldJoint::b2WeldJoint (b2WeldJoint.cpp:46-56) */
int b2o_create_weld_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, float referenceAngle,
	float frequencyHz, float dampingRatio, int collideConnected)
{
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors4, referenceAngle, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_WELD;
	j->frequencyHz = frequencyHz;
	j->dampingRatio = dampingRatio;
	return id;
}

/* b2WheelJoint::b2WheelJoint (b2WheelJoint.cpp:47-77): the axis is taken as given (not normalised) */
int b2o_create_wheel_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, const float* axis2, float frequencyHz,
	float dampingRatio, int enableMotor, float motorSpeed, float maxMotorTorque, int collideConnected)
{
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors4, 0.0f, 0, 0.0f, 0.0f, enableMotor, motorSpeed, maxMotorTorque, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_WHEEL;
	j->localXAxisA = v_make(axis2[0], axis2[1]);
	j->localYAxisA = v_cross_sv(1.0f, j->localXAxisA);
	j->frequencyHz = frequencyHz;
	j->dampingRatio = dampingRatio;
	return id;
}

/* b2RopeJoint::b2RopeJoint (b2RopeJoint.cpp:34-46) */
int b2o_create_rope_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, float maxLength, int collideConnected)
{
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors4, 0.0f, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_ROPE;
	j->length = maxLength;
	return id;
}

/* b2FrictionJoint::b2FrictionJoint (b2FrictionJoint.cpp:45-56) */
int b2o_create_friction_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, float maxForce, float maxTorque, int collideConnected)
{
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors4, 0.0f, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_FRICTION;
	j->maxForce = maxForce;
	j->maxTorque = maxTorque;
	return id;
}

/* b2MotorJoint::b2MotorJoint (b2MotorJoint.cpp:48-60) */
int b2o_create_motor_joint(b2o_world* w, int bodyA, int bodyB, const float* linearOffset2, float angularOffset, float maxForce,
	float maxTorque, float correctionFactor, int collideConnected)
{
	float anchors[4] = { linearOffset2[0], linearOffset2[1], 0.0f, 0.0f };
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors, angularOffset, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_MOTOR;
	j->maxForce = maxForce;
	j->maxTorque = maxTorque;
	j->correctionFactor = correctionFactor;
	return id;
}

/* b2PulleyJoint::b2PulleyJoint (b2PulleyJoint.cpp:62-79) */
int b2o_create_pulley_joint(b2o_world* w, int bodyA, int bodyB, const float* anchors4, const float* groundAnchors4, float lengthA,
	float lengthB, float ratio, int collideConnected)
{
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors4, 0.0f, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_PULLEY;
	j->groundAnchorA = v_make(groundAnchors4[0], groundAnchors4[1]);
	j->groundAnchorB = v_make(groundAnchors4[2], groundAnchors4[3]);
	j->ratio = ratio;
	j->constant = lengthA + ratio * lengthB;
	return id;
}

/* b2MouseJoint::b2MouseJoint (b2MouseJoint.cpp:36-55): the anchor on bodyB is where the target lies at creation */
int b2o_create_mouse_joint(b2o_world* w, int bodyA, int bodyB, float tx, float ty, float maxForce, float frequencyHz,
	float dampingRatio, int collideConnected)
{
	vec2 target = v_make(tx, ty);
	vec2 local = xf_mul_tv(w->bodies[bodyB].xf, target);
	float anchors[4] = { tx, ty, local.x, local.y };
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors, 0.0f, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	j->type = B2O_JOINT_MOUSE;
	j->maxForce = maxForce;
	j->frequencyHz = frequencyHz;
	j->dampingRatio = dampingRatio;
	return id;
}

/* b2GearJoint::b2GearJoint (b2GearJoint.cpp:50-129) */
static float gear_coordinate(const b2o_world* w, const revolute_t* jt)
{
	const body_t* bm = &w->bodies[jt->bodyB];
	const body_t* bf = &w->bodies[jt->bodyA];
	if (jt->type == B2O_JOINT_REVOLUTE) return bm->a - bf->a - jt->referenceAngle;
	vec2 pC = jt->localAnchorA;
	vec2 pA = r_mul_t(bf->xf.q, v_add(r_mul(bm->xf.q, jt->localAnchorB), v_sub(bm->xf.p, bf->xf.p)));
	return v_dot(v_sub(pA, pC), jt->localXAxisA);
}

int b2o_create_gear_joint(b2o_world* w, int joint1, int joint2, float ratio, int collideConnected)
{
	float coordinateA = gear_coordinate(w, &w->joints[joint1]);
	float coordinateB = gear_coordinate(w, &w->joints[joint2]);
	int bodyA = w->joints[joint1].bodyB, bodyB = w->joints[joint2].bodyB;
	float anchors[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
	int id = b2o_create_revolute_joint(w, bodyA, bodyB, anchors, 0.0f, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f, collideConnected);
	revolute_t* j = &w->joints[id];
	const revolute_t* j1 = &w->joints[joint1];
	const revolute_t* j2 = &w->joints[joint2];
	j->type = B2O_JOINT_GEAR;
	j->bodyC = j1->bodyA; j->bodyD = j2->bodyA;
	j->typeA = j1->type; j->typeB = j2->type;
	j->gLocalAnchorC = j1->localAnchorA; j->gLocalAnchorA = j1->localAnchorB; j->referenceAngleA = j1->referenceAngle;
	j->localAxisC = j1->type == B2O_JOINT_PRISMATIC ? j1->localXAxisA : v_make(0.0f, 0.0f);
	j->gLocalAnchorD = j2->localAnchorA; j->gLocalAnchorB = j2->localAnchorB; j->referenceAngleB = j2->referenceAngle;
	j->localAxisD = j2->type == B2O_JOINT_PRISMATIC ? j2->localXAxisA : v_make(0.0f, 0.0f);
	j->ratio = ratio;
	j->constant = coordinateA + ratio * coordinateB;
	return id;
}

static void set_awake(body_t* b);

/* b2MouseJoint::SetTarget (b2MouseJoint.cpp:57-64): wakes bodyB only */
void b2o_joint_set_target(b2o_world* w, int joint, float tx, float ty)
{
	revolute_t* j = &w->joints[joint];
	if (tx == j->localAnchorA.x && ty == j->localAnchorA.y) return;
	set_awake(&w->bodies[j->bodyB]);
	j->localAnchorA = v_make(tx, ty);
}

/* b2MotorJoint::SetLinearOffset / SetAngularOffset (b2MotorJoint.cpp:253-281) */
void b2o_joint_set_offsets(b2o_world* w, int joint, float lx, float ly, float angular)
{
	revolute_t* j = &w->joints[joint];
	if (lx == j->localAnchorA.x && ly == j->localAnchorA.y && angular == j->referenceAngle) return;
	set_awake(&w->bodies[j->bodyA]); /* b2Body::SetAwake(true) always restarts the sleep timer (b2Body.h:699-703) */
	set_awake(&w->bodies[j->bodyB]);
	j->localAnchorA = v_make(lx, ly);
	j->referenceAngle = angular;
}

/* b2World::DestroyJoint (b2World.cpp:762-846) */
static void unlink_joint_edge(b2o_world* w, int body, int edge)
{
	int* link = &w->bodies[body].jointHead;
	while (*link >= 0 && *link != edge)
	{
		revolute_t* o = &w->joints[*link >> 1];
		link = (*link & 1) ? &o->nextB : &o->nextA;
	}
	if (*link == edge)
	{
		revolute_t* me = &w->joints[edge >> 1];
		*link = (edge & 1) ? me->nextB : me->nextA;
	}
}

void b2o_destroy_joint(b2o_world* w, int joint)
{
	revolute_t* j = &w->joints[joint];
	if (j->type < 0) return;
	set_awake(&w->bodies[j->bodyA]); /* b2Body::SetAwake(true) always restarts the sleep timer (b2Body.h:699-703) */
	set_awake(&w->bodies[j->bodyB]);
	unlink_joint_edge(w, j->bodyA, joint * 2);
	unlink_joint_edge(w, j->bodyB, joint * 2 + 1);
	if (!j->collideConnected)
	{
		for (int e = w->bodies[j->bodyB].contactHead; e >= 0; e = w->contacts[e >> 1].next[e & 1])
		{
			contact_t* c = &w->contacts[e >> 1];
			int other = (e & 1) == 0 ? c->bodyB : c->bodyA;
			if (other == j->bodyA) c->flags |= CF_FILTER;
		}
	}
	j->type = -1;
}

/* EnableMotor / SetMotorSpeed / SetMaxMotorTorque|Force (b2RevoluteJoint.cpp:418-452, b2PrismaticJoint.cpp:588-616) */
void b2o_joint_set_motor(b2o_world* w, int joint, int enableMotor, float motorSpeed, float maxMotor)
{
	revolute_t* j = &w->joints[joint];
	if ((enableMotor != 0) == (j->enableMotor != 0) && motorSpeed == j->motorSpeed && maxMotor == j->maxMotorTorque) return;
	set_awake(&w->bodies[j->bodyA]); /* b2Body::SetAwake(true) always restarts the sleep timer (b2Body.h:699-703) */
	set_awake(&w->bodies[j->bodyB]);
	j->enableMotor = enableMotor != 0;
	j->motorSpeed = motorSpeed;
	j->maxMotorTorque = maxMotor;
}

/* EnableLimit / SetLimits (b2RevoluteJoint.cpp:459-500, b2PrismaticJoint.cpp:549-581) */
void b2o_joint_set_limits(b2o_world* w, int joint, int enableLimit, float lower, float upper)
{
	revolute_t* j = &w->joints[joint];
	if ((enableLimit != 0) == (j->enableLimit != 0) && lower == j->lowerAngle && upper == j->upperAngle) return;
	set_awake(&w->bodies[j->bodyA]); /* b2Body::SetAwake(true) always restarts the sleep timer (b2Body.h:699-703) */
	set_awake(&w->bodies[j->bodyB]);
	j->enableLimit = enableLimit != 0;
	j->lowerAngle = lower;
	j->upperAngle = upper;
	j->impulse[2] = 0.0f;
}

void b2o_get_joint_reaction(const b2o_world* w, int joint, float inv_dt, float out4[4]) { b2o_joint_reaction(&w->joints[joint], inv_dt, out4); }

/* ---- contacts ------------------------------------------------------------------------------------ */
static int body_active_for_contact(const body_t* b) { return (b->flags & BF_AWAKE) != 0 && b->type != 0; }

/* b2Body::SetAwake(true)  b2Body.h:690-718 */
static void set_awake(body_t* b)
{
	b->flags |= BF_AWAKE;
	b->sleepTime = 0.0f;
}

/* b2ContactFilter::ShouldCollide  b2WorldCallbacks.cpp:24-38 */
static int filter_should_collide(const fixture_t* a, const fixture_t* b)
{
	if (a->groupIndex == b->groupIndex && a->groupIndex != 0) return a->groupIndex > 0;
	return (a->maskBits & b->categoryBits) != 0 && (a->categoryBits & b->maskBits) != 0;
}

/* which fixture becomes A: b2Contact::Create register table (b2Contact.cpp:42-52, 72-98) */
static int contact_swap(int t1, int t2)
{
	/* b2ChainAndCircleContact / b2ChainAndPolygonContact are primary like the edge forms (b2Contact.cpp:47-52) */
	if (t1 == SHAPE_CHAIN) t1 = SHAPE_EDGE;
	if (t2 == SHAPE_CHAIN) t2 = SHAPE_EDGE;
	if (t1 == SHAPE_CIRCLE && t2 == SHAPE_CIRCLE) return 0;
	if (t1 == SHAPE_POLYGON && t2 == SHAPE_CIRCLE) return 0;
	if (t1 == SHAPE_CIRCLE && t2 == SHAPE_POLYGON) return 1;
	if (t1 == SHAPE_POLYGON && t2 == SHAPE_POLYGON) return 0;
	if (t1 == SHAPE_EDGE && t2 == SHAPE_CIRCLE) return 0;
	if (t1 == SHAPE_CIRCLE && t2 == SHAPE_EDGE) return 1;
	if (t1 == SHAPE_EDGE && t2 == SHAPE_POLYGON) return 0;
	if (t1 == SHAPE_POLYGON && t2 == SHAPE_EDGE) return 1;
	return -1;
}

static int* edge_head(b2o_world* w, int contact, int side)
{
	contact_t* c = &w->contacts[contact];
	return &w->bodies[side == 0 ? c->bodyA : c->bodyB].contactHead;
}

/* b2Contact::IsToiCandidate  b2Contact.cpp:300-324 */
static int is_toi_candidate(const b2o_world* w, const fixture_t* fA, const fixture_t* fB)
{
	if (fA->isSensor || fB->isSensor) return 0;
	const body_t* bA = &w->bodies[fA->body];
	const body_t* bB = &w->bodies[fB->body];
	if ((bA->flags & BF_BULLET) || (bB->flags & BF_BULLET)) return 1;
	int includesNonDynamic = bA->type != 2 || bB->type != 2;
	int neitherThick = !fA->thick && !fB->thick;
	return includesNonDynamic && neitherThick;
}

/* b2ContactManager::AddToContactArray  b2ContactManager.cpp:659-686 */
static void carray_add(b2o_world* w, int slot)
{
	GROW(w->carray, w->capArr, w->nArr + 1, int);
	contact_t* c = &w->contacts[slot];
	if ((c->flags & CF_TOI_CANDIDATE) && w->toiCount < w->nArr)
	{
		int moved = w->carray[w->toiCount];
		w->contacts[moved].managerIndex = w->nArr;
		w->carray[w->nArr++] = moved;
		w->carray[w->toiCount] = slot;
		c->managerIndex = w->toiCount++;
		return;
	}
	c->managerIndex = w->nArr;
	w->carray[w->nArr++] = slot;
	if (c->flags & CF_TOI_CANDIDATE) w->toiCount++;
}

/* b2ContactManager::RemoveFromContactArray  b2ContactManager.cpp:688-714 */
static void carray_remove(b2o_world* w, int slot)
{
	contact_t* c = &w->contacts[slot];
	if (c->managerIndex < w->toiCount)
	{
		--w->toiCount;
		int lastToi = w->carray[w->toiCount];
		w->contacts[lastToi].managerIndex = c->managerIndex;
		w->carray[c->managerIndex] = lastToi;
		int back = w->carray[--w->nArr];
		if (w->nArr > w->toiCount)
		{
			w->carray[w->toiCount] = back;
			w->contacts[back].managerIndex = w->toiCount;
		}
	}
	else
	{
		int back = w->carray[--w->nArr];
		if (c->managerIndex < w->nArr)
		{
			w->contacts[back].managerIndex = c->managerIndex;
			w->carray[c->managerIndex] = back;
		}
	}
	c->managerIndex = -1;
}

/* b2ContactManager::OnContactCreate (:507-564) + b2Contact::b2Contact (b2Contact.cpp:125-159) */
static void create_contact(b2o_world* w, int fLo, int fHi)
{
	int fA = fLo, fB = fHi;
	int sw = contact_swap(w->fixtures[fA].shape.type, w->fixtures[fB].shape.type);
	if (sw < 0) return;
	if (sw == 1) { int t = fA; fA = fB; fB = t; }
	int slot;
	if (w->nFreeContacts > 0) slot = w->freeContacts[--w->nFreeContacts];
	else
	{
		GROW(w->contacts, w->capContacts, w->nContactSlots + 1, contact_t);
		slot = w->nContactSlots++;
	}
	contact_t* c = &w->contacts[slot];
	memset(c, 0, sizeof(*c));
	c->alive = 1;
	c->fixtureA = fA;
	c->fixtureB = fB;
	c->bodyA = w->fixtures[fA].body;
	c->bodyB = w->fixtures[fB].body;
	c->flags = CF_ENABLED;
	if (is_toi_candidate(w, &w->fixtures[fA], &w->fixtures[fB])) c->flags |= CF_TOI_CANDIDATE;
	c->toi = 1.0f;
	c->toiCount = 0;
	c->managerIndex = -1;
	c->proxyLo = w->fixtures[fLo].proxyId;
	c->proxyHi = w->fixtures[fHi].proxyId;
	c->friction = sqrtf(w->fixtures[fA].friction * w->fixtures[fB].friction);
	c->restitution = w->fixtures[fA].restitution > w->fixtures[fB].restitution ? w->fixtures[fA].restitution : w->fixtures[fB].restitution;
	c->tangentSpeed = 0.0f;
	c->seq = w->nextSeq++;
	if (!w->fixtures[fA].isSensor && !w->fixtures[fB].isSensor)
	{
		set_awake(&w->bodies[c->bodyA]);
		set_awake(&w->bodies[c->bodyB]);
	}
	for (int side = 0; side < 2; ++side)
	{
		int* head = edge_head(w, slot, side);
		c->prev[side] = -1;
		c->next[side] = *head;
		if (*head >= 0) w->contacts[*head >> 1].prev[*head & 1] = slot * 2 + side;
		*head = slot * 2 + side;
	}
	carray_add(w, slot);
	w->liveContacts++;
}

/* b2ContactManager::Destroy (:120-172) + b2Contact::Destroy (b2Contact.cpp:100-123) */
static void push_event(b2o_world* w, const contact_t* c, int kind, int index)
{
	GROW(w->events, w->capEvents, w->nEvents + 1, b2o_contact_event);
	GROW(w->eventKeys, w->capEventKeys, w->nEvents + 1, uint64_t);
	b2o_contact_event* e = &w->events[w->nEvents];
	e->fixture_a = c->fixtureA;
	e->fixture_b = c->fixtureB;
	e->kind = kind;
	e->contact_index = index;
	w->eventKeys[w->nEvents] = ((uint64_t)(uint32_t)c->proxyLo << 32) | (uint32_t)c->proxyHi;
	w->nEvents++;
}

static int sp_foreign_body(const b2o_world* w, int bi)
{
	return w->spatial && w->bodies[bi].type != 0 && w->owner[bi] != (unsigned char)w->shardRank;
}
static int sp_foreign_contact(const b2o_world* w, const contact_t* c)
{
	return w->spatial && (sp_foreign_body(w, c->bodyA) || sp_foreign_body(w, c->bodyB));
}
static void destroy_contact(b2o_world* w, int slot)
{
	contact_t* c = &w->contacts[slot];
	/* b2ContactManager::Destroy (b2ContactManager.cpp:104-107): a touching contact ends when it is destroyed */
	if (w->eventsOn && c->reported) push_event(w, c, 1, -1);
	c->reported = 0;
	/* (a foreign contact's point count is not maintained here: its owner wakes the bodies) */
	if (c->m.pointCount > 0 && !w->fixtures[c->fixtureA].isSensor && !w->fixtures[c->fixtureB].isSensor && !sp_foreign_contact(w, c))
	{
		set_awake(&w->bodies[c->bodyA]);
		set_awake(&w->bodies[c->bodyB]);
	}
	for (int side = 0; side < 2; ++side)
	{
		int p = c->prev[side], n = c->next[side];
		if (p >= 0) w->contacts[p >> 1].next[p & 1] = n;
		if (n >= 0) w->contacts[n >> 1].prev[n & 1] = p;
		int* head = edge_head(w, slot, side);
		if (*head == slot * 2 + side) *head = n;
	}
	carray_remove(w, slot);
	c->alive = 0;
	GROW(w->freeContacts, w->capFreeContacts, w->nFreeContacts + 1, int);
	w->freeContacts[w->nFreeContacts++] = slot;
	w->liveContacts--;
}

static int fat_overlap(const float a[4], const float b[4])
{
	/* b2TestOverlap(b2AABB)  b2Collision.h:273-286 */
	float d1x = b[0] - a[2], d1y = b[1] - a[3];
	float d2x = a[0] - b[2], d2y = a[1] - b[3];
	if (d1x > 0.0f || d1y > 0.0f) return 0;
	if (d2x > 0.0f || d2y > 0.0f) return 0;
	return 1;
}

/* b2Body::ShouldCollide  b2Body.cpp:428-449 */
static int bodies_should_collide_w(const b2o_world* w, int ia, int ib)
{
	const body_t* a = &w->bodies[ia];
	const body_t* b = &w->bodies[ib];
	if (a->type != 2 && b->type != 2) return 0;
	for (int e = a->jointHead; e >= 0; )
	{
		const revolute_t* j = &w->joints[e >> 1];
		int other = (e & 1) == 0 ? j->bodyB : j->bodyA;
		if (other == ib && j->collideConnected == 0) return 0;
		e = (e & 1) == 0 ? j->nextA : j->nextB;
	}
	return 1;
}

/* b2Contact::UpdateImpl (b2Contact.cpp:173-298) without the wake-up, which differs between the
 * multi-threaded Collide (deferred, body A only) and the single-threaded TOI path (both bodies) */
static void contact_update(b2o_world* w, contact_t* c)
{
	const fixture_t* fA = &w->fixtures[c->fixtureA];
	const fixture_t* fB = &w->fixtures[c->fixtureB];
	manifold old = c->m;
	c->flags |= CF_ENABLED;
	int touching = 0;
	if (fA->isSensor || fB->isSensor)
	{
		/* b2Contact::Update, sensor branch (b2Contact.cpp:193-202): touching = b2TestOverlap (b2Collision.cpp:233-252),
		 * i.e. GJK distance with the shape radii < 10 * epsilon; sensors generate no manifold */
		gjk_proxy pA, pB;
		b2o_proxy_set(&pA, &fA->shape);
		b2o_proxy_set(&pB, &fB->shape);
		gjk_cache cache;
		memset(&cache, 0, sizeof(cache));
		gjk_output out;
		b2o_distance(&out, &cache, &pA, w->bodies[c->bodyA].xf, &pB, w->bodies[c->bodyB].xf, 1);
		touching = out.distance < 10.0f * B2O_EPSILON;
		c->m.pointCount = 0;
	}
	else
	{
		b2o_evaluate(&c->m, &fA->shape, w->bodies[c->bodyA].xf, &fB->shape, w->bodies[c->bodyB].xf);
		touching = c->m.pointCount > 0;
		for (int k = 0; k < c->m.pointCount; ++k)
		{
			c->m.ni[k] = 0.0f;
			c->m.ti[k] = 0.0f;
			for (int j = 0; j < old.pointCount; ++j)
			{
				if (old.id[j] == c->m.id[k])
				{
					c->m.ni[k] = old.ni[j];
					c->m.ti[k] = old.ti[j];
					break;
				}
			}
		}
	}
	if (touching) c->flags |= CF_TOUCHING; else c->flags &= ~CF_TOUCHING;
	c->oldm = old;
	c->preSolveDue = touching && !(fA->isSensor || fB->isSensor);
}

static const b2o_world* destroy_cmp_world;
static int destroy_cmp(const void* a, const void* b)
{
	const contact_t* p = &destroy_cmp_world->contacts[*(const int*)a];
	const contact_t* q = &destroy_cmp_world->contacts[*(const int*)b];
	if (p->proxyLo != q->proxyLo) return p->proxyLo < q->proxyLo ? -1 : 1;
	if (p->proxyHi != q->proxyHi) return p->proxyHi < q->proxyHi ? -1 : 1;
	return 0;
}

static int seq_cmp(const void* a, const void* b);

/* index of every live contact in b2o_get_contacts order (creation order); -1 for dead slots */
static int* contact_ranks(const b2o_world* w)
{
	const contact_t** live = (const contact_t**)malloc(sizeof(void*) * (size_t)(w->liveContacts + 1));
	int* rank = (int*)malloc(sizeof(int) * (size_t)(w->nContactSlots + 1));
	int n = 0;
	for (int i = 0; i < w->nContactSlots; ++i)
	{
		rank[i] = -1;
		if (w->contacts[i].alive) live[n++] = &w->contacts[i];
	}
	qsort(live, (size_t)n, sizeof(void*), seq_cmp);
	for (int i = 0; i < n; ++i) rank[(int)(live[i] - w->contacts)] = i;
	free(live);
	return rank;
}

static void fill_manifold(b2o_manifold* o, const manifold* m)
{
	o->type = m->type;
	o->point_count = m->pointCount;
	o->local_normal[0] = m->localNormal.x; o->local_normal[1] = m->localNormal.y;
	o->local_point[0] = m->localPoint.x; o->local_point[1] = m->localPoint.y;
	for (int k = 0; k < 2; ++k)
	{
		o->point_local[k][0] = m->p[k].x; o->point_local[k][1] = m->p[k].y;
		o->normal_impulse[k] = m->ni[k]; o->tangent_impulse[k] = m->ti[k];
		o->id_key[k] = m->id[k];
	}
}

/* b2ContactListener::PreSolve, deferred form (b2Contact.cpp:283-297, b2ContactManager.cpp:431-434): after the destroys of
 * FinishCollide here, so that the contact index is the one b2o_get_contacts shows; proxy-id-pair order */
static void deliver_pre_solve(b2o_world* w)
{
	int n = 0;
	int* due = (int*)malloc(sizeof(int) * (size_t)(w->nContactSlots + 1));
	for (int i = 0; i < w->nContactSlots; ++i)
		if (w->contacts[i].alive && w->contacts[i].preSolveDue) due[n++] = i;
	destroy_cmp_world = w;
	qsort(due, (size_t)n, sizeof(int), destroy_cmp);
	int* rank = contact_ranks(w);
	b2o_pre_solve_record* recs = (b2o_pre_solve_record*)malloc(sizeof(b2o_pre_solve_record) * (size_t)(n + 1));
	for (int k = 0; k < n; ++k)
	{
		contact_t* c = &w->contacts[due[k]];
		b2o_pre_solve_record* r = &recs[k];
		r->contact_index = rank[due[k]];
		r->fixture_a = c->fixtureA;
		r->fixture_b = c->fixtureB;
		r->enabled = 1;
		fill_manifold(&r->old_manifold, &c->oldm);
		fill_manifold(&r->manifold, &c->m);
		r->material.friction = c->friction;
		r->material.restitution = c->restitution;
		r->material.tangent_speed = c->tangentSpeed;
	}
	if (w->preSolveBatchFn) w->preSolveBatchFn(w->preSolveUser, n, recs);
	else for (int k = 0; k < n; ++k)
	{
		b2o_pre_solve_record* r = &recs[k];
		r->enabled = w->preSolveFn(w->preSolveUser, r->contact_index, r->fixture_a, r->fixture_b, &r->old_manifold, &r->manifold, &r->material) != 0;
	}
	for (int k = 0; k < n; ++k)
	{
		contact_t* c = &w->contacts[due[k]];
		if (!recs[k].enabled) c->flags &= ~CF_ENABLED;
		c->friction = recs[k].material.friction; /* (the setters' values stay with the contact, b2Contact.h:129-160) */
		c->restitution = recs[k].material.restitution;
		c->tangentSpeed = recs[k].material.tangent_speed;
	}
	free(recs);
	free(rank);
	free(due);
}

/* b2ContactManager::Collide (:177-230) + FinishCollide (:388-439: destroys sorted by proxy ids) */
static void sp_exchange_state(b2o_world* w, int afterToi);
static void sp_resolve(b2o_world* w);

static void collide(b2o_world* w)
{
	int nSlots = w->nContactSlots;
	int* destroys = (int*)malloc(sizeof(int) * (size_t)(nSlots + 1));
	int* awakes = (int*)malloc(sizeof(int) * (size_t)(nSlots + 1));
	int nDestroy = 0, nAwake = 0;
	for (int i = 0; i < nSlots; ++i)
	{
		contact_t* c = &w->contacts[i];
		if (!c->alive) continue;
		fixture_t* fA = &w->fixtures[c->fixtureA];
		fixture_t* fB = &w->fixtures[c->fixtureB];
		body_t* bA = &w->bodies[c->bodyA];
		body_t* bB = &w->bodies[c->bodyB];
		c->preSolveDue = 0;
		if (c->flags & CF_FILTER)
		{
			/* (a user contact filter replaces the default one, b2ContactManager.cpp:195-203) */
			int pass = w->filterFn ? w->filterFn(w->filterUser, c->fixtureA, c->fixtureB) != 0 : filter_should_collide(fA, fB);
			if (!bodies_should_collide_w(w, c->bodyB, c->bodyA) || !pass)
			{
				destroys[nDestroy++] = i;
				continue;
			}
			c->flags &= ~CF_FILTER;
		}
		if (!body_active_for_contact(bA) && !body_active_for_contact(bB)) continue; /* e_inactiveFlag */
		if (!fat_overlap(fA->fat, fB->fat))
		{
			destroys[nDestroy++] = i;
			continue;
		}
		/* (a spatially sharded world: the bodies are another rank's, which evaluates the manifold; here the contact only keeps
		 * existing - the tests above are structure, the same on every rank) */
		if (sp_foreign_contact(w, c)) continue;
		int wasTouching = (c->flags & CF_TOUCHING) != 0;
		contact_update(w, c);
		int sensor = fA->isSensor || fB->isSensor;
		if (!sensor && ((c->flags & CF_TOUCHING) != 0) != wasTouching) awakes[nAwake++] = i;
	}
	/* ConsumeAwakes (:472-485): only fixture A's body (m_nodeB.other twice) */
	for (int k = 0; k < nAwake; ++k) set_awake(&w->bodies[w->contacts[awakes[k]].bodyA]);
	destroy_cmp_world = w;
	qsort(destroys, (size_t)nDestroy, sizeof(int), destroy_cmp);
	for (int k = 0; k < nDestroy; ++k) destroy_contact(w, destroys[k]);
	free(destroys);
	free(awakes);
	if (w->preSolveFn || w->preSolveBatchFn) deliver_pre_solve(w);
}

/* ---- broad-phase --------------------------------------------------------------------------------- */
typedef struct { int lo, hi; int fLo, fHi; } pair_t;

static int pair_cmp(const void* a, const void* b)
{
	const pair_t* p = (const pair_t*)a;
	const pair_t* q = (const pair_t*)b;
	if (p->lo != q->lo) return p->lo < q->lo ? -1 : 1;
	if (p->hi != q->hi) return p->hi < q->hi ? -1 : 1;
	return 0;
}

/* b2BroadPhase::UpdatePairs (b2BroadPhase.h:211-267) + b2ContactManager::AddPair (:237-312) +
 * FinishFindNewContacts (:366-386) */
static void sp_exchange_pairs(b2o_world* w, pair_t** pairs, int* n, int* cap);

static void find_new_contacts(b2o_world* w)
{
	if (w->nMoves == 0 && !(w->spatial && w->shardCount > 1)) return; /* (every rank of a sharded world takes part in the exchange) */
	int cap = 1024, n = 0;
	pair_t* pairs = (pair_t*)malloc(sizeof(pair_t) * (size_t)cap);
	for (int k = 0; k < w->nMoves; ++k)
	{
		int p = w->moves[k];
		const fixture_t* fp = &w->fixtures[p];
		if (fp->dead) continue;
		if (sp_foreign_body(w, fp->body)) continue; /* every rank searches for the proxies ITS bodies moved */
		for (int q = 0; q < w->nFixtures; ++q)
		{
			if (q == p) continue;
			const fixture_t* fq = &w->fixtures[q];
			if (fq->dead) continue;
			if (!fat_overlap(fp->fat, fq->fat)) continue;
			if (n == cap)
			{
				cap *= 2;
				pairs = (pair_t*)realloc(pairs, sizeof(pair_t) * (size_t)cap);
			}
			pair_t pr;
			if (fp->proxyId < fq->proxyId) { pr.lo = fp->proxyId; pr.hi = fq->proxyId; pr.fLo = p; pr.fHi = q; }
			else { pr.lo = fq->proxyId; pr.hi = fp->proxyId; pr.fLo = q; pr.fHi = p; }
			pairs[n++] = pr;
		}
	}
	if (w->spatial && w->shardCount > 1) sp_exchange_pairs(w, &pairs, &n, &cap); /* all ranks order and create the union */
	qsort(pairs, (size_t)n, sizeof(pair_t), pair_cmp);
	int prevLo = -1, prevHi = -1;
	for (int i = 0; i < n; ++i)
	{
		if (pairs[i].lo == prevLo && pairs[i].hi == prevHi) continue;
		prevLo = pairs[i].lo;
		prevHi = pairs[i].hi;
		const fixture_t* fA = &w->fixtures[pairs[i].fLo];
		const fixture_t* fB = &w->fixtures[pairs[i].fHi];
		if (fA->body == fB->body) continue;
		/* does a contact already exist? scan bodyB's list like the reference */
		int exists = 0;
		for (int e = w->bodies[fB->body].contactHead; e >= 0; e = w->contacts[e >> 1].next[e & 1])
		{
			const contact_t* c = &w->contacts[e >> 1];
			if (c->proxyLo == pairs[i].lo && c->proxyHi == pairs[i].hi)
			{
				exists = 1;
				break;
			}
		}
		if (exists) continue;
		if (!bodies_should_collide_w(w, fB->body, fA->body)) continue;
		/* m_contactFilter->ShouldCollide (b2ContactManager.cpp:283-287): a user filter replaces the default one */
		if (w->filterFn ? !w->filterFn(w->filterUser, pairs[i].fLo, pairs[i].fHi) : !filter_should_collide(fA, fB)) continue;
		create_contact(w, pairs[i].fLo, pairs[i].fHi);
	}
	free(pairs);
	w->nMoves = 0;
	if (w->spatial) sp_resolve(w); /* a new contact may join components of different owners */
}

/* b2Body::SynchronizeFixtures (b2Body.cpp:475-489) + b2Fixture::Synchronize (b2Fixture.cpp:143-163) +
 * b2DynamicTree::MoveProxy (b2DynamicTree.cpp:130-174) */
static void sync_body_fixtures(b2o_world* w, body_t* b)
{
	xform xf1;
	xf1.q = r_make(b->a0);
	xf1.p = v_sub(b->c0, r_mul(xf1.q, b->localCenter));
	for (int f = b->fixtureHead; f >= 0; f = w->fixtures[f].nextInBody)
	{
		fixture_t* fx = &w->fixtures[f];
		if (fx->proxyId < 0) continue; /* (an inactive body's fixtures have no proxies) */
		float a1[4], a2[4], aabb[4];
		shape_aabb(&fx->shape, xf1, a1);
		shape_aabb(&fx->shape, b->xf, a2);
		aabb[0] = f_min(a1[0], a2[0]); aabb[1] = f_min(a1[1], a2[1]);
		aabb[2] = f_max(a1[2], a2[2]); aabb[3] = f_max(a1[3], a2[3]);
		int contains = fx->fat[0] <= aabb[0] && fx->fat[1] <= aabb[1] && aabb[2] <= fx->fat[2] && aabb[3] <= fx->fat[3];
		if (contains) continue;
		vec2 disp = v_sub(b->xf.p, xf1.p);
		float lo0 = aabb[0] - B2O_AABB_EXTENSION, lo1 = aabb[1] - B2O_AABB_EXTENSION;
		float hi0 = aabb[2] + B2O_AABB_EXTENSION, hi1 = aabb[3] + B2O_AABB_EXTENSION;
		vec2 d = v_scale(B2O_AABB_MULTIPLIER, disp);
		if (d.x < 0.0f) lo0 += d.x; else hi0 += d.x;
		if (d.y < 0.0f) lo1 += d.y; else hi1 += d.y;
		fx->fat[0] = lo0; fx->fat[1] = lo1; fx->fat[2] = hi0; fx->fat[3] = hi1;
		GROW(w->moves, w->capMoves, w->nMoves + 1, int);
		w->moves[w->nMoves++] = f;
	}
}

/* b2ContactManager::SynchronizeFixtures (:315-364) + FinishSynchronizeFixtures (:441-452) */
static void synchronize_fixtures(b2o_world* w)
{
	for (int i = 0; i < w->nBodies; ++i)
	{
		body_t* b = &w->bodies[i];
		if (b->type == 0) continue;
		if ((b->flags & BF_ISLAND) == 0) continue;
		sync_body_fixtures(w, b);
	}
}

/* ---- island solve -------------------------------------------------------------------------------- */
typedef struct { vec2 c; float a; } pos_t;
typedef struct { vec2 v; float w; } vel_t;

/* b2ContactSolver::b2ContactSolver (:47-133) + InitializeVelocityConstraints (:142-251) */
static void init_constraint(b2o_world* w, constraint_t* cc, int contact, const pos_t* positions, const vel_t* velocities,
	float dtRatio)
{
	contact_t* c = &w->contacts[contact];
	body_t* bA = &w->bodies[c->bodyA];
	body_t* bB = &w->bodies[c->bodyB];
	const manifold* mf = &c->m;
	memset(cc, 0, sizeof(*cc));
	cc->contact = contact;
	cc->friction = c->friction;
	cc->restitution = c->restitution;
	cc->tangentSpeed = c->tangentSpeed;
	cc->indexA = bA->islandIndex;
	cc->indexB = bB->islandIndex;
	cc->invMassA = bA->invMass; cc->invMassB = bB->invMass;
	cc->invIA = bA->invI; cc->invIB = bB->invI;
	cc->pointCount = mf->pointCount;
	cc->localCenterA = bA->localCenter; cc->localCenterB = bB->localCenter;
	cc->localNormal = mf->localNormal; cc->localPoint = mf->localPoint;
	cc->pcPointCount = mf->pointCount;
	cc->radiusA = w->fixtures[c->fixtureA].shape.radius;
	cc->radiusB = w->fixtures[c->fixtureB].shape.radius;
	cc->type = mf->type;
	for (int j = 0; j < mf->pointCount; ++j)
	{
		if (w->warmStarting)
		{
			cc->normalImpulse[j] = dtRatio * mf->ni[j];
			cc->tangentImpulse[j] = dtRatio * mf->ti[j];
		}
		cc->localPoints[j] = mf->p[j];
	}
	float mA = cc->invMassA, mB = cc->invMassB, iA = cc->invIA, iB = cc->invIB;
	vec2 cA = positions[cc->indexA].c, cB = positions[cc->indexB].c;
	float aA = positions[cc->indexA].a, aB = positions[cc->indexB].a;
	vec2 vA = velocities[cc->indexA].v, vB = velocities[cc->indexB].v;
	float wA = velocities[cc->indexA].w, wB = velocities[cc->indexB].w;
	xform xfA, xfB;
	xfA.q = r_make(aA);
	xfB.q = r_make(aB);
	xfA.p = v_sub(cA, r_mul(xfA.q, cc->localCenterA));
	xfB.p = v_sub(cB, r_mul(xfB.q, cc->localCenterB));
	/* b2WorldManifold::Initialize  b2Collision.cpp:22-86 */
	vec2 normal = v_make(0, 0), points[2];
	points[0] = points[1] = v_make(0, 0);
	if (mf->type == MANIFOLD_CIRCLES)
	{
		normal = v_make(1.0f, 0.0f);
		vec2 pointA = xf_mul(xfA, mf->localPoint);
		vec2 pointB = xf_mul(xfB, mf->p[0]);
		if (v_dist_sq(pointA, pointB) > B2O_EPSILON * B2O_EPSILON)
		{
			normal = v_sub(pointB, pointA);
			v_normalize(&normal);
		}
		vec2 ca = v_add(pointA, v_scale(cc->radiusA, normal));
		vec2 cb = v_sub(pointB, v_scale(cc->radiusB, normal));
		points[0] = v_scale(0.5f, v_add(ca, cb));
	}
	else if (mf->type == MANIFOLD_FACE_A)
	{
		normal = r_mul(xfA.q, mf->localNormal);
		vec2 planePoint = xf_mul(xfA, mf->localPoint);
		for (int i = 0; i < mf->pointCount; ++i)
		{
			vec2 clip = xf_mul(xfB, mf->p[i]);
			vec2 ca = v_add(clip, v_scale(cc->radiusA - v_dot(v_sub(clip, planePoint), normal), normal));
			vec2 cb = v_sub(clip, v_scale(cc->radiusB, normal));
			points[i] = v_scale(0.5f, v_add(ca, cb));
		}
	}
	else
	{
		normal = r_mul(xfB.q, mf->localNormal);
		vec2 planePoint = xf_mul(xfB, mf->localPoint);
		for (int i = 0; i < mf->pointCount; ++i)
		{
			vec2 clip = xf_mul(xfA, mf->p[i]);
			vec2 cb = v_add(clip, v_scale(cc->radiusB - v_dot(v_sub(clip, planePoint), normal), normal));
			vec2 ca = v_sub(clip, v_scale(cc->radiusA, normal));
			points[i] = v_scale(0.5f, v_add(ca, cb));
		}
		normal = v_neg(normal);
	}
	cc->normal = normal;
	for (int j = 0; j < cc->pointCount; ++j)
	{
		cc->rA[j] = v_sub(points[j], cA);
		cc->rB[j] = v_sub(points[j], cB);
		float rnA = v_cross(cc->rA[j], normal), rnB = v_cross(cc->rB[j], normal);
		float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
		cc->normalMass[j] = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
		vec2 tangent = v_cross_vs(normal, 1.0f);
		float rtA = v_cross(cc->rA[j], tangent), rtB = v_cross(cc->rB[j], tangent);
		float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
		cc->tangentMass[j] = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
		cc->velocityBias[j] = 0.0f;
		vec2 rel = v_sub(v_sub(v_add(vB, v_cross_sv(wB, cc->rB[j])), vA), v_cross_sv(wA, cc->rA[j]));
		float vRel = v_dot(normal, rel);
		if (vRel < -B2O_VELOCITY_THRESHOLD) cc->velocityBias[j] = -cc->restitution * vRel;
	}
	if (cc->pointCount == 2)
	{
		float rn1A = v_cross(cc->rA[0], normal), rn1B = v_cross(cc->rB[0], normal);
		float rn2A = v_cross(cc->rA[1], normal), rn2B = v_cross(cc->rB[1], normal);
		float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
		float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
		float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
		if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12))
		{
			cc->K[0] = k11; cc->K[1] = k12; cc->K[2] = k12; cc->K[3] = k22;
			float a = k11, b = k12, c2 = k12, d = k22;
			float det = a * d - b * c2;
			if (det != 0.0f) det = 1.0f / det;
			cc->nm[0] = det * d;   /* ex.x */
			cc->nm[2] = -det * b;  /* ey.x */
			cc->nm[1] = -det * c2; /* ex.y */
			cc->nm[3] = det * a;   /* ey.y */
		}
		else
		{
			cc->pointCount = 1;
		}
	}
}

/* b2ContactSolver::WarmStart  :253-291 */
static void warm_start(const constraint_t* cc, vel_t* vel)
{
	float mA = cc->invMassA, iA = cc->invIA, mB = cc->invMassB, iB = cc->invIB;
	vec2 vA = vel[cc->indexA].v, vB = vel[cc->indexB].v;
	float wA = vel[cc->indexA].w, wB = vel[cc->indexB].w;
	vec2 normal = cc->normal, tangent = v_cross_vs(normal, 1.0f);
	for (int j = 0; j < cc->pointCount; ++j)
	{
		vec2 P = v_add(v_scale(cc->normalImpulse[j], normal), v_scale(cc->tangentImpulse[j], tangent));
		wA -= iA * v_cross(cc->rA[j], P);
		vA = v_sub(vA, v_scale(mA, P));
		wB += iB * v_cross(cc->rB[j], P);
		vB = v_add(vB, v_scale(mB, P));
	}
	vel[cc->indexA].v = vA; vel[cc->indexA].w = wA;
	vel[cc->indexB].v = vB; vel[cc->indexB].w = wB;
}

static vec2 rel_vel(vec2 vA, float wA, vec2 vB, float wB, vec2 rA, vec2 rB)
{
	return v_sub(v_sub(v_add(vB, v_cross_sv(wB, rB)), vA), v_cross_sv(wA, rA));
}

/* b2ContactSolver::SolveVelocityConstraints  :293-603 */
static void solve_velocity(constraint_t* cc, vel_t* vel)
{
	float mA = cc->invMassA, iA = cc->invIA, mB = cc->invMassB, iB = cc->invIB;
	vec2 vA = vel[cc->indexA].v, vB = vel[cc->indexB].v;
	float wA = vel[cc->indexA].w, wB = vel[cc->indexB].w;
	vec2 normal = cc->normal, tangent = v_cross_vs(normal, 1.0f);
	for (int j = 0; j < cc->pointCount; ++j)
	{
		vec2 dv = rel_vel(vA, wA, vB, wB, cc->rA[j], cc->rB[j]);
		float vt = v_dot(dv, tangent) - cc->tangentSpeed;
		float lambda = cc->tangentMass[j] * (-vt);
		float maxFriction = cc->friction * cc->normalImpulse[j];
		float newImpulse = f_clamp(cc->tangentImpulse[j] + lambda, -maxFriction, maxFriction);
		lambda = newImpulse - cc->tangentImpulse[j];
		cc->tangentImpulse[j] = newImpulse;
		vec2 P = v_scale(lambda, tangent);
		vA = v_sub(vA, v_scale(mA, P));
		wA -= iA * v_cross(cc->rA[j], P);
		vB = v_add(vB, v_scale(mB, P));
		wB += iB * v_cross(cc->rB[j], P);
	}
	if (cc->pointCount == 1)
	{
		vec2 dv = rel_vel(vA, wA, vB, wB, cc->rA[0], cc->rB[0]);
		float vn = v_dot(dv, normal);
		float lambda = -cc->normalMass[0] * (vn - cc->velocityBias[0]);
		float newImpulse = f_max(cc->normalImpulse[0] + lambda, 0.0f);
		lambda = newImpulse - cc->normalImpulse[0];
		cc->normalImpulse[0] = newImpulse;
		vec2 P = v_scale(lambda, normal);
		vA = v_sub(vA, v_scale(mA, P));
		wA -= iA * v_cross(cc->rA[0], P);
		vB = v_add(vB, v_scale(mB, P));
		wB += iB * v_cross(cc->rB[0], P);
	}
	else
	{
		vec2 a = v_make(cc->normalImpulse[0], cc->normalImpulse[1]);
		vec2 dv1 = rel_vel(vA, wA, vB, wB, cc->rA[0], cc->rB[0]);
		vec2 dv2 = rel_vel(vA, wA, vB, wB, cc->rA[1], cc->rB[1]);
		float vn1 = v_dot(dv1, normal), vn2 = v_dot(dv2, normal);
		vec2 b = v_make(vn1 - cc->velocityBias[0], vn2 - cc->velocityBias[1]);
		b.x -= cc->K[0] * a.x + cc->K[2] * a.y;
		b.y -= cc->K[1] * a.x + cc->K[3] * a.y;
		vec2 x;
		int found = 0;
		x = v_make(-(cc->nm[0] * b.x + cc->nm[2] * b.y), -(cc->nm[1] * b.x + cc->nm[3] * b.y));
		if (x.x >= 0.0f && x.y >= 0.0f) found = 1;
		if (!found)
		{
			x.x = -cc->normalMass[0] * b.x;
			x.y = 0.0f;
			vn2 = cc->K[1] * x.x + b.y;
			if (x.x >= 0.0f && vn2 >= 0.0f) found = 1;
		}
		if (!found)
		{
			x.x = 0.0f;
			x.y = -cc->normalMass[1] * b.y;
			vn1 = cc->K[2] * x.y + b.x;
			if (x.y >= 0.0f && vn1 >= 0.0f) found = 1;
		}
		if (!found)
		{
			x.x = 0.0f;
			x.y = 0.0f;
			vn1 = b.x;
			vn2 = b.y;
			if (vn1 >= 0.0f && vn2 >= 0.0f) found = 1;
		}
		if (found)
		{
			vec2 d = v_sub(x, a);
			vec2 P1 = v_scale(d.x, normal), P2 = v_scale(d.y, normal);
			vA = v_sub(vA, v_scale(mA, v_add(P1, P2)));
			wA -= iA * (v_cross(cc->rA[0], P1) + v_cross(cc->rA[1], P2));
			vB = v_add(vB, v_scale(mB, v_add(P1, P2)));
			wB += iB * (v_cross(cc->rB[0], P1) + v_cross(cc->rB[1], P2));
			cc->normalImpulse[0] = x.x;
			cc->normalImpulse[1] = x.y;
		}
	}
	vel[cc->indexA].v = vA; vel[cc->indexA].w = wA;
	vel[cc->indexB].v = vB; vel[cc->indexB].w = wB;
}

/* b2ContactSolver::SolvePositionConstraints (:676-752) with b2PositionSolverManifold (:620-673), one constraint */
static float solve_position(const constraint_t* cc, pos_t* pos, float minSeparation)
{
	float mA = cc->invMassA, iA = cc->invIA, mB = cc->invMassB, iB = cc->invIB;
	vec2 cA = pos[cc->indexA].c, cB = pos[cc->indexB].c;
	float aA = pos[cc->indexA].a, aB = pos[cc->indexB].a;
	for (int j = 0; j < cc->pcPointCount; ++j)
	{
		xform xfA, xfB;
		xfA.q = r_make(aA);
		xfB.q = r_make(aB);
		xfA.p = v_sub(cA, r_mul(xfA.q, cc->localCenterA));
		xfB.p = v_sub(cB, r_mul(xfB.q, cc->localCenterB));
		vec2 normal, point;
		float separation;
		if (cc->type == MANIFOLD_CIRCLES)
		{
			vec2 pointA = xf_mul(xfA, cc->localPoint);
			vec2 pointB = xf_mul(xfB, cc->localPoints[0]);
			normal = v_sub(pointB, pointA);
			v_normalize(&normal);
			point = v_scale(0.5f, v_add(pointA, pointB));
			separation = v_dot(v_sub(pointB, pointA), normal) - cc->radiusA - cc->radiusB;
		}
		else if (cc->type == MANIFOLD_FACE_A)
		{
			normal = r_mul(xfA.q, cc->localNormal);
			vec2 planePoint = xf_mul(xfA, cc->localPoint);
			vec2 clip = xf_mul(xfB, cc->localPoints[j]);
			separation = v_dot(v_sub(clip, planePoint), normal) - cc->radiusA - cc->radiusB;
			point = clip;
		}
		else
		{
			normal = r_mul(xfB.q, cc->localNormal);
			vec2 planePoint = xf_mul(xfB, cc->localPoint);
			vec2 clip = xf_mul(xfA, cc->localPoints[j]);
			separation = v_dot(v_sub(clip, planePoint), normal) - cc->radiusA - cc->radiusB;
			point = clip;
			normal = v_neg(normal);
		}
		vec2 rA = v_sub(point, cA), rB = v_sub(point, cB);
		minSeparation = f_min(minSeparation, separation);
		float C = f_clamp(B2O_BAUMGARTE * (separation + B2O_LINEAR_SLOP), -B2O_MAX_LINEAR_CORRECTION, 0.0f);
		float rnA = v_cross(rA, normal), rnB = v_cross(rB, normal);
		float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
		float impulse = K > 0.0f ? -C / K : 0.0f;
		vec2 P = v_scale(impulse, normal);
		cA = v_sub(cA, v_scale(mA, P));
		aA -= iA * v_cross(rA, P);
		cB = v_add(cB, v_scale(mB, P));
		aB += iB * v_cross(rB, P);
	}
	pos[cc->indexA].c = cA; pos[cc->indexA].a = aA;
	pos[cc->indexB].c = cB; pos[cc->indexB].a = aB;
	return minSeparation;
}

/* b2Island::Solve  b2Island.cpp:184-396 */
/* the gear joint's four bodies in island arrays, A, B, C, D (copies, written back in that order: b2GearJoint.cpp:215-222) */
static void gear_gather(b2o_world* w, const revolute_t* j, const pos_t* positions, const vel_t* velocities, int idx[4], gear_bodies* g)
{
	idx[0] = w->bodies[j->bodyA].islandIndex; idx[1] = w->bodies[j->bodyB].islandIndex;
	idx[2] = w->bodies[j->bodyC].islandIndex; idx[3] = w->bodies[j->bodyD].islandIndex;
	for (int k = 0; k < 4; ++k)
	{
		g->c[k] = positions[idx[k]].c; g->a[k] = positions[idx[k]].a;
		g->v[k] = velocities[idx[k]].v; g->w[k] = velocities[idx[k]].w;
	}
}

#ifdef B2O_ORDER_EXPERIMENT
#include <stdio.h>
/* ---- ORDER EXPERIMENT (compiled only with -DB2O_ORDER_EXPERIMENT by tools/order_experiment.py; never in liboracle.so) -----
 * How much of the one-step deviation of a reordered Gauss-Seidel sweep can a better visiting order buy? With the environment
 * variable B2O_ORDER = k the constraints of every island of more than 1 000 contacts are visited in another order than the
 * reference's (b2Island.cpp:184-396 solves them in island order): 1 greedy colouring by hashed priority (what the device's
 * coloured solver does), 2 bottom-up by contact height, 3 / 6 greedy colouring with a spatial priority, 4 the reference
 * order's own dependency levels wrapped modulo B2O_ORDER_D (flips about 1 / D of the pairwise precedences), 5 the same on
 * the bottom-up order, 7 the reference's order inside the blocks of a spatial partition (cells of B2O_ORDER_D metres) and a
 * coloured order on the cut constraints only. Result on Pyramid 141 at steps 245 / 300: DESIGN.md section 3. */
static const b2o_world* exw;
static float* exkey;
static int ex_cmp(const void* a, const void* b)
{
	int ia = *(const int*)a, ib = *(const int*)b;
	if (exkey[ia] < exkey[ib]) return -1;
	if (exkey[ia] > exkey[ib]) return 1;
	return ia - ib;
}
static float ex_y(const b2o_world* w, int slot)
{
	const contact_t* c = &w->contacts[slot];
	const body_t* a = &w->bodies[c->bodyA];
	const body_t* b = &w->bodies[c->bodyB];
	if (a->type == 0) return b->c.y - 0.5f;
	if (b->type == 0) return a->c.y - 0.5f;
	return 0.5f * (a->c.y + b->c.y);
}
static float ex_x(const b2o_world* w, int slot)
{
	const contact_t* c = &w->contacts[slot];
	const body_t* a = &w->bodies[c->bodyA];
	const body_t* b = &w->bodies[c->bodyB];
	if (a->type == 0) return b->c.x;
	if (b->type == 0) return a->c.x;
	return 0.5f * (a->c.x + b->c.x);
}
static void ex_reorder(b2o_world* w, int* ic, int n, int mode, int D)
{
	int* idx = (int*)malloc(sizeof(int) * n);
	float* key = (float*)malloc(sizeof(float) * n);
	int* out = (int*)malloc(sizeof(int) * n);
	for (int i = 0; i < n; ++i) idx[i] = i;
	exkey = key;
	if (mode == 2)
	{
		for (int i = 0; i < n; ++i) key[i] = ex_y(w, ic[i]);
	}
	else if (mode == 1 || mode == 3 || mode == 6)
	{
		/* greedy colouring in priority order; colour classes ascending */
		float* pri = (float*)malloc(sizeof(float) * n);
		for (int i = 0; i < n; ++i) pri[i] = mode == 1 ? (float)(((uint32_t)ic[i] * 2654435761u) >> 8) : (mode == 3 ? ex_y(w, ic[i]) : ex_y(w, ic[i]) + 0.37f * ex_x(w, ic[i]));
		exkey = pri;
		qsort(idx, n, sizeof(int), ex_cmp);
		uint64_t* used = (uint64_t*)calloc(w->nBodies, sizeof(uint64_t));
		int maxc = 0;
		for (int k = 0; k < n; ++k)
		{
			int i = idx[k];
			const contact_t* c = &w->contacts[ic[i]];
			uint64_t m = 0;
			if (w->bodies[c->bodyA].type != 0) m |= used[c->bodyA];
			if (w->bodies[c->bodyB].type != 0) m |= used[c->bodyB];
			int col = 0;
			while (m & (1ull << col)) ++col;
			if (w->bodies[c->bodyA].type != 0) used[c->bodyA] |= 1ull << col;
			if (w->bodies[c->bodyB].type != 0) used[c->bodyB] |= 1ull << col;
			key[i] = (float)col;
			if (col > maxc) maxc = col;
		}
		fprintf(stderr, "[ex] mode %d: %d colours\n", mode, maxc + 1);
		free(used);
		free(pri);
		exkey = key;
		for (int i = 0; i < n; ++i) idx[i] = i;
	}
	else if (mode == 4 || mode == 5)
	{
		/* DAG levels of the reference order (mode 4) or of the bottom-up order (mode 5), wrapped mod D */
		if (mode == 5)
		{
			for (int i = 0; i < n; ++i) key[i] = ex_y(w, ic[i]) + 0.37f * ex_x(w, ic[i]);
			qsort(idx, n, sizeof(int), ex_cmp);
		}
		int* last = (int*)calloc(w->nBodies, sizeof(int));
		int maxl = 0;
		int* lev = (int*)malloc(sizeof(int) * n);
		for (int k = 0; k < n; ++k)
		{
			int i = idx[k];
			const contact_t* c = &w->contacts[ic[i]];
			int l = 0;
			if (w->bodies[c->bodyA].type != 0 && last[c->bodyA] > l) l = last[c->bodyA];
			if (w->bodies[c->bodyB].type != 0 && last[c->bodyB] > l) l = last[c->bodyB];
			l += 1;
			if (w->bodies[c->bodyA].type != 0) last[c->bodyA] = l;
			if (w->bodies[c->bodyB].type != 0) last[c->bodyB] = l;
			lev[i] = l;
			if (l > maxl) maxl = l;
		}
		fprintf(stderr, "[ex] mode %d: %d levels, D %d\n", mode, maxl, D);
		/* key = (level mod D), ties in the base order */
		int* rank = (int*)malloc(sizeof(int) * n);
		for (int k = 0; k < n; ++k) rank[idx[k]] = k;
		for (int i = 0; i < n; ++i) key[i] = (float)((lev[i] - 1) % D) + (float)rank[i] / (float)(n + 1);
		free(rank);
		free(lev);
		free(last);
		for (int i = 0; i < n; ++i) idx[i] = i;
	}
	else if (mode == 7)
	{
		/* the reference's order INSIDE a block of a spatial partition, a coloured order only on the cut
		 * constraints. Blocks = square cells of D metres (a body's block: the cell of its centre); a constraint is interior if
		 * its non-static bodies share a block. Interior constraints of different blocks share no body, so "every block in the
		 * reference's order" is the reference's order restricted to the interior constraints; the cut constraints follow, greedy
		 * colours by hashed priority (the device's), colour classes ascending. */
		const float G = (float)D;
		float* pri = (float*)malloc(sizeof(float) * n);
		int* cut = (int*)malloc(sizeof(int) * n);
		int nCut = 0;
		for (int i = 0; i < n; ++i)
		{
			const contact_t* c = &w->contacts[ic[i]];
			const body_t* a = &w->bodies[c->bodyA];
			const body_t* b = &w->bodies[c->bodyB];
			cut[i] = 0;
			if (a->type != 0 && b->type != 0)
			{
				const int ax = (int)floorf(a->c.x / G), ay = (int)floorf(a->c.y / G), bx = (int)floorf(b->c.x / G), by = (int)floorf(b->c.y / G);
				cut[i] = ax != bx || ay != by;
			}
			nCut += cut[i];
			pri[i] = (float)(((uint32_t)ic[i] * 2654435761u) >> 8);
		}
		exkey = pri;
		qsort(idx, n, sizeof(int), ex_cmp);
		uint64_t* used = (uint64_t*)calloc(w->nBodies, sizeof(uint64_t));
		int maxc = 0;
		for (int k = 0; k < n; ++k)
		{
			const int i = idx[k];
			if (!cut[i]) continue;
			const contact_t* c = &w->contacts[ic[i]];
			const uint64_t m = used[c->bodyA] | used[c->bodyB];
			int col = 0;
			while (m & (1ull << col)) ++col;
			used[c->bodyA] |= 1ull << col;
			used[c->bodyB] |= 1ull << col;
			key[i] = 1.0f + (float)col;
			if (col > maxc) maxc = col;
		}
		for (int i = 0; i < n; ++i) if (!cut[i]) key[i] = (float)i / (float)(n + 1);
		fprintf(stderr, "[ex] mode 7: blocks of %d m, %d of %d constraints cut, %d cut colours\n", D, nCut, n, maxc + 1);
		free(used); free(pri); free(cut);
		exkey = key;
		for (int i = 0; i < n; ++i) idx[i] = i;
	}
	qsort(idx, n, sizeof(int), ex_cmp);
	for (int k = 0; k < n; ++k) out[k] = ic[idx[k]];
	memcpy(ic, out, sizeof(int) * n);
	free(idx); free(key); free(out);
}
#endif

static void solve_island(b2o_world* w, int* islandBodies, int bodyCount, int* islandContacts, int contactCount,
	int* islandJoints, int jointCount, float h, float dtRatio, int velIters, int posIters)
{
	pos_t* positions = (pos_t*)malloc(sizeof(pos_t) * (size_t)(bodyCount + 1));
	vel_t* velocities = (vel_t*)malloc(sizeof(vel_t) * (size_t)(bodyCount + 1));
	constraint_t* cs = (constraint_t*)malloc(sizeof(constraint_t) * (size_t)(contactCount + 1));
	for (int i = 0; i < bodyCount; ++i)
	{
		body_t* b = &w->bodies[islandBodies[i]];
		b->islandIndex = i;
		vec2 c = b->c;
		float a = b->a;
		vec2 v = b->v;
		float ww = b->w;
		if (b->type != 0)
		{
			b->c0 = b->c;
			b->a0 = b->a;
		}
		if (b->type == 2)
		{
			vec2 acc = v_add(v_scale(b->gravityScale, w->gravity), v_scale(b->invMass, b->force));
			v = v_add(v, v_scale(h, acc));
			ww += h * b->invI * b->torque;
			v = v_scale(1.0f / (1.0f + h * b->linearDamping), v);
			ww *= 1.0f / (1.0f + h * b->angularDamping);
		}
		positions[i].c = c; positions[i].a = a;
		velocities[i].v = v; velocities[i].w = ww;
	}
#ifdef B2O_ORDER_EXPERIMENT
	{
		const char* ex = getenv("B2O_ORDER");
		const char* exd = getenv("B2O_ORDER_D");
		if (ex && atoi(ex) > 0 && contactCount > 1000) ex_reorder(w, islandContacts, contactCount, atoi(ex), exd ? atoi(exd) : 64);
	}
#endif
	for (int i = 0; i < contactCount; ++i) init_constraint(w, &cs[i], islandContacts[i], positions, velocities, dtRatio);
	if (w->warmStarting)
	{
		for (int i = 0; i < contactCount; ++i) warm_start(&cs[i], velocities);
	}
	for (int i = 0; i < jointCount; ++i)
	{
		revolute_t* j = &w->joints[islandJoints[i]];
		body_t* bA = &w->bodies[j->bodyA];
		body_t* bB = &w->bodies[j->bodyB];
		int ia = bA->islandIndex, ib = bB->islandIndex;
		vec2* vA = &velocities[ia].v; float* wA = &velocities[ia].w;
		vec2* vB = &velocities[ib].v; float* wB = &velocities[ib].w;
		if (j->type == B2O_JOINT_GEAR)
		{
			int idx[4];
			gear_bodies g;
			gear_gather(w, j, positions, velocities, idx, &g);
			const int ids[4] = { j->bodyA, j->bodyB, j->bodyC, j->bodyD };
			float im[4], ii[4];
			vec2 lc[4];
			for (int k = 0; k < 4; ++k) { im[k] = w->bodies[ids[k]].invMass; ii[k] = w->bodies[ids[k]].invI; lc[k] = w->bodies[ids[k]].localCenter; }
			b2o_gear_init(j, &g, im, ii, lc, w->warmStarting);
			for (int k = 0; k < 4; ++k) { velocities[idx[k]].v = g.v[k]; velocities[idx[k]].w = g.w[k]; }
			continue;
		}
		switch (j->type)
		{
		case B2O_JOINT_DISTANCE:
			b2o_distance_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].c, positions[ia].a, vA, wA, positions[ib].c, positions[ib].a, vB, wB, w->warmStarting, dtRatio, h);
			break;
		case B2O_JOINT_PRISMATIC:
			b2o_prismatic_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].c, positions[ia].a, vA, wA, positions[ib].c, positions[ib].a, vB, wB, w->warmStarting, dtRatio);
			break;
		case B2O_JOINT_WELD:
			b2o_weld_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].a, vA, wA, positions[ib].a, vB, wB, w->warmStarting, dtRatio, h);
			break;
		case B2O_JOINT_WHEEL:
			b2o_wheel_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].c, positions[ia].a, vA, wA, positions[ib].c, positions[ib].a, vB, wB, w->warmStarting, dtRatio, h);
			break;
		case B2O_JOINT_ROPE:
			b2o_rope_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].c, positions[ia].a, vA, wA, positions[ib].c, positions[ib].a, vB, wB, w->warmStarting, dtRatio);
			break;
		case B2O_JOINT_FRICTION:
			b2o_friction_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].a, vA, wA, positions[ib].a, vB, wB, w->warmStarting, dtRatio);
			break;
		case B2O_JOINT_MOUSE:
			b2o_mouse_init(j, bB->mass, bB->invMass, bB->invI, bB->localCenter, positions[ib].c, positions[ib].a, vB, wB,
				w->warmStarting, dtRatio, h);
			break;
		case B2O_JOINT_PULLEY:
			b2o_pulley_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].c, positions[ia].a, vA, wA, positions[ib].c, positions[ib].a, vB, wB, w->warmStarting, dtRatio);
			break;
		case B2O_JOINT_MOTOR:
			b2o_motor_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].c, positions[ia].a, vA, wA, positions[ib].c, positions[ib].a, vB, wB, w->warmStarting, dtRatio);
			break;
		default:
			b2o_revolute_init(j, bA->invMass, bA->invI, bA->localCenter, bB->invMass, bB->invI, bB->localCenter,
				positions[ia].a, vA, wA, positions[ib].a, vB, wB, w->warmStarting, dtRatio);
		}
	}
	for (int it = 0; it < velIters; ++it)
	{
		for (int i = 0; i < jointCount; ++i)
		{
			revolute_t* j = &w->joints[islandJoints[i]];
			int ia = w->bodies[j->bodyA].islandIndex, ib = w->bodies[j->bodyB].islandIndex;
			vec2* vA = &velocities[ia].v; float* wA = &velocities[ia].w;
			vec2* vB = &velocities[ib].v; float* wB = &velocities[ib].w;
			if (j->type == B2O_JOINT_GEAR)
			{
				int idx[4];
				gear_bodies g;
				gear_gather(w, j, positions, velocities, idx, &g);
				b2o_gear_velocity(j, &g);
				for (int k = 0; k < 4; ++k) { velocities[idx[k]].v = g.v[k]; velocities[idx[k]].w = g.w[k]; }
				continue;
			}
			switch (j->type)
			{
			case B2O_JOINT_DISTANCE: b2o_distance_velocity(j, vA, wA, vB, wB); break;
			case B2O_JOINT_PRISMATIC: b2o_prismatic_velocity(j, vA, wA, vB, wB, h); break;
			case B2O_JOINT_WELD: b2o_weld_velocity(j, vA, wA, vB, wB); break;
			case B2O_JOINT_WHEEL: b2o_wheel_velocity(j, vA, wA, vB, wB, h); break;
			case B2O_JOINT_ROPE: b2o_rope_velocity(j, vA, wA, vB, wB, 1.0f / h); break;
			case B2O_JOINT_PULLEY: b2o_pulley_velocity(j, vA, wA, vB, wB); break;
			case B2O_JOINT_MOUSE: b2o_mouse_velocity(j, vB, wB, h); break;
			case B2O_JOINT_FRICTION: b2o_friction_velocity(j, vA, wA, vB, wB, h); break;
			case B2O_JOINT_MOTOR: b2o_motor_velocity(j, vA, wA, vB, wB, h, 1.0f / h); break;
			default: b2o_revolute_velocity(j, vA, wA, vB, wB, h);
			}
		}
		for (int i = 0; i < contactCount; ++i) solve_velocity(&cs[i], velocities);
	}
	/* StoreImpulses :605-618 */
	for (int i = 0; i < contactCount; ++i)
	{
		manifold* m = &w->contacts[cs[i].contact].m;
		for (int j = 0; j < cs[i].pointCount; ++j)
		{
			m->ni[j] = cs[i].normalImpulse[j];
			m->ti[j] = cs[i].tangentImpulse[j];
		}
	}
	/* b2Island::Report (b2Island.cpp:532-570): the velocity constraints' impulses, one record per contact of the island */
	if (w->postSolveOn)
	{
		for (int i = 0; i < contactCount; ++i)
		{
			GROW(w->postSolve, w->capPostSolve, w->nPostSolve + 1, b2o_contact_impulse);
			GROW(w->postSolveSlot, w->capPostSolveSlot, w->nPostSolve + 1, int);
			b2o_contact_impulse* r = &w->postSolve[w->nPostSolve];
			const contact_t* c = &w->contacts[cs[i].contact];
			memset(r, 0, sizeof(*r));
			r->fixture_a = c->fixtureA;
			r->fixture_b = c->fixtureB;
			r->count = cs[i].pointCount;
			for (int j = 0; j < 2; ++j)
			{
				/* like the device's record: the manifold's stored impulses (points beyond the solver's count keep the warm-start value) */
				r->normal_impulses[j] = c->m.ni[j];
				r->tangent_impulses[j] = c->m.ti[j];
			}
			w->postSolveSlot[w->nPostSolve] = cs[i].contact;
			w->nPostSolve++;
		}
	}
	for (int i = 0; i < bodyCount; ++i)
	{
		vec2 c = positions[i].c, v = velocities[i].v;
		float a = positions[i].a, ww = velocities[i].w;
		vec2 translation = v_scale(h, v);
		if (v_dot(translation, translation) > B2O_MAX_TRANSLATION_SQ)
		{
			float ratio = B2O_MAX_TRANSLATION / v_length(translation);
			v = v_scale(ratio, v);
		}
		float rotation = h * ww;
		if (rotation * rotation > B2O_MAX_ROTATION_SQ)
		{
			float ratio = B2O_MAX_ROTATION / f_abs(rotation);
			ww *= ratio;
		}
		c = v_add(c, v_scale(h, v));
		a += h * ww;
		positions[i].c = c; positions[i].a = a;
		velocities[i].v = v; velocities[i].w = ww;
	}
	int positionSolved = 0;
	for (int it = 0; it < posIters; ++it)
	{
		float minSeparation = 0.0f;
		for (int i = 0; i < contactCount; ++i) minSeparation = solve_position(&cs[i], positions, minSeparation);
		int jointsOkay = 1;
		for (int i = 0; i < jointCount; ++i)
		{
			revolute_t* j = &w->joints[islandJoints[i]];
			int ia = w->bodies[j->bodyA].islandIndex, ib = w->bodies[j->bodyB].islandIndex;
			vec2* cA = &positions[ia].c; float* aA = &positions[ia].a;
			vec2* cB = &positions[ib].c; float* aB = &positions[ib].a;
			int ok;
			if (j->type == B2O_JOINT_GEAR)
			{
				int idx[4];
				gear_bodies g;
				gear_gather(w, j, positions, velocities, idx, &g);
				ok = b2o_gear_position(j, &g);
				for (int k = 0; k < 4; ++k) { positions[idx[k]].c = g.c[k]; positions[idx[k]].a = g.a[k]; }
				jointsOkay = jointsOkay && ok;
				continue;
			}
			switch (j->type)
			{
			case B2O_JOINT_DISTANCE: ok = b2o_distance_position(j, cA, aA, cB, aB); break;
			case B2O_JOINT_PRISMATIC: ok = b2o_prismatic_position(j, cA, aA, cB, aB); break;
			case B2O_JOINT_WELD: ok = b2o_weld_position(j, cA, aA, cB, aB); break;
			case B2O_JOINT_WHEEL: ok = b2o_wheel_position(j, cA, aA, cB, aB); break;
			case B2O_JOINT_ROPE: ok = b2o_rope_position(j, cA, aA, cB, aB); break;
			case B2O_JOINT_PULLEY: ok = b2o_pulley_position(j, cA, aA, cB, aB); break;
			case B2O_JOINT_FRICTION: case B2O_JOINT_MOTOR: case B2O_JOINT_MOUSE: ok = 1; break;
			default: ok = b2o_revolute_position(j, cA, aA, cB, aB);
			}
			jointsOkay = jointsOkay && ok;
		}
		if (minSeparation >= -3.0f * B2O_LINEAR_SLOP && jointsOkay)
		{
			positionSolved = 1;
			break;
		}
	}
	for (int i = 0; i < bodyCount; ++i)
	{
		body_t* b = &w->bodies[islandBodies[i]];
		if (b->type == 0) continue;
		b->c = positions[i].c;
		b->a = positions[i].a;
		b->v = velocities[i].v;
		b->w = velocities[i].w;
		b->xf.q = r_make(b->a);
		b->xf.p = v_sub(b->c, r_mul(b->xf.q, b->localCenter));
	}
	if (w->allowSleep)
	{
		float minSleepTime = B2O_MAXFLOAT;
		const float linTolSqr = B2O_LINEAR_SLEEP_TOL * B2O_LINEAR_SLEEP_TOL;
		const float angTolSqr = B2O_ANGULAR_SLEEP_TOL * B2O_ANGULAR_SLEEP_TOL;
		for (int i = 0; i < bodyCount; ++i)
		{
			body_t* b = &w->bodies[islandBodies[i]];
			if (b->type == 0) continue;
			if ((b->flags & BF_AUTOSLEEP) == 0 || b->w * b->w > angTolSqr || v_dot(b->v, b->v) > linTolSqr)
			{
				b->sleepTime = 0.0f;
				minSleepTime = 0.0f;
			}
			else
			{
				b->sleepTime += h;
				minSleepTime = f_min(minSleepTime, b->sleepTime);
			}
		}
		if (minSleepTime >= B2O_TIME_TO_SLEEP && positionSolved)
		{
			for (int i = 0; i < bodyCount; ++i)
			{
				body_t* b = &w->bodies[islandBodies[i]];
				if (b->type == 0) continue;
				/* SetAwake(false)  b2Body.h:704-712 */
				b->flags &= ~BF_AWAKE;
				b->sleepTime = 0.0f;
				b->v = v_make(0.0f, 0.0f);
				b->w = 0.0f;
				b->force = v_make(0.0f, 0.0f);
				b->torque = 0.0f;
			}
		}
	}
	free(positions);
	free(velocities);
	free(cs);
}

/* b2World::Solve  b2World.cpp:1166-1431 */
#define B2O_SHARD_BIG_BODIES 4096
static uint32_t uf_priority(int i) { return (uint32_t)i * 2654435761u; }
static int shard_hash_owner(int root, int count) { return (int)(((uint32_t)root * 2654435761u >> 12) % (uint32_t)count); }

typedef struct { int root; int* bodies; int nb; int* contacts; int nc; int* joints; int nj; } pending_island;
static int pending_cmp(const void* a, const void* b)
{
	const pending_island* p = (const pending_island*)a;
	const pending_island* q = (const pending_island*)b;
	return p->root < q->root ? -1 : (p->root > q->root ? 1 : 0);
}

static void mark_owned(b2o_world* w, const int* bodies, int nb, const int* contacts, int nc, const int* joints, int nj)
{
	for (int i = 0; i < nb; ++i) if (w->bodies[bodies[i]].type != 0) w->bodyOwned[bodies[i]] = 1;
	for (int i = 0; i < nc; ++i) w->contactOwned[contacts[i]] = 1;
	for (int i = 0; i < nj; ++i) w->jointOwned[joints[i]] = 1;
}

static void solve_islands(b2o_world* w, float h, float dtRatio, int velIters, int posIters)
{
	int nb = w->nBodies;
	const int sharded = w->shardCount > 1 && !w->spatial;
	pending_island* pending = NULL;
	int nPending = 0, capPending = 0;
	GROW(w->bodyOwned, w->capBodyOwned, w->nBodies + 1, unsigned char);
	GROW(w->contactOwned, w->capContactOwned, w->nContactSlots + 1, unsigned char);
	GROW(w->jointOwned, w->capJointOwned, w->nJoints + 1, unsigned char);
	memset(w->bodyOwned, 0, (size_t)w->nBodies + 1);
	memset(w->contactOwned, 0, (size_t)w->nContactSlots + 1);
	memset(w->jointOwned, 0, (size_t)w->nJoints + 1);
	memset(w->shardBodies, 0, sizeof(w->shardBodies));
	memset(w->shardContacts, 0, sizeof(w->shardContacts));
	memset(w->shardJoints, 0, sizeof(w->shardJoints));
	int* islandBodies = (int*)malloc(sizeof(int) * (size_t)(nb + w->liveContacts + 2));
	int* islandContacts = (int*)malloc(sizeof(int) * (size_t)(w->liveContacts + 1));
	int* stack = (int*)malloc(sizeof(int) * (size_t)(nb + 1));
	int* islandJoints = (int*)malloc(sizeof(int) * (size_t)(w->nJoints + 1));
	for (int i = 0; i < nb; ++i) w->bodies[i].label = -1;
	for (int i = 0; i < w->nJoints; ++i) w->joints[i].islandFlag = 0;
	for (int seedSlot = 0; seedSlot < w->nNonStatic; ++seedSlot)
	{
		/* seeds in m_nonStaticBodies order (b2World.cpp:1207-1221) */
		const int seedIdx = w->nonStatic[seedSlot];
		body_t* seed = &w->bodies[seedIdx];
		if (seed->type == 0) continue;
		if (seed->flags & BF_ISLAND) continue;
		if ((seed->flags & BF_AWAKE) == 0 || (seed->flags & BF_ACTIVE) == 0) continue;
		if (sp_foreign_body(w, seedIdx)) continue; /* (another rank's: its whole component is) */
		int bodyCount = 0, contactCount = 0, jointCount = 0, sp = 0;
		stack[sp++] = seedIdx;
		seed->flags |= BF_ISLAND;
		while (sp > 0)
		{
			int bi = stack[--sp];
			body_t* b = &w->bodies[bi];
			islandBodies[bodyCount++] = bi;
			if (b->type == 0) continue;
			b->flags |= BF_AWAKE;
			for (int e = b->contactHead; e >= 0; e = w->contacts[e >> 1].next[e & 1])
			{
				contact_t* c = &w->contacts[e >> 1];
				if (c->flags & CF_ISLAND) continue;
				if ((c->flags & CF_ENABLED) == 0 || (c->flags & CF_TOUCHING) == 0) continue;
				if (w->fixtures[c->fixtureA].isSensor || w->fixtures[c->fixtureB].isSensor) continue;
				islandContacts[contactCount++] = e >> 1;
				c->flags |= CF_ISLAND;
				int other = (e & 1) == 0 ? c->bodyB : c->bodyA;
				if (w->bodies[other].flags & BF_ISLAND) continue;
				stack[sp++] = other;
				w->bodies[other].flags |= BF_ISLAND;
			}
			/* joints (b2World.cpp:1292-1318) */
			for (int e = b->jointHead; e >= 0; )
			{
				revolute_t* j = &w->joints[e >> 1];
				int other = (e & 1) == 0 ? j->bodyB : j->bodyA;
				int next = (e & 1) == 0 ? j->nextA : j->nextB;
				e = next;
				if (j->islandFlag) continue;
				if ((w->bodies[other].flags & BF_ACTIVE) == 0) continue;
				islandJoints[jointCount++] = (int)(j - w->joints);
				j->islandFlag = 1;
				if (w->bodies[other].flags & BF_ISLAND) continue;
				stack[sp++] = other;
				w->bodies[other].flags |= BF_ISLAND;
			}
		}
		int label = 0x7fffffff;
		for (int j = 0; j < bodyCount; ++j)
		{
			body_t* b = &w->bodies[islandBodies[j]];
			if (b->type == 0) b->flags &= ~BF_ISLAND;
			else if (islandBodies[j] < label) label = islandBodies[j];
		}
		for (int j = 0; j < bodyCount; ++j)
		{
			body_t* b = &w->bodies[islandBodies[j]];
			if (b->type != 0) b->label = label;
		}
		if (sharded)
		{
			/* who solves this island? its root is the non-static member of lowest priority (the device's union-find root) */
			int root = -1, nonStatic = 0;
			for (int j = 0; j < bodyCount; ++j)
			{
				const int bi = islandBodies[j];
				if (w->bodies[bi].type == 0) continue;
				++nonStatic;
				if (root < 0 || uf_priority(bi) < uf_priority(root)) root = bi;
			}
			if (nonStatic > B2O_SHARD_BIG_BODIES)
			{
				/* big islands are dealt round robin in root-id order once all of them are known */
				GROW(pending, capPending, nPending + 1, pending_island);
				pending_island* p = &pending[nPending++];
				p->root = root;
				p->nb = bodyCount; p->nc = contactCount; p->nj = jointCount;
				p->bodies = (int*)malloc(sizeof(int) * (size_t)(bodyCount + 1));
				p->contacts = (int*)malloc(sizeof(int) * (size_t)(contactCount + 1));
				p->joints = (int*)malloc(sizeof(int) * (size_t)(jointCount + 1));
				memcpy(p->bodies, islandBodies, sizeof(int) * (size_t)bodyCount);
				memcpy(p->contacts, islandContacts, sizeof(int) * (size_t)contactCount);
				memcpy(p->joints, islandJoints, sizeof(int) * (size_t)jointCount);
				continue;
			}
			{
				const int owner = shard_hash_owner(root, w->shardCount);
				w->shardBodies[owner] += nonStatic;
				w->shardContacts[owner] += contactCount;
				w->shardJoints[owner] += jointCount;
				if (owner != w->shardRank) continue; /* another rank's: its bodies stay flagged */
			}
		}
		mark_owned(w, islandBodies, bodyCount, islandContacts, contactCount, islandJoints, jointCount);
		solve_island(w, islandBodies, bodyCount, islandContacts, contactCount, islandJoints, jointCount, h, dtRatio, velIters, posIters);
	}
	if (nPending > 0)
	{
		qsort(pending, (size_t)nPending, sizeof(pending_island), pending_cmp);
		for (int k = 0; k < nPending; ++k)
		{
			pending_island* p = &pending[k];
			{
				int nonStatic = 0;
				for (int j = 0; j < p->nb; ++j) nonStatic += w->bodies[p->bodies[j]].type != 0;
				w->shardBodies[k % w->shardCount] += nonStatic;
				w->shardContacts[k % w->shardCount] += p->nc;
				w->shardJoints[k % w->shardCount] += p->nj;
			}
			if (k % w->shardCount == w->shardRank)
			{
				mark_owned(w, p->bodies, p->nb, p->contacts, p->nc, p->joints, p->nj);
				solve_island(w, p->bodies, p->nb, p->contacts, p->nc, p->joints, p->nj, h, dtRatio, velIters, posIters);
			}
			free(p->bodies); free(p->contacts); free(p->joints);
		}
	}
	free(pending);
	free(islandBodies);
	free(islandContacts);
	free(stack);
	free(islandJoints);
}

/* the rest of b2World::Solve: SynchronizeFixtures, FindNewContacts, ClearPostSolve (b2World.cpp:1373-1465) */
static void clear_post_solve(b2o_world* w)
{
	for (int i = 0; i < w->nContactSlots; ++i) w->contacts[i].flags &= ~CF_ISLAND;
	for (int i = 0; i < w->nBodies; ++i) w->bodies[i].flags &= ~BF_ISLAND;
}

static void solve(b2o_world* w, float h, float dtRatio, int velIters, int posIters)
{
	solve_islands(w, h, dtRatio, velIters, posIters);
	synchronize_fixtures(w);
	find_new_contacts(w);
	clear_post_solve(w);
}

/* ---- continuous collision (TOI) ------------------------------------------------------------------ */
static sweep_t body_sweep(const body_t* b)
{
	sweep_t s;
	s.localCenter = b->localCenter;
	s.c0 = b->c0; s.c = b->c;
	s.a0 = b->a0; s.a = b->a;
	s.alpha0 = b->alpha0;
	return s;
}

static void body_set_sweep(body_t* b, const sweep_t* s)
{
	b->c0 = s->c0; b->c = s->c;
	b->a0 = s->a0; b->a = s->a;
	b->alpha0 = s->alpha0;
}

/* b2Body::SynchronizeTransform  b2Body.h:958-962 */
static void body_sync_transform(body_t* b)
{
	b->xf.q = r_make(b->a);
	b->xf.p = v_sub(b->c, r_mul(b->xf.q, b->localCenter));
}

/* b2Body::Advance  b2Body.h:964-972 */
static void body_advance(body_t* b, float alpha)
{
	sweep_t s = body_sweep(b);
	b2o_sweep_advance(&s, alpha);
	s.c = s.c0;
	s.a = s.a0;
	body_set_sweep(b, &s);
	body_sync_transform(b);
}

/* b2World::ComputeToi  b2World.cpp:362-444 */
static float compute_toi(b2o_world* w, contact_t* c)
{
	if (c->flags & CF_TOI) return c->toi;
	body_t* bA = &w->bodies[c->bodyA];
	body_t* bB = &w->bodies[c->bodyB];
	float alpha0 = bA->alpha0;
	if (bA->alpha0 < bB->alpha0)
	{
		alpha0 = bB->alpha0;
		sweep_t s = body_sweep(bA);
		b2o_sweep_advance(&s, alpha0);
		body_set_sweep(bA, &s);
	}
	else if (bB->alpha0 < bA->alpha0)
	{
		alpha0 = bA->alpha0;
		sweep_t s = body_sweep(bB);
		b2o_sweep_advance(&s, alpha0);
		body_set_sweep(bB, &s);
	}
	gjk_proxy pA, pB;
	b2o_proxy_set(&pA, &w->fixtures[c->fixtureA].shape);
	b2o_proxy_set(&pB, &w->fixtures[c->fixtureB].shape);
	sweep_t sA = body_sweep(bA), sB = body_sweep(bB);
	toi_output out;
	b2o_time_of_impact(&out, &pA, &sA, &pB, &sB, 1.0f);
	float alpha = 1.0f;
	if (out.state == TOI_TOUCHING) alpha = f_min(alpha0 + (1.0f - alpha0) * out.t, 1.0f);
	c->toi = alpha;
	c->flags |= CF_TOI;
	w->toiCalls++;
	return alpha;
}

/* b2World::FindMinToiContact (b2World.cpp:1525-1611). The multi-threaded first pass and the serial
 * later passes both reduce to: lexicographic minimum of (alpha, proxyLo, proxyHi) over the active,
 * enabled TOI candidates (b2Contact::ToiLessThan b2Contact.cpp:326-334), visiting the candidates in
 * contact-array order, which fixes the order of the sweep side effects inside ComputeToi. */
static int find_min_toi(b2o_world* w, float* alphaOut)
{
	int minContact = -1;
	float minAlpha = 1.0f;
	for (int i = 0; i < w->toiCount; ++i)
	{
		int slot = w->carray[i];
		contact_t* c = &w->contacts[slot];
		if (!body_active_for_contact(&w->bodies[c->bodyA]) && !body_active_for_contact(&w->bodies[c->bodyB])) continue;
		if ((c->flags & CF_ENABLED) == 0 || c->toiCount > B2O_MAX_SUB_STEPS) continue; /* IsMinToiCandidate */
		if (sp_foreign_contact(w, c)) continue;
		float alpha = compute_toi(w, c);
		int less;
		if (minContact < 0) less = 1;
		else if (alpha == minAlpha)
		{
			const contact_t* m = &w->contacts[minContact];
			less = c->proxyLo != m->proxyLo ? c->proxyLo < m->proxyLo : c->proxyHi < m->proxyHi;
		}
		else less = alpha < minAlpha;
		if (less)
		{
			minContact = slot;
			minAlpha = alpha;
		}
	}
	*alphaOut = minAlpha;
	return minContact;
}

/* b2Contact::Update(listener) - the single-threaded variant wakes both bodies when the touching state flips */
static int listener_on(const b2o_world* w) { return w->eventsOn || w->preSolveFn != NULL || w->preSolveBatchFn != NULL || w->postSolveOn; }

static b2o_toi_callback* toi_log_push(b2o_world* w)
{
	GROW(w->toiLog, w->capToiLog, w->nToiLog + 1, b2o_toi_callback);
	b2o_toi_callback* r = &w->toiLog[w->nToiLog++];
	memset(r, 0, sizeof(*r));
	return r;
}

/* b2Contact::Update(listener) in a TOI sub-step (b2World.cpp:866,946 -> b2Contact.cpp:253-297): the callbacks it makes,
 * as one log record */
static void contact_update_st(b2o_world* w, contact_t* c)
{
	int wasTouching = (c->flags & CF_TOUCHING) != 0;
	contact_update(w, c);
	if (listener_on(w))
	{
		const int touching = (c->flags & CF_TOUCHING) != 0;
		int kind = 0;
		if (w->eventsOn && !wasTouching && touching) kind |= 1;
		if (w->eventsOn && wasTouching && !touching) kind |= 2;
		if ((w->preSolveFn != NULL || w->preSolveBatchFn != NULL) && touching) kind |= 4;
		if (w->eventsOn) c->reported = touching;
		if (kind & 4)
		{
			/* PreSolve is called where the reference calls it (b2Contact.cpp:283-297, from the Update of the sub-step): what it
			 * does to the contact - SetEnabled(false), SetFriction / SetRestitution / SetTangentSpeed - acts on this sub-step
			 * (b2World.cpp:873,948: a disabled contact keeps the sweeps and stays out of the sub-step's island). The log keeps
			 * the calls that change nothing: begin / end / PostSolve. */
			b2o_pre_solve_record rec;
			int* rank = contact_ranks(w);
			rec.contact_index = rank[c - w->contacts];
			free(rank);
			rec.fixture_a = c->fixtureA;
			rec.fixture_b = c->fixtureB;
			rec.enabled = 1;
			fill_manifold(&rec.old_manifold, &c->oldm);
			fill_manifold(&rec.manifold, &c->m);
			rec.material.friction = c->friction;
			rec.material.restitution = c->restitution;
			rec.material.tangent_speed = c->tangentSpeed;
			if (w->preSolveBatchFn) w->preSolveBatchFn(w->preSolveUser, 1, &rec);
			else rec.enabled = w->preSolveFn(w->preSolveUser, rec.contact_index, rec.fixture_a, rec.fixture_b, &rec.old_manifold, &rec.manifold, &rec.material) != 0;
			if (!rec.enabled) c->flags &= ~CF_ENABLED;
			c->friction = rec.material.friction;
			c->restitution = rec.material.restitution;
			c->tangentSpeed = rec.material.tangent_speed;
			kind &= ~4;
		}
		if (kind)
		{
			b2o_toi_callback* r = toi_log_push(w);
			r->kind = kind;
			r->contact_index = (int32_t)(c - w->contacts);
			r->fixture_a = c->fixtureA;
			r->fixture_b = c->fixtureB;
			fill_manifold(&r->old_manifold, &c->oldm);
			fill_manifold(&r->manifold, &c->m);
			r->material.friction = c->friction;
			r->material.restitution = c->restitution;
			r->material.tangent_speed = c->tangentSpeed;
		}
	}
	int sensor = w->fixtures[c->fixtureA].isSensor || w->fixtures[c->fixtureB].isSensor;
	if (!sensor && ((c->flags & CF_TOUCHING) != 0) != wasTouching)
	{
		set_awake(&w->bodies[c->bodyA]);
		set_awake(&w->bodies[c->bodyB]);
	}
}

/* b2ContactSolver::SolveTOIPositionConstraints  b2ContactSolver.cpp:755-843, one constraint */
static float solve_position_toi(const constraint_t* cc, pos_t* pos, float minSeparation, int toiIndexA, int toiIndexB)
{
	float mA = 0.0f, iA = 0.0f, mB = 0.0f, iB = 0.0f;
	if (cc->indexA == toiIndexA || cc->indexA == toiIndexB) { mA = cc->invMassA; iA = cc->invIA; }
	if (cc->indexB == toiIndexA || cc->indexB == toiIndexB) { mB = cc->invMassB; iB = cc->invIB; }
	vec2 cA = pos[cc->indexA].c, cB = pos[cc->indexB].c;
	float aA = pos[cc->indexA].a, aB = pos[cc->indexB].a;
	for (int j = 0; j < cc->pcPointCount; ++j)
	{
		xform xfA, xfB;
		xfA.q = r_make(aA);
		xfB.q = r_make(aB);
		xfA.p = v_sub(cA, r_mul(xfA.q, cc->localCenterA));
		xfB.p = v_sub(cB, r_mul(xfB.q, cc->localCenterB));
		vec2 normal, point;
		float separation;
		if (cc->type == MANIFOLD_CIRCLES)
		{
			vec2 pointA = xf_mul(xfA, cc->localPoint);
			vec2 pointB = xf_mul(xfB, cc->localPoints[0]);
			normal = v_sub(pointB, pointA);
			v_normalize(&normal);
			point = v_scale(0.5f, v_add(pointA, pointB));
			separation = v_dot(v_sub(pointB, pointA), normal) - cc->radiusA - cc->radiusB;
		}
		else if (cc->type == MANIFOLD_FACE_A)
		{
			normal = r_mul(xfA.q, cc->localNormal);
			vec2 planePoint = xf_mul(xfA, cc->localPoint);
			vec2 clip = xf_mul(xfB, cc->localPoints[j]);
			separation = v_dot(v_sub(clip, planePoint), normal) - cc->radiusA - cc->radiusB;
			point = clip;
		}
		else
		{
			normal = r_mul(xfB.q, cc->localNormal);
			vec2 planePoint = xf_mul(xfB, cc->localPoint);
			vec2 clip = xf_mul(xfA, cc->localPoints[j]);
			separation = v_dot(v_sub(clip, planePoint), normal) - cc->radiusA - cc->radiusB;
			point = clip;
			normal = v_neg(normal);
		}
		vec2 rA = v_sub(point, cA), rB = v_sub(point, cB);
		minSeparation = f_min(minSeparation, separation);
		float C = f_clamp(B2O_TOI_BAUMGARTE * (separation + B2O_LINEAR_SLOP), -B2O_MAX_LINEAR_CORRECTION, 0.0f);
		float rnA = v_cross(rA, normal), rnB = v_cross(rB, normal);
		float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
		float impulse = K > 0.0f ? -C / K : 0.0f;
		vec2 P = v_scale(impulse, normal);
		cA = v_sub(cA, v_scale(mA, P));
		aA -= iA * v_cross(rA, P);
		cB = v_add(cB, v_scale(mB, P));
		aB += iB * v_cross(rB, P);
	}
	pos[cc->indexA].c = cA; pos[cc->indexA].a = aA;
	pos[cc->indexB].c = cB; pos[cc->indexB].a = aB;
	return minSeparation;
}

/* b2Island::SolveTOI  b2Island.cpp:398-530 */
static void solve_toi_island(b2o_world* w, const int* bodies, int bodyCount, const int* contacts, int contactCount,
	float h, int velIters, int toiIndexA, int toiIndexB)
{
	pos_t positions[B2O_TOI_BODY_CAP];
	vel_t velocities[B2O_TOI_BODY_CAP];
	constraint_t cs[B2O_TOI_CONTACT_CAP];
	for (int i = 0; i < bodyCount; ++i)
	{
		body_t* b = &w->bodies[bodies[i]];
		b->islandIndex = i;
		positions[i].c = b->c; positions[i].a = b->a;
		velocities[i].v = b->v; velocities[i].w = b->w;
	}
	/* no warm starting in TOI sub-steps (b2World.cpp:995) */
	int warm = w->warmStarting;
	w->warmStarting = 0;
	for (int i = 0; i < contactCount; ++i) init_constraint(w, &cs[i], contacts[i], positions, velocities, 1.0f);
	for (int it = 0; it < 20; ++it)
	{
		float minSeparation = 0.0f;
		for (int i = 0; i < contactCount; ++i) minSeparation = solve_position_toi(&cs[i], positions, minSeparation, toiIndexA, toiIndexB);
		if (minSeparation >= -1.5f * B2O_LINEAR_SLOP) break;
	}
	/* leap of faith to the new safe state */
	w->bodies[bodies[toiIndexA]].c0 = positions[toiIndexA].c;
	w->bodies[bodies[toiIndexA]].a0 = positions[toiIndexA].a;
	w->bodies[bodies[toiIndexB]].c0 = positions[toiIndexB].c;
	w->bodies[bodies[toiIndexB]].a0 = positions[toiIndexB].a;
	/* InitializeVelocityConstraints at the corrected positions */
	for (int i = 0; i < contactCount; ++i) init_constraint(w, &cs[i], contacts[i], positions, velocities, 1.0f);
	w->warmStarting = warm;
	for (int it = 0; it < velIters; ++it)
		for (int i = 0; i < contactCount; ++i) solve_velocity(&cs[i], velocities);
	/* impulses are not stored back; b2Island::Report (b2Island.cpp:527, 532-570) shows them to PostSolve */
	if (w->postSolveOn)
	{
		for (int i = 0; i < contactCount; ++i)
		{
			const contact_t* c = &w->contacts[contacts[i]];
			b2o_toi_callback* r = toi_log_push(w);
			r->kind = 8;
			r->contact_index = contacts[i];
			r->fixture_a = c->fixtureA;
			r->fixture_b = c->fixtureB;
			fill_manifold(&r->manifold, &c->m);
			r->manifold.point_count = cs[i].pointCount;
			for (int k = 0; k < 2; ++k)
			{
				r->manifold.normal_impulse[k] = k < cs[i].pointCount ? cs[i].normalImpulse[k] : 0.0f;
				r->manifold.tangent_impulse[k] = k < cs[i].pointCount ? cs[i].tangentImpulse[k] : 0.0f;
			}
			r->material.friction = c->friction;
			r->material.restitution = c->restitution;
			r->material.tangent_speed = c->tangentSpeed;
		}
	}
	for (int i = 0; i < bodyCount; ++i)
	{
		vec2 c = positions[i].c, v = velocities[i].v;
		float a = positions[i].a, ww = velocities[i].w;
		vec2 translation = v_scale(h, v);
		if (v_dot(translation, translation) > B2O_MAX_TRANSLATION_SQ)
		{
			float ratio = B2O_MAX_TRANSLATION / v_length(translation);
			v = v_scale(ratio, v);
		}
		float rotation = h * ww;
		if (rotation * rotation > B2O_MAX_ROTATION_SQ)
		{
			float ratio = B2O_MAX_ROTATION / f_abs(rotation);
			ww *= ratio;
		}
		c = v_add(c, v_scale(h, v));
		a += h * ww;
		body_t* b = &w->bodies[bodies[i]];
		b->c = c;
		b->a = a;
		b->v = v;
		b->w = ww;
		body_sync_transform(b);
	}
}

/* b2World::StepSolveTOI  b2World.cpp:851-1024 */
static void step_solve_toi(b2o_world* w, float dt, int velIters, int minSlot, float minAlpha)
{
	contact_t* mc = &w->contacts[minSlot];
	int ia = mc->bodyA, ib = mc->bodyB;
	body_t* bA = &w->bodies[ia];
	body_t* bB = &w->bodies[ib];
	sweep_t backup1 = body_sweep(bA), backup2 = body_sweep(bB);
	body_advance(bA, minAlpha);
	body_advance(bB, minAlpha);
	contact_update_st(w, mc);
	mc->flags &= ~CF_TOI;
	++mc->toiCount;
	if ((mc->flags & CF_ENABLED) == 0 || (mc->flags & CF_TOUCHING) == 0)
	{
		mc->flags &= ~CF_ENABLED;
		body_set_sweep(bA, &backup1);
		body_set_sweep(bB, &backup2);
		body_sync_transform(bA);
		body_sync_transform(bB);
		return;
	}
	set_awake(bA);
	set_awake(bB);
	int bodies[B2O_TOI_BODY_CAP], contacts[B2O_TOI_CONTACT_CAP];
	int bodyCount = 0, contactCount = 0;
	bodies[bodyCount++] = ia;
	bodies[bodyCount++] = ib;
	contacts[contactCount++] = minSlot;
	bA->flags |= BF_ISLAND;
	bB->flags |= BF_ISLAND;
	mc->flags |= CF_ISLAND;
	const int seeds[2] = { ia, ib };
	for (int k = 0; k < 2; ++k)
	{
		body_t* body = &w->bodies[seeds[k]];
		if (body->type != 2) continue;
		for (int e = body->contactHead; e >= 0; e = w->contacts[e >> 1].next[e & 1])
		{
			if (bodyCount == B2O_TOI_BODY_CAP) break;
			if (contactCount == B2O_TOI_CONTACT_CAP) break;
			contact_t* c = &w->contacts[e >> 1];
			if (c->flags & CF_ISLAND) continue;
			int oi = (e & 1) == 0 ? c->bodyB : c->bodyA;
			body_t* other = &w->bodies[oi];
			if (other->type == 2 && (body->flags & BF_BULLET) == 0 && (other->flags & BF_BULLET) == 0) continue;
			if (w->fixtures[c->fixtureA].isSensor || w->fixtures[c->fixtureB].isSensor) continue;
			sweep_t backup = body_sweep(other);
			if ((other->flags & BF_ISLAND) == 0) body_advance(other, minAlpha);
			contact_update_st(w, c);
			if ((c->flags & CF_ENABLED) == 0 || (c->flags & CF_TOUCHING) == 0)
			{
				body_set_sweep(other, &backup);
				body_sync_transform(other);
				continue;
			}
			c->flags |= CF_ISLAND;
			contacts[contactCount++] = e >> 1;
			if (other->flags & BF_ISLAND) continue;
			other->flags |= BF_ISLAND;
			if (other->type != 0) set_awake(other);
			bodies[bodyCount++] = oi;
		}
	}
	float subDt = (1.0f - minAlpha) * dt;
	solve_toi_island(w, bodies, bodyCount, contacts, contactCount, subDt, velIters, 0, 1);
	for (int i = 0; i < bodyCount; ++i)
	{
		body_t* body = &w->bodies[bodies[i]];
		body->flags &= ~BF_ISLAND;
		if (body->type != 2) continue;
		sync_body_fixtures(w, body);
		for (int e = body->contactHead; e >= 0; e = w->contacts[e >> 1].next[e & 1])
			w->contacts[e >> 1].flags &= ~(CF_ISLAND | CF_TOI);
	}
	find_new_contacts(w);
	w->toiEvents++;
}

/* b2World::SolveTOI (b2World.cpp:1026-1093) + ClearPostSolveTOI (:1467-1523) */
static void solve_toi(b2o_world* w, float dt, int velIters)
{
	/* (b2World.cpp:1045: flags were modified by a previous sub-step if the step is not complete) */
	int clearPost = w->stepComplete == 0, first = w->stepComplete;
	for (;;)
	{
		float minAlpha = 1.0f;
		int minSlot = find_min_toi(w, &minAlpha);
		if (first && minSlot >= 0) clearPost = 1;
		first = 0;
		if (minSlot < 0 || 1.0f - 10.0f * B2O_EPSILON < minAlpha)
		{
			w->stepComplete = 1;
			break;
		}
		step_solve_toi(w, dt, velIters, minSlot, minAlpha);
		/* b2World::SetSubStepping (b2World.cpp:1082-1086): one TOI event per Step call; the flags and sweeps stay as they are */
		if (w->subStepping)
		{
			w->stepComplete = 0;
			return;
		}
	}
	if (!clearPost) return;
	for (int i = 0; i < w->nContactSlots; ++i)
	{
		contact_t* c = &w->contacts[i];
		if (!c->alive) continue;
		c->flags &= ~(CF_TOI | CF_ISLAND);
		c->toiCount = 0;
		c->toi = 1.0f;
	}
	for (int i = 0; i < w->nBodies; ++i)
	{
		w->bodies[i].flags &= ~BF_ISLAND;
		w->bodies[i].alpha0 = 0.0f;
	}
}

/* b2World::Step  b2World.cpp:1613-1710 */
static int seq_cmp(const void* a, const void* b);

/* Events of the step: every contact whose touching state differs from what the host was last told, begins first, each
 * group by proxy-id pair (the order of the reference's deferred callbacks, b2ContactManager.cpp:420-438, :64-67). */
static void collect_contact_events(b2o_world* w)
{
	/* indices refer to b2o_get_contacts' creation order */
	const contact_t** live = (const contact_t**)malloc(sizeof(void*) * (size_t)(w->liveContacts + 1));
	int n = 0;
	for (int i = 0; i < w->nContactSlots; ++i) if (w->contacts[i].alive) live[n++] = &w->contacts[i];
	qsort(live, (size_t)n, sizeof(void*), seq_cmp);
	for (int i = 0; i < n; ++i)
	{
		contact_t* c = (contact_t*)live[i];
		const int touching = (c->flags & CF_TOUCHING) != 0;
		if (touching == c->reported) continue;
		c->reported = touching;
		push_event(w, c, touching ? 0 : 1, i);
	}
	free(live);
	/* insertion sort by (kind, key): the lists are short */
	for (int i = 1; i < w->nEvents; ++i)
	{
		b2o_contact_event e = w->events[i];
		uint64_t k = w->eventKeys[i];
		int j = i - 1;
		while (j >= 0 && (w->events[j].kind > e.kind || (w->events[j].kind == e.kind && w->eventKeys[j] > k)))
		{
			w->events[j + 1] = w->events[j];
			w->eventKeys[j + 1] = w->eventKeys[j];
			--j;
		}
		w->events[j + 1] = e;
		w->eventKeys[j + 1] = k;
	}
}

int b2o_get_toi_callbacks(const b2o_world* w, int cap, b2o_toi_callback* out)
{
	int* rank = contact_ranks(w);
	for (int i = 0; i < w->nToiLog && i < cap; ++i)
	{
		out[i] = w->toiLog[i];
		out[i].contact_index = rank[w->toiLog[i].contact_index]; /* slot -> index in b2o_get_contacts order */
	}
	free(rank);
	return w->nToiLog;
}

void b2o_enable_contact_events(b2o_world* w, int enable)
{
	w->eventsOn = enable != 0;
	w->nEvents = 0;
}

int b2o_get_contact_events(const b2o_world* w, int cap, b2o_contact_event* out)
{
	for (int i = 0; i < w->nEvents && i < cap; ++i) out[i] = w->events[i];
	return w->nEvents;
}

static int post_cmp(const void* a, const void* b)
{
	return destroy_cmp(a, b);
}

/* deferred PostSolve order (b2ContactManager.cpp:454-470): proxy-id pair; contact index as b2o_get_contacts shows it */
static void finish_post_solve(b2o_world* w)
{
	const int n = w->nPostSolve;
	if (n == 0) return;
	int* order = (int*)malloc(sizeof(int) * (size_t)n);
	int* slots = (int*)malloc(sizeof(int) * (size_t)n);
	for (int i = 0; i < n; ++i) slots[i] = w->postSolveSlot[i];
	/* sort record indices by the proxy ids of their contacts: sort the slots, then match (slots are unique) */
	destroy_cmp_world = w;
	qsort(slots, (size_t)n, sizeof(int), post_cmp);
	int* rank = contact_ranks(w);
	b2o_contact_impulse* sorted = (b2o_contact_impulse*)malloc(sizeof(b2o_contact_impulse) * (size_t)n);
	int* where = (int*)malloc(sizeof(int) * (size_t)(w->nContactSlots + 1));
	for (int i = 0; i < n; ++i) where[w->postSolveSlot[i]] = i;
	for (int k = 0; k < n; ++k)
	{
		sorted[k] = w->postSolve[where[slots[k]]];
		sorted[k].contact_index = rank[slots[k]];
	}
	memcpy(w->postSolve, sorted, sizeof(b2o_contact_impulse) * (size_t)n);
	free(where); free(sorted); free(rank); free(slots); free(order);
}

/* ---- the phases of a step as separate entry points (same sequence as b2o_step; the sharded driver exchanges the solved
 * islands between b2o_phase_solve and b2o_phase_sync_fixtures) ---------------------------------------------------------- */
static void sp_begin_step(b2o_world* w);
void b2o_step_begin(b2o_world* w, float dt, int velIters, int posIters)
{
	if (w->spatial) sp_begin_step(w);
	w->nEvents = 0;
	w->nPostSolve = 0;
	w->nToiLog = 0;
	w->stepDt = dt; w->stepVelIters = velIters; w->stepPosIters = posIters;
	if (w->newFixture)
	{
		find_new_contacts(w);
		w->newFixture = 0;
	}
}
void b2o_phase_collide(b2o_world* w)
{
	collide(w);
	if (w->eventsOn && w->continuous && w->stepDt > 0.0f) collect_contact_events(w);
}
void b2o_phase_solve(b2o_world* w)
{
	if (!w->stepComplete) return; /* b2World.cpp:1668 */
	if (w->stepDt > 0.0f) solve_islands(w, w->stepDt, w->inv_dt0 * w->stepDt, w->stepVelIters, w->stepPosIters);
}
void b2o_phase_sync_fixtures(b2o_world* w)
{
	if (w->stepComplete && w->stepDt > 0.0f)
	{
		synchronize_fixtures(w);
		if (w->spatial) sp_exchange_state(w, 0);
	}
}
void b2o_phase_find_new_contacts(b2o_world* w)
{
	if (w->stepComplete && w->stepDt > 0.0f)
	{
		find_new_contacts(w);
		clear_post_solve(w);
	}
}
void b2o_phase_solve_toi(b2o_world* w)
{
	if (w->continuous && w->stepDt > 0.0f)
	{
		const uint64_t seq0 = w->nextSeq;
		solve_toi(w, w->stepDt, w->stepVelIters);
		if (w->spatial)
		{
			/* (the merge of contacts created inside TOI sub-steps over the ranks is the product's: k_sp_merge_tails; this shim
			 * covers worlds whose events create none and says so otherwise) */
			if (w->nextSeq != seq0) w->spFailed = 1;
			sp_exchange_state(w, 1);
		}
	}
}
void b2o_step_end(b2o_world* w)
{
	if (w->stepDt > 0.0f) w->inv_dt0 = 1.0f / w->stepDt;
	for (int i = 0; i < w->nBodies; ++i)
	{
		w->bodies[i].force = v_make(0.0f, 0.0f);
		w->bodies[i].torque = 0.0f;
	}
	if (w->eventsOn) collect_contact_events(w);
	if (w->postSolveOn) finish_post_solve(w);
}

/* ---- island sharding: every rank packs the records of the islands it solved (each record carries its id) into its SLAB,
 * the slabs are all-gathered, every rank imports the others'. Layout and ownership as in b2d_kernels_shard.h; every rank
 * has counted every rank's slab during its (replicated) island build. ------------------------------------------------------ */
#define SHARD_BODY_WORDS 13
#define SHARD_CONTACT_WORDS 5
#define SHARD_JOINT_WORDS 6
static int32_t fbits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static float bitsf(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

void b2o_set_shard(b2o_world* w, int rank, int count) { w->shardRank = rank; w->shardCount = count; }

size_t b2o_shard_slab_words(const b2o_world* w, int rank)
{
	return (size_t)w->shardBodies[rank] * SHARD_BODY_WORDS + (size_t)w->shardContacts[rank] * SHARD_CONTACT_WORDS + (size_t)w->shardJoints[rank] * SHARD_JOINT_WORDS;
}

void b2o_shard_export(const b2o_world* w, int32_t* out)
{
	const int me = w->shardRank;
	int32_t* o = out;
	int n = 0;
	for (int i = 0; i < w->nBodies && w->bodyOwned; ++i)
	{
		if (!w->bodyOwned[i]) continue;
		const body_t* b = &w->bodies[i];
		o[0] = i;
		o[1] = fbits(b->c.x); o[2] = fbits(b->c.y); o[3] = fbits(b->a); o[4] = fbits(b->sleepTime);
		o[5] = fbits(b->v.x); o[6] = fbits(b->v.y); o[7] = fbits(b->w);
		o[8] = (b->flags & BF_AWAKE) ? 1 : 0;
		o[9] = fbits(b->xf.p.x); o[10] = fbits(b->xf.p.y); o[11] = fbits(b->xf.q.s); o[12] = fbits(b->xf.q.c);
		o += SHARD_BODY_WORDS;
		++n;
	}
	(void)n;
	o = out + (size_t)w->shardBodies[me] * SHARD_BODY_WORDS;
	int* rank = contact_ranks(w);
	for (int slot = 0; slot < w->nContactSlots && w->contactOwned; ++slot)
	{
		if (rank[slot] < 0 || !w->contactOwned[slot]) continue;
		const contact_t* c = &w->contacts[slot];
		o[0] = rank[slot]; /* the contact's index in b2o_get_contacts order = every rank's index for it */
		o[1] = fbits(c->m.ni[0]); o[2] = fbits(c->m.ti[0]); o[3] = fbits(c->m.ni[1]); o[4] = fbits(c->m.ti[1]);
		o += SHARD_CONTACT_WORDS;
	}
	free(rank);
	o = out + (size_t)w->shardBodies[me] * SHARD_BODY_WORDS + (size_t)w->shardContacts[me] * SHARD_CONTACT_WORDS;
	for (int j = 0; j < w->nJoints && w->jointOwned; ++j)
	{
		if (!w->jointOwned[j]) continue;
		const revolute_t* jn = &w->joints[j];
		o[0] = j;
		o[1] = fbits(jn->impulse[0]);
		o[2] = fbits(jn->type == B2O_JOINT_WHEEL ? jn->springImpulse : jn->impulse[1]);
		o[3] = fbits(jn->impulse[2]);
		o[4] = fbits(jn->motorImpulse);
		o[5] = jn->limitState;
		o += SHARD_JOINT_WORDS;
	}
}

void b2o_shard_import(b2o_world* w, const int32_t* in, size_t strideWords)
{
	int* slotOf = NULL; /* contact index (b2o_get_contacts order) -> slot */
	{
		int* rank = contact_ranks(w);
		slotOf = (int*)malloc(sizeof(int) * (size_t)(w->liveContacts + 1));
		for (int slot = 0; slot < w->nContactSlots; ++slot) if (rank[slot] >= 0) slotOf[rank[slot]] = slot;
		free(rank);
	}
	for (int r = 0; r < w->shardCount; ++r)
	{
		if (r == w->shardRank) continue;
		const int32_t* o = in + (size_t)r * strideWords;
		for (int k = 0; k < w->shardBodies[r]; ++k, o += SHARD_BODY_WORDS)
		{
			body_t* b = &w->bodies[o[0]];
			b->c0 = b->c; b->a0 = b->a; b->alpha0 = 0.0f;
			b->c = v_make(bitsf(o[1]), bitsf(o[2])); b->a = bitsf(o[3]); b->sleepTime = bitsf(o[4]);
			b->v = v_make(bitsf(o[5]), bitsf(o[6])); b->w = bitsf(o[7]);
			b->xf.p = v_make(bitsf(o[9]), bitsf(o[10])); b->xf.q.s = bitsf(o[11]); b->xf.q.c = bitsf(o[12]);
			if (o[8]) b->flags |= BF_AWAKE;
			else
			{
				b->flags &= ~BF_AWAKE;
				b->force = v_make(0.0f, 0.0f);
				b->torque = 0.0f;
			}
		}
		for (int k = 0; k < w->shardContacts[r]; ++k, o += SHARD_CONTACT_WORDS)
		{
			contact_t* c = &w->contacts[slotOf[o[0]]];
			c->m.ni[0] = bitsf(o[1]); c->m.ti[0] = bitsf(o[2]); c->m.ni[1] = bitsf(o[3]); c->m.ti[1] = bitsf(o[4]);
		}
		for (int k = 0; k < w->shardJoints[r]; ++k, o += SHARD_JOINT_WORDS)
		{
			revolute_t* jn = &w->joints[o[0]];
			jn->impulse[0] = bitsf(o[1]);
			if (jn->type == B2O_JOINT_WHEEL) jn->springImpulse = bitsf(o[2]); else jn->impulse[1] = bitsf(o[2]);
			jn->impulse[2] = bitsf(o[3]);
			jn->motorImpulse = bitsf(o[4]);
			jn->limitState = o[5];
		}
	}
	free(slotOf);
}

void b2o_step(b2o_world* w, float dt, int velIters, int posIters)
{
	if (w->spatial)
	{
		/* (the exchanges of a spatially sharded world hang on the phase entry points) */
		b2o_step_begin(w, dt, velIters, posIters);
		b2o_phase_collide(w);
		b2o_phase_solve(w);
		b2o_phase_sync_fixtures(w);
		b2o_phase_find_new_contacts(w);
		b2o_phase_solve_toi(w);
		b2o_step_end(w);
		return;
	}
	w->nEvents = 0;
	w->nPostSolve = 0;
	if (w->newFixture)
	{
		find_new_contacts(w);
		w->newFixture = 0;
	}
	w->nToiLog = 0;
	collide(w);
	if (w->eventsOn && w->continuous && dt > 0.0f) collect_contact_events(w);
	float inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
	float dtRatio = w->inv_dt0 * dt;
	if (w->stepComplete && dt > 0.0f) solve(w, dt, dtRatio, velIters, posIters); /* b2World.cpp:1668 */
	if (w->continuous && dt > 0.0f) solve_toi(w, dt, velIters);
	if (dt > 0.0f) w->inv_dt0 = inv_dt;
	for (int i = 0; i < w->nBodies; ++i)
	{
		w->bodies[i].force = v_make(0.0f, 0.0f);
		w->bodies[i].torque = 0.0f;
	}
	if (w->eventsOn) collect_contact_events(w);
	if (w->postSolveOn) finish_post_solve(w);
}

void b2o_set_contact_filter(b2o_world* w, b2o_should_collide_fn fn, void* user) { w->filterFn = fn; w->filterUser = user; }
void b2o_set_pre_solve(b2o_world* w, b2o_pre_solve_fn fn, void* user) { w->preSolveFn = fn; w->preSolveUser = user; }
void b2o_set_pre_solve_batch(b2o_world* w, b2o_pre_solve_batch_fn fn, void* user) { w->preSolveBatchFn = fn; if (fn) w->preSolveUser = user; }
void b2o_enable_post_solve(b2o_world* w, int enable) { w->postSolveOn = enable != 0; w->nPostSolve = 0; }
int b2o_get_post_solve(const b2o_world* w, int cap, b2o_contact_impulse* out)
{
	for (int i = 0; i < w->nPostSolve && i < cap; ++i) out[i] = w->postSolve[i];
	return w->nPostSolve;
}
int b2o_default_should_collide(const b2o_world* w, int fixtureA, int fixtureB)
{
	return filter_should_collide(&w->fixtures[fixtureA], &w->fixtures[fixtureB]);
}

/* b2Body::ApplyForceToCenter + ApplyTorque  b2Body.h:740-775 */
void b2o_apply_force(b2o_world* w, int body, float fx, float fy, float torque, int wake)
{
	body_t* b = &w->bodies[body];
	if (b->type != 2) return;
	if (wake && (b->flags & BF_AWAKE) == 0) set_awake(b);
	if (b->flags & BF_AWAKE)
	{
		b->force = v_add(b->force, v_make(fx, fy));
		b->torque += torque;
	}
}

/* b2Body::SetLinearVelocity / SetAngularVelocity  b2Body.h:575-611 */
void b2o_set_velocity(b2o_world* w, int body, float vx, float vy, float omega)
{
	body_t* b = &w->bodies[body];
	if (b->type == 0) return;
	if (vx * vx + vy * vy > 0.0f) set_awake(b);
	b->v = v_make(vx, vy);
	if (omega * omega > 0.0f) set_awake(b);
	b->w = omega;
}

int b2o_body_count(const b2o_world* w) { return w->nBodies; }

void b2o_get_body_states(const b2o_world* w, float* out)
{
	for (int i = 0; i < w->nBodies; ++i)
	{
		const body_t* b = &w->bodies[i];
		float* o = out + 10 * i;
		o[0] = b->xf.p.x; o[1] = b->xf.p.y; o[2] = b->a;
		o[3] = b->v.x; o[4] = b->v.y; o[5] = b->w;
		o[6] = b->c.x; o[7] = b->c.y;
		uint32_t f = (b->flags & 0x7cu) | (uint32_t)b->type;
		memcpy(o + 8, &f, 4);
		o[9] = b->sleepTime;
	}
}

void b2o_get_mass(const b2o_world* w, int body, float* mass, float* inertia, float* lcx, float* lcy)
{
	const body_t* b = &w->bodies[body];
	*mass = b->mass;
	*inertia = b->I + b->mass * v_dot(b->localCenter, b->localCenter);
	*lcx = b->localCenter.x;
	*lcy = b->localCenter.y;
}

int b2o_contact_count(const b2o_world* w) { return w->liveContacts; }

static int seq_cmp(const void* a, const void* b)
{
	const contact_t* p = *(const contact_t* const*)a;
	const contact_t* q = *(const contact_t* const*)b;
	return p->seq < q->seq ? -1 : (p->seq > q->seq ? 1 : 0);
}

/* contacts in creation order (oldest first), like the device array */
int b2o_get_contacts(const b2o_world* w, int cap, b2o_contact* out)
{
	const contact_t** live = (const contact_t**)malloc(sizeof(void*) * (size_t)(w->liveContacts + 1));
	int n = 0;
	for (int i = 0; i < w->nContactSlots; ++i) if (w->contacts[i].alive) live[n++] = &w->contacts[i];
	qsort(live, (size_t)n, sizeof(void*), seq_cmp);
	if (n > cap) n = cap;
	for (int i = 0; i < n; ++i)
	{
		const contact_t* c = live[i];
		b2o_contact* o = &out[i];
		memset(o, 0, sizeof(*o));
		o->fixture_a = c->fixtureA; o->fixture_b = c->fixtureB;
		o->body_a = c->bodyA; o->body_b = c->bodyB;
		o->flags = ((c->flags & CF_TOUCHING) ? 1u : 0u) | ((c->flags & CF_ENABLED) ? 2u : 0u);
		o->manifold_type = c->m.type;
		o->point_count = c->m.pointCount;
		o->local_normal[0] = c->m.localNormal.x; o->local_normal[1] = c->m.localNormal.y;
		o->local_point[0] = c->m.localPoint.x; o->local_point[1] = c->m.localPoint.y;
		for (int k = 0; k < 2; ++k)
		{
			o->point_local[k][0] = c->m.p[k].x; o->point_local[k][1] = c->m.p[k].y;
			o->normal_impulse[k] = c->m.ni[k]; o->tangent_impulse[k] = c->m.ti[k];
			o->id_key[k] = c->m.id[k];
		}
		o->friction = c->friction;
		o->restitution = c->restitution;
		o->tangent_speed = c->tangentSpeed;
	}
	free(live);
	return n;
}

void b2o_get_island_labels(const b2o_world* w, int32_t* out)
{
	for (int i = 0; i < w->nBodies; ++i) out[i] = w->bodies[i].label;
}

void b2o_get_fat_aabb(const b2o_world* w, int fixture, float out4[4])
{
	memcpy(out4, w->fixtures[fixture].fat, 16);
}

void b2o_get_toi_stats(const b2o_world* w, int32_t out[2])
{
	out[0] = w->toiEvents;
	out[1] = w->toiCalls;
}


/* ==== life cycle and mutators between steps (same semantics as include/b2hip.h; reference lines cited there) ============ */

/* b2ContactManager::RecalculateToiCandidacy(b2Contact*) (b2ContactManager.cpp:590-640) */
static void recalc_toi_candidacy(b2o_world* w, int slot)
{
	contact_t* c = &w->contacts[slot];
	int cand = is_toi_candidate(w, &w->fixtures[c->fixtureA], &w->fixtures[c->fixtureB]);
	if (cand == ((c->flags & CF_TOI_CANDIDATE) != 0)) return;
	c->flags = (c->flags ^ CF_TOI_CANDIDATE) & ~CF_TOI;
	c->toiCount = 0;
	c->toi = 1.0f;
	if (c->managerIndex < 0) return;
	if (cand)
	{
		/* swap with the first non-candidate: the new candidate takes slot toiCount */
		int other = w->carray[w->toiCount];
		w->contacts[other].managerIndex = c->managerIndex;
		w->carray[c->managerIndex] = other;
		w->carray[w->toiCount] = slot;
		c->managerIndex = w->toiCount;
		w->toiCount++;
	}
	else
	{
		w->toiCount--;
		int other = w->carray[w->toiCount];
		w->contacts[other].managerIndex = c->managerIndex;
		w->carray[c->managerIndex] = other;
		w->carray[w->toiCount] = slot;
		c->managerIndex = w->toiCount;
	}
}

static void recalc_body(b2o_world* w, int body, int fixture /* -1: every contact of the body */)
{
	for (int e = w->bodies[body].contactHead; e >= 0; )
	{
		contact_t* c = &w->contacts[e >> 1];
		int next = c->next[e & 1];
		if (fixture < 0 || c->fixtureA == fixture || c->fixtureB == fixture) recalc_toi_candidacy(w, e >> 1);
		e = next;
	}
}

static void unbuffer_move(b2o_world* w, int fixture)
{
	int n = 0;
	for (int k = 0; k < w->nMoves; ++k)
		if (w->moves[k] != fixture) w->moves[n++] = w->moves[k];
	w->nMoves = n;
}

static void buffer_move(b2o_world* w, int fixture)
{
	GROW(w->moves, w->capMoves, w->nMoves + 1, int);
	w->moves[w->nMoves++] = fixture;
}

/* b2Fixture::DestroyProxies (b2Fixture.cpp:143-157) */
static void release_proxy(b2o_world* w, int fixture)
{
	fixture_t* f = &w->fixtures[fixture];
	if (f->proxyId < 0) return;
	free_proxy_id(w, f->proxyId);
	f->proxyId = -1;
	unbuffer_move(w, fixture);
	f->fat[0] = f->fat[1] = 1e30f; /* overlaps nothing */
	f->fat[2] = f->fat[3] = -1e30f;
}

/* ... + the bookkeeping of a fixture that is gone */
static void drop_fixture(b2o_world* w, int fixture)
{
	release_proxy(w, fixture);
	w->fixtures[fixture].dead = 1;
}

void b2o_destroy_fixture(b2o_world* w, int fixture)
{
	fixture_t* f = &w->fixtures[fixture];
	if (f->dead) return;
	body_t* b = &w->bodies[f->body];
	/* unlink from the body's fixture list */
	int* link = &b->fixtureHead;
	while (*link >= 0 && *link != fixture) link = &w->fixtures[*link].nextInBody;
	if (*link == fixture) *link = f->nextInBody;
	/* its contacts, in the body's contact-list order (b2Body.cpp:262-275) */
	for (int e = b->contactHead; e >= 0; )
	{
		contact_t* c = &w->contacts[e >> 1];
		int next = c->next[e & 1];
		if (c->fixtureA == fixture || c->fixtureB == fixture) destroy_contact(w, e >> 1);
		e = next;
	}
	drop_fixture(w, fixture);
	reset_mass(w, b);
}

void b2o_destroy_body(b2o_world* w, int body)
{
	body_t* b = &w->bodies[body];
	if (b->dead) return;
	/* joints, newest first; a gear joint goes with a body of the joints it couples */
	for (int j = w->nJoints - 1; j >= 0; --j)
	{
		revolute_t* jt = &w->joints[j];
		if (jt->type < 0) continue;
		int touches = jt->bodyA == body || jt->bodyB == body;
		if (jt->type == B2O_JOINT_GEAR) touches = touches || jt->bodyC == body || jt->bodyD == body;
		if (touches) b2o_destroy_joint(w, j);
	}
	for (int e = b->contactHead; e >= 0; )
	{
		int next = w->contacts[e >> 1].next[e & 1];
		destroy_contact(w, e >> 1);
		e = next;
	}
	for (int f = b->fixtureHead; f >= 0; )
	{
		int next = w->fixtures[f].nextInBody;
		drop_fixture(w, f);
		f = next;
	}
	b->fixtureHead = -1;
	if (b->worldIndex >= 0)
	{
		/* b2RemoveAndSwapBack on m_nonStaticBodies (b2World.cpp:662-667) */
		int slot = b->worldIndex, last = w->nonStatic[w->nNonStatic - 1];
		w->nonStatic[slot] = last;
		w->bodies[last].worldIndex = slot;
		w->nNonStatic--;
		b->worldIndex = -1;
	}
	b->dead = 1;
	b->type = 0;
	b->flags &= ~(BF_ACTIVE | BF_AWAKE | BF_BULLET);
	b->v = v_make(0.0f, 0.0f);
	b->w = 0.0f;
	b->force = v_make(0.0f, 0.0f);
	b->torque = 0.0f;
	b->invMass = b->invI = 0.0f;
}

/* b2Body::SetTransform (b2Body.cpp:451-473) */
void b2o_set_transform(b2o_world* w, int body, float x, float y, float angle)
{
	body_t* b = &w->bodies[body];
	b->xf.q = r_make(angle);
	b->xf.p = v_make(x, y);
	b->c = xf_mul(b->xf, b->localCenter);
	b->a = angle;
	b->c0 = b->c;
	b->a0 = angle;
	for (int f = b->fixtureHead; f >= 0; f = w->fixtures[f].nextInBody)
	{
		fixture_t* fx = &w->fixtures[f];
		if (fx->proxyId < 0) continue;
		float aabb[4];
		shape_aabb(&fx->shape, b->xf, aabb);
		if (fx->fat[0] <= aabb[0] && fx->fat[1] <= aabb[1] && aabb[2] <= fx->fat[2] && aabb[3] <= fx->fat[3]) continue;
		fx->fat[0] = aabb[0] - B2O_AABB_EXTENSION;
		fx->fat[1] = aabb[1] - B2O_AABB_EXTENSION;
		fx->fat[2] = aabb[2] + B2O_AABB_EXTENSION;
		fx->fat[3] = aabb[3] + B2O_AABB_EXTENSION;
		buffer_move(w, f);
	}
}

/* b2Body::SetActive (b2Body.cpp:496-544) */
void b2o_set_active(b2o_world* w, int body, int active)
{
	body_t* b = &w->bodies[body];
	if (b->dead || ((b->flags & BF_ACTIVE) != 0) == (active != 0)) return;
	if (active)
	{
		b->flags |= BF_ACTIVE;
		/* proxies for all fixtures, newest first (b2Fixture::CreateProxies: fat AABB at the body's transform, the move is
		 * buffered); contacts appear with the next pair update */
		for (int f = b->fixtureHead; f >= 0; f = w->fixtures[f].nextInBody)
		{
			fixture_t* fx = &w->fixtures[f];
			float aabb[4];
			shape_aabb(&fx->shape, b->xf, aabb);
			fx->fat[0] = aabb[0] - B2O_AABB_EXTENSION;
			fx->fat[1] = aabb[1] - B2O_AABB_EXTENSION;
			fx->fat[2] = aabb[2] + B2O_AABB_EXTENSION;
			fx->fat[3] = aabb[3] + B2O_AABB_EXTENSION;
			fx->proxyId = alloc_proxy_id(w);
			buffer_move(w, f);
		}
		return;
	}
	b->flags &= ~BF_ACTIVE;
	for (int f = b->fixtureHead; f >= 0; f = w->fixtures[f].nextInBody) release_proxy(w, f);
	for (int e = b->contactHead; e >= 0; )
	{
		int next = w->contacts[e >> 1].next[e & 1];
		destroy_contact(w, e >> 1);
		e = next;
	}
}

/* b2Body::SetType (b2Body.cpp:118-188) */
void b2o_set_type(b2o_world* w, int body, int type)
{
	body_t* b = &w->bodies[body];
	if (b->dead || b->type == type) return;
	if (b->type == 0)
	{
		/* out of m_staticBodies (their order is never used), to the end of m_nonStaticBodies */
		GROW(w->nonStatic, w->capNonStatic, w->nNonStatic + 1, int);
		b->worldIndex = w->nNonStatic;
		w->nonStatic[w->nNonStatic++] = body;
	}
	b->type = type;
	reset_mass(w, b);
	if (type == 0)
	{
		b->v = v_make(0.0f, 0.0f);
		b->w = 0.0f;
		b->a0 = b->a;
		b->c0 = b->c;
		sync_body_fixtures(w, b);
		int slot = b->worldIndex, last = w->nonStatic[w->nNonStatic - 1];
		w->nonStatic[slot] = last;
		w->bodies[last].worldIndex = slot;
		w->nNonStatic--;
		b->worldIndex = -1;
	}
	set_awake(b);
	b->force = v_make(0.0f, 0.0f);
	b->torque = 0.0f;
	for (int e = b->contactHead; e >= 0; )
	{
		int next = w->contacts[e >> 1].next[e & 1];
		destroy_contact(w, e >> 1);
		e = next;
	}
	/* TouchProxy on every proxy, newest fixture first: new contacts can form with the next pair update */
	for (int f = b->fixtureHead; f >= 0; f = w->fixtures[f].nextInBody)
		if (w->fixtures[f].proxyId >= 0) buffer_move(w, f);
}

/* b2Body::SetAwake (b2Body.h:690-718) */
void b2o_set_awake(b2o_world* w, int body, int awake)
{
	body_t* b = &w->bodies[body];
	if (awake) { set_awake(b); return; }
	b->flags &= ~BF_AWAKE;
	b->sleepTime = 0.0f;
	b->v = v_make(0.0f, 0.0f);
	b->w = 0.0f;
	b->force = v_make(0.0f, 0.0f);
	b->torque = 0.0f;
}

/* b2Body::SetBullet (b2Body.cpp:575-601) */
void b2o_set_bullet(b2o_world* w, int body, int bullet)
{
	body_t* b = &w->bodies[body];
	int was = (b->flags & BF_BULLET) != 0;
	if (bullet) b->flags |= BF_BULLET; else b->flags &= ~BF_BULLET;
	if (was != (bullet != 0)) recalc_body(w, body, -1);
}

/* b2Body::ApplyLinearImpulse / ToCenter / ApplyAngularImpulse (b2Body.h:885-950) */
void b2o_apply_linear_impulse(b2o_world* w, int body, float ix, float iy, float px, float py, int toCenter, int wake)
{
	body_t* b = &w->bodies[body];
	if (b->type != 2) return;
	if (wake && (b->flags & BF_AWAKE) == 0) set_awake(b);
	if ((b->flags & BF_AWAKE) == 0) return;
	float sx = b->invMass * ix, sy = b->invMass * iy;
	b->v.x += sx;
	b->v.y += sy;
	if (!toCenter) b->w += b->invI * ((px - b->c.x) * iy - (py - b->c.y) * ix);
}

void b2o_apply_angular_impulse(b2o_world* w, int body, float impulse, int wake)
{
	body_t* b = &w->bodies[body];
	if (b->type != 2) return;
	if (wake && (b->flags & BF_AWAKE) == 0) set_awake(b);
	if (b->flags & BF_AWAKE) b->w += b->invI * impulse;
}

/* b2Fixture::SetSensor / SetThickShape / Refilter / SetFilterData (b2Fixture.cpp:180-257) */
int b2o_get_joint_limit_state(const b2o_world* w, int joint) { return w->joints[joint].limitState; }

/* the scalar setters of the joint classes (plain assignments: b2DistanceJoint.h:117, b2RopeJoint.h:80, b2FrictionJoint.cpp:206-228,
 * b2MotorJoint.cpp:222-251, b2MouseJoint.cpp:48-76, b2GearJoint.cpp:402-406); param as in include/b2hip.h */
int b2o_joint_set_param(b2o_world* w, int joint, int param, float value)
{
	revolute_t* j = &w->joints[joint];
	const int t = j->type;
	if (param == 0 && (t == B2O_JOINT_DISTANCE || t == B2O_JOINT_ROPE)) j->length = value;
	else if (param == 1 && (t == B2O_JOINT_FRICTION || t == B2O_JOINT_MOTOR || t == B2O_JOINT_MOUSE)) j->maxForce = value;
	else if (param == 2 && (t == B2O_JOINT_FRICTION || t == B2O_JOINT_MOTOR)) j->maxTorque = value;
	else if (param == 3 && t == B2O_JOINT_GEAR) j->ratio = value;
	else if (param == 4 && t == B2O_JOINT_MOTOR) j->correctionFactor = value;
	else return -1;
	return 0;
}

/* b2World::ShiftOrigin (b2World.cpp:1862-1887; b2DynamicTree.cpp:768-776; b2MouseJoint / b2PulleyJoint::ShiftOrigin) */
void b2o_shift_origin(b2o_world* w, float x, float y)
{
	vec2 o = v_make(x, y);
	for (int i = 0; i < w->nBodies; ++i)
	{
		body_t* b = &w->bodies[i];
		b->xf.p = v_sub(b->xf.p, o);
		b->c0 = v_sub(b->c0, o);
		b->c = v_sub(b->c, o);
	}
	for (int f = 0; f < w->nFixtures; ++f)
	{
		fixture_t* fx = &w->fixtures[f];
		fx->fat[0] -= x; fx->fat[1] -= y; fx->fat[2] -= x; fx->fat[3] -= y;
	}
	for (int j = 0; j < w->nJoints; ++j)
	{
		revolute_t* jn = &w->joints[j];
		if (jn->type == B2O_JOINT_MOUSE) jn->localAnchorA = v_sub(jn->localAnchorA, o);
		else if (jn->type == B2O_JOINT_PULLEY)
		{
			jn->groundAnchorA = v_sub(jn->groundAnchorA, o);
			jn->groundAnchorB = v_sub(jn->groundAnchorB, o);
		}
	}
}

/* b2Body::SetLinearDamping / SetAngularDamping / SetGravityScale (b2Body.h:620-648) */
void b2o_set_body_damping(b2o_world* w, int body, float linearDamping, float angularDamping, float gravityScale)
{
	body_t* b = &w->bodies[body];
	b->linearDamping = linearDamping;
	b->angularDamping = angularDamping;
	b->gravityScale = gravityScale;
}

/* b2Body::SetFixedRotation (b2Body.cpp:546-565) */
void b2o_set_fixed_rotation(b2o_world* w, int body, int flag)
{
	body_t* b = &w->bodies[body];
	if (((b->flags & BF_FIXEDROT) != 0) == (flag != 0)) return;
	if (flag) b->flags |= BF_FIXEDROT; else b->flags &= ~BF_FIXEDROT;
	b->w = 0.0f;
	reset_mass(w, b);
}

/* b2Body::SetSleepingAllowed (b2Body.h:674-688) */
void b2o_set_sleeping_allowed(b2o_world* w, int body, int flag)
{
	body_t* b = &w->bodies[body];
	if (flag) b->flags |= BF_AUTOSLEEP;
	else
	{
		b->flags &= ~BF_AUTOSLEEP;
		set_awake(b);
	}
}

/* b2Body::SetMassData (b2Body.cpp:387-424); set == 0: b2Body::ResetMassData */
void b2o_set_mass_data(b2o_world* w, int body, int set, float mass, float inertia, float cx, float cy)
{
	body_t* b = &w->bodies[body];
	if (!set)
	{
		reset_mass(w, b);
		return;
	}
	if (b->type != 2) return;
	b->invMass = 0.0f;
	b->I = 0.0f;
	b->invI = 0.0f;
	b->mass = mass;
	if (b->mass <= 0.0f) b->mass = 1.0f;
	b->invMass = 1.0f / b->mass;
	vec2 center = v_make(cx, cy);
	if (inertia > 0.0f && (b->flags & BF_FIXEDROT) == 0)
	{
		b->I = inertia - b->mass * v_dot(center, center);
		b->invI = 1.0f / b->I;
	}
	vec2 oldCenter = b->c;
	b->localCenter = center;
	b->c0 = b->c = xf_mul(b->xf, center);
	b->v = v_add(b->v, v_cross_sv(b->w, v_sub(b->c, oldCenter)));
}

/* b2Fixture::SetDensity / SetFriction / SetRestitution (b2Fixture.h:306-334) */
void b2o_fixture_set_material(b2o_world* w, int fixture, float density, float friction, float restitution)
{
	fixture_t* f = &w->fixtures[fixture];
	f->density = density;
	f->friction = friction;
	f->restitution = restitution;
}

void b2o_fixture_set_sensor(b2o_world* w, int fixture, int isSensor)
{
	fixture_t* f = &w->fixtures[fixture];
	if ((f->isSensor != 0) == (isSensor != 0)) return;
	set_awake(&w->bodies[f->body]);
	f->isSensor = isSensor != 0;
	recalc_body(w, f->body, fixture);
}

void b2o_fixture_set_thick(b2o_world* w, int fixture, int thick)
{
	fixture_t* f = &w->fixtures[fixture];
	if ((f->thick != 0) == (thick != 0)) return;
	f->thick = thick != 0;
	recalc_body(w, f->body, fixture);
}

void b2o_fixture_refilter(b2o_world* w, int fixture)
{
	fixture_t* f = &w->fixtures[fixture];
	for (int e = w->bodies[f->body].contactHead; e >= 0; e = w->contacts[e >> 1].next[e & 1])
	{
		contact_t* c = &w->contacts[e >> 1];
		if (c->fixtureA == fixture || c->fixtureB == fixture) c->flags |= CF_FILTER;
	}
	if (w->bodies[f->body].flags & BF_ACTIVE) buffer_move(w, fixture); /* TouchProxy */
}

void b2o_fixture_set_filter(b2o_world* w, int fixture, uint16_t categoryBits, uint16_t maskBits, int16_t groupIndex)
{
	fixture_t* f = &w->fixtures[fixture];
	f->categoryBits = categoryBits;
	f->maskBits = maskBits;
	f->groupIndex = groupIndex;
	b2o_fixture_refilter(w, fixture);
}

void b2o_joint_set_spring(b2o_world* w, int joint, float frequencyHz, float dampingRatio)
{
	revolute_t* j = &w->joints[joint];
	j->frequencyHz = frequencyHz;
	j->dampingRatio = dampingRatio;
}

int b2o_body_is_destroyed(const b2o_world* w, int body) { return w->bodies[body].dead; }
int b2o_fixture_is_destroyed(const b2o_world* w, int fixture) { return w->fixtures[fixture].dead; }


/* ---- spatial ownership (see struct b2o_world::spatial) ------------------------------------------------------------------------
 * A serial restatement of the protocol of box2d-mt_amd/csrc/b2d_kernels_spatial.h, for the multi-process CPU tests (gloo):
 * E1 / E4 rows and fat AABBs of the bodies a rank moved (full rows: every rank holds every body's state), E2 new pairs, E3
 * migration of components that a new contact or joint joins over an ownership boundary. Contacts are named by their slot
 * (every rank creates and destroys the same contacts in the same order, so the slots agree). Not restated: the merge of
 * contacts created INSIDE TOI sub-steps over the ranks (k_sp_merge_tails) - a step in which an event creates one fails. */
#define SP_BODY_WORDS 18
static int32_t sp_fb(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static float sp_bf(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

/* all-gather of a variable number of words per rank: the counts first, then slabs of the largest count. Returns a buffer of
 * shardCount * (*stride) words (caller frees) and every rank's count. */
static int32_t* sp_gather(b2o_world* w, const int32_t* send, int words, int* counts, int* stride)
{
	const int n = w->shardCount;
	int32_t mine = words;
	int32_t* all = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
	if (w->gatherFn == NULL || w->gatherFn(w->gatherUser, &mine, sizeof(mine), all) != 0) { w->spFailed = 1; for (int r = 0; r < n; ++r) all[r] = r == w->shardRank ? words : 0; }
	int most = 1;
	for (int r = 0; r < n; ++r) { counts[r] = all[r]; if (all[r] > most) most = all[r]; }
	free(all);
	int32_t* pad = (int32_t*)calloc((size_t)most, sizeof(int32_t));
	if (words > 0) memcpy(pad, send, sizeof(int32_t) * (size_t)words);
	int32_t* recv = (int32_t*)calloc((size_t)most * (size_t)n, sizeof(int32_t));
	if (w->gatherFn == NULL || w->gatherFn(w->gatherUser, pad, sizeof(int32_t) * (size_t)most, recv) != 0) w->spFailed = 1;
	w->spBytes += (long long)sizeof(int32_t) * most * (n - 1);
	free(pad);
	*stride = most;
	return recv;
}

static void sp_body_record(const body_t* b, int id, int32_t* o)
{
	o[0] = id;
	o[1] = sp_fb(b->xf.p.x); o[2] = sp_fb(b->xf.p.y); o[3] = sp_fb(b->xf.q.s); o[4] = sp_fb(b->xf.q.c);
	o[5] = sp_fb(b->c0.x); o[6] = sp_fb(b->c0.y); o[7] = sp_fb(b->c.x); o[8] = sp_fb(b->c.y);
	o[9] = sp_fb(b->a0); o[10] = sp_fb(b->a); o[11] = 0; /* (alpha0 is 0 once the owner's step has ended) */
	o[12] = sp_fb(b->v.x); o[13] = sp_fb(b->v.y); o[14] = sp_fb(b->w); o[15] = sp_fb(b->sleepTime);
	o[16] = (b->flags & BF_AWAKE) ? 1 : 0;
	o[17] = 0;
}
static void sp_body_apply(b2o_world* w, const int32_t* o)
{
	body_t* b = &w->bodies[o[0]];
	b->xf.p = v_make(sp_bf(o[1]), sp_bf(o[2])); b->xf.q.s = sp_bf(o[3]); b->xf.q.c = sp_bf(o[4]);
	b->c0 = v_make(sp_bf(o[5]), sp_bf(o[6])); b->c = v_make(sp_bf(o[7]), sp_bf(o[8]));
	b->a0 = sp_bf(o[9]); b->a = sp_bf(o[10]); b->alpha0 = 0.0f;
	b->v = v_make(sp_bf(o[12]), sp_bf(o[13])); b->w = sp_bf(o[14]); b->sleepTime = sp_bf(o[15]);
	if (o[16]) b->flags |= BF_AWAKE;
	else
	{
		b->flags &= ~BF_AWAKE;
		b->force = v_make(0.0f, 0.0f);
		b->torque = 0.0f;
	}
}

/* E1 (behind SynchronizeFixtures: the bodies that were in an island) / E4 (behind SolveTOI: every body this rank owns - the
 * shim keeps no snapshot to tell which ones the events touched) with the fat AABBs of their fixtures.
 * layout: [nBodies][body records][fixture id, fat x 4]... */
static void sp_exchange_state(b2o_world* w, int afterToi)
{
	if (w->shardCount < 2) return;
	int cap = 1024, n = 1;
	int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
	int nb = 0;
	for (int pass = 0; pass < 2; ++pass)
	{
		for (int i = 0; i < w->nBodies; ++i)
		{
			const body_t* b = &w->bodies[i];
			if (b->type == 0 || b->dead || w->owner[i] != (unsigned char)w->shardRank) continue;
			if (!afterToi && (b->flags & BF_ISLAND) == 0) continue;
			if (pass == 0)
			{
				GROW(out, cap, n + SP_BODY_WORDS, int32_t);
				sp_body_record(b, i, out + n);
				n += SP_BODY_WORDS;
				++nb;
			}
			else
			{
				for (int f = b->fixtureHead; f >= 0; f = w->fixtures[f].nextInBody)
				{
					const fixture_t* fx = &w->fixtures[f];
					if (fx->proxyId < 0 || fx->dead) continue;
					GROW(out, cap, n + 5, int32_t);
					out[n] = f;
					for (int k = 0; k < 4; ++k) out[n + 1 + k] = sp_fb(fx->fat[k]);
					n += 5;
				}
			}
		}
	}
	out[0] = nb;
	int counts[8], stride = 0;
	int32_t* in = sp_gather(w, out, n, counts, &stride);
	for (int r = 0; r < w->shardCount; ++r)
	{
		if (r == w->shardRank) continue;
		const int32_t* o = in + (size_t)r * (size_t)stride;
		const int bodies = o[0];
		const int32_t* q = o + 1;
		for (int k = 0; k < bodies; ++k, q += SP_BODY_WORDS) sp_body_apply(w, q);
		for (; q + 5 <= o + counts[r]; q += 5)
		{
			fixture_t* fx = &w->fixtures[q[0]];
			for (int k = 0; k < 4; ++k) fx->fat[k] = sp_bf(q[1 + k]);
		}
	}
	free(in);
	free(out);
}

/* E2: the candidate pairs of this rank's moved proxies out, everybody's in (the sort that follows drops the duplicates) */
static void sp_exchange_pairs(b2o_world* w, pair_t** pairs, int* n, int* cap)
{
	int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * *n + 1));
	for (int i = 0; i < *n; ++i) { out[2 * i] = (*pairs)[i].fLo; out[2 * i + 1] = (*pairs)[i].fHi; }
	int counts[8], stride = 0;
	int32_t* in = sp_gather(w, out, 2 * *n, counts, &stride);
	for (int r = 0; r < w->shardCount; ++r)
	{
		if (r == w->shardRank) continue;
		const int32_t* o = in + (size_t)r * (size_t)stride;
		for (int k = 0; k + 1 < counts[r]; k += 2)
		{
			if (*n == *cap)
			{
				*cap *= 2;
				*pairs = (pair_t*)realloc(*pairs, sizeof(pair_t) * (size_t)*cap);
			}
			pair_t pr;
			pr.fLo = o[k]; pr.fHi = o[k + 1];
			pr.lo = w->fixtures[pr.fLo].proxyId; pr.hi = w->fixtures[pr.fHi].proxyId;
			(*pairs)[(*n)++] = pr;
		}
	}
	free(in);
	free(out);
}

static int sp_find(int* parent, int i)
{
	while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
	return i;
}
static void sp_union(int* parent, int a, int b)
{
	a = sp_find(parent, a); b = sp_find(parent, b);
	if (a == b) return;
	if (a < b) parent[b] = a; else parent[a] = b;
}

/* E3: contacts / joints that join non-static bodies of different owners make their components (over ALL existing contacts
 * between non-static bodies, and joints) merge under the owner that holds most of the bodies - every rank computes the
 * same from the replicated structure - and the losers ship the content of what they maintained. */
static void sp_resolve(b2o_world* w)
{
	int straddle = 0;
	for (int s = 0; s < w->nContactSlots && !straddle; ++s)
	{
		const contact_t* c = &w->contacts[s];
		if (c->alive && w->bodies[c->bodyA].type != 0 && w->bodies[c->bodyB].type != 0 && w->owner[c->bodyA] != w->owner[c->bodyB]) straddle = 1;
	}
	for (int j = 0; j < w->nJoints && !straddle; ++j)
	{
		const revolute_t* jn = &w->joints[j];
		if (jn->type >= 0 && w->bodies[jn->bodyA].type != 0 && w->bodies[jn->bodyB].type != 0 && w->owner[jn->bodyA] != w->owner[jn->bodyB]) straddle = 1;
	}
	if (!straddle) return;
	const int nb = w->nBodies, R = w->shardCount;
	int* parent = (int*)malloc(sizeof(int) * (size_t)(nb + 1));
	for (int i = 0; i < nb; ++i) parent[i] = i;
	for (int s = 0; s < w->nContactSlots; ++s)
	{
		const contact_t* c = &w->contacts[s];
		if (c->alive && w->bodies[c->bodyA].type != 0 && w->bodies[c->bodyB].type != 0) sp_union(parent, c->bodyA, c->bodyB);
	}
	for (int j = 0; j < w->nJoints; ++j)
	{
		const revolute_t* jn = &w->joints[j];
		if (jn->type < 0) continue;
		const int nsA = w->bodies[jn->bodyA].type != 0, nsB = w->bodies[jn->bodyB].type != 0;
		if (nsA && nsB) sp_union(parent, jn->bodyA, jn->bodyB);
		if (jn->type == B2O_JOINT_GEAR)
		{
			const int others[2] = { jn->bodyC, jn->bodyD };
			for (int k = 0; k < 2; ++k)
			{
				if (others[k] < 0 || w->bodies[others[k]].type == 0) continue;
				if (nsA) sp_union(parent, jn->bodyA, others[k]); else if (nsB) sp_union(parent, jn->bodyB, others[k]);
			}
		}
	}
	/* components that hold a straddling contact / joint: bodies per owner, then the new owner */
	int* count = (int*)calloc((size_t)(nb + 1) * (size_t)R, sizeof(int));
	unsigned char* marked = (unsigned char*)calloc((size_t)nb + 1, 1);
	for (int s = 0; s < w->nContactSlots; ++s)
	{
		const contact_t* c = &w->contacts[s];
		if (c->alive && w->bodies[c->bodyA].type != 0 && w->bodies[c->bodyB].type != 0 && w->owner[c->bodyA] != w->owner[c->bodyB]) marked[sp_find(parent, c->bodyA)] = 1;
	}
	for (int j = 0; j < w->nJoints; ++j)
	{
		const revolute_t* jn = &w->joints[j];
		if (jn->type >= 0 && w->bodies[jn->bodyA].type != 0 && w->bodies[jn->bodyB].type != 0 && w->owner[jn->bodyA] != w->owner[jn->bodyB]) marked[sp_find(parent, jn->bodyA)] = 1;
	}
	for (int i = 0; i < nb; ++i)
	{
		if (w->bodies[i].type == 0) continue;
		const int root = sp_find(parent, i);
		if (marked[root]) count[(size_t)root * (size_t)R + w->owner[i]] += 1;
	}
	unsigned char* newOwner = (unsigned char*)malloc((size_t)nb + 1);
	for (int i = 0; i < nb; ++i)
	{
		newOwner[i] = w->owner[i];
		if (w->bodies[i].type == 0) continue;
		const int root = sp_find(parent, i);
		if (!marked[root]) continue;
		int best = 0;
		for (int r = 1; r < R; ++r) if (count[(size_t)root * (size_t)R + r] > count[(size_t)root * (size_t)R + best]) best = r;
		newOwner[i] = (unsigned char)best;
	}
	/* what this rank loses: [nBodies, nContacts, nJoints] body records, contact content (slot, flags, 16 manifold words,
	 * friction, restitution, tangent speed, toi, toiCount, new owner), joint records (id, 4 impulses, limit state) */
	int cap = 1024, n = 3, cb = 0, cc = 0, cj = 0;
	int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
	const int me = w->shardRank;
	for (int i = 0; i < nb; ++i)
	{
		if (w->bodies[i].type == 0 || w->owner[i] != me || newOwner[i] == me) continue;
		GROW(out, cap, n + SP_BODY_WORDS, int32_t);
		sp_body_record(&w->bodies[i], i, out + n);
		out[n + 11] = sp_fb(w->bodies[i].alpha0);
		out[n + 17] = newOwner[i];
		n += SP_BODY_WORDS;
		++cb;
		w->spMigrated += 1;
	}
	for (int s = 0; s < w->nContactSlots; ++s)
	{
		const contact_t* c = &w->contacts[s];
		if (!c->alive) continue;
		const int nsA = w->bodies[c->bodyA].type != 0, nsB = w->bodies[c->bodyB].type != 0;
		if (!nsA && !nsB) continue;
		if (nsA && nsB && w->owner[c->bodyA] != w->owner[c->bodyB]) continue; /* (straddling: nobody has content for it yet) */
		const int body = nsA ? c->bodyA : c->bodyB;
		if (w->owner[body] != me || newOwner[body] == me) continue;
		GROW(out, cap, n + 24, int32_t);
		int32_t* o = out + n;
		o[0] = s; o[1] = (int32_t)c->flags;
		memcpy(o + 2, &c->m, sizeof(manifold)); /* 16 words */
		o[18] = sp_fb(c->friction); o[19] = sp_fb(c->restitution); o[20] = sp_fb(c->tangentSpeed); o[21] = sp_fb(c->toi);
		o[22] = c->toiCount; o[23] = newOwner[body];
		n += 24;
		++cc;
	}
	for (int j = 0; j < w->nJoints; ++j)
	{
		const revolute_t* jn = &w->joints[j];
		if (jn->type < 0) continue;
		const int nsA = w->bodies[jn->bodyA].type != 0, nsB = w->bodies[jn->bodyB].type != 0;
		if (!nsA && !nsB) continue;
		if (nsA && nsB && w->owner[jn->bodyA] != w->owner[jn->bodyB]) continue;
		const int body = nsA ? jn->bodyA : jn->bodyB;
		if (w->owner[body] != me || newOwner[body] == me) continue;
		GROW(out, cap, n + 6, int32_t);
		int32_t* o = out + n;
		o[0] = j;
		o[1] = sp_fb(jn->impulse[0]);
		o[2] = sp_fb(jn->type == B2O_JOINT_WHEEL ? jn->springImpulse : jn->impulse[1]);
		o[3] = sp_fb(jn->impulse[2]);
		o[4] = sp_fb(jn->motorImpulse);
		o[5] = jn->limitState;
		n += 6;
		++cj;
	}
	out[0] = cb; out[1] = cc; out[2] = cj;
	int counts[8], stride = 0;
	int32_t* in = sp_gather(w, out, n, counts, &stride);
	for (int r = 0; r < R; ++r)
	{
		if (r == me) continue;
		const int32_t* o = in + (size_t)r * (size_t)stride;
		const int rb = o[0], rc = o[1], rj = o[2];
		const int32_t* q = o + 3;
		for (int k = 0; k < rb; ++k, q += SP_BODY_WORDS)
		{
			sp_body_apply(w, q);
			w->bodies[q[0]].alpha0 = sp_bf(q[11]);
		}
		for (int k = 0; k < rc; ++k, q += 24)
		{
			if (q[23] != me) continue;
			contact_t* c = &w->contacts[q[0]];
			/* (the structural bits are equal on every rank: the TOI-candidate bit, the filter bit; the rest is content) */
			c->flags = (uint32_t)q[1];
			memcpy(&c->m, q + 2, sizeof(manifold));
			c->friction = sp_bf(q[18]); c->restitution = sp_bf(q[19]); c->tangentSpeed = sp_bf(q[20]); c->toi = sp_bf(q[21]);
			c->toiCount = q[22];
		}
		for (int k = 0; k < rj; ++k, q += 6)
		{
			revolute_t* jn = &w->joints[q[0]];
			jn->impulse[0] = sp_bf(q[1]);
			if (jn->type == B2O_JOINT_WHEEL) jn->springImpulse = sp_bf(q[2]); else jn->impulse[1] = sp_bf(q[2]);
			jn->impulse[2] = sp_bf(q[3]);
			jn->motorImpulse = sp_bf(q[4]);
			jn->limitState = q[5];
		}
	}
	memcpy(w->owner, newOwner, (size_t)nb);
	w->spResolves += 1;
	free(in); free(out); free(newOwner); free(marked); free(count); free(parent);
}

static unsigned char sp_strip_of(const b2o_world* w, float x)
{
	int r = 0;
	while (r + 1 < w->shardCount && x >= w->spBounds[r + 1]) ++r;
	return (unsigned char)r;
}

static void sp_begin_step(b2o_world* w)
{
	w->spBytes = 0;
	if (w->capOwner < w->nBodies)
	{
		/* bodies created since the world was sharded fall into the strip of their x */
		const int old = w->capOwner;
		w->owner = (unsigned char*)realloc(w->owner, (size_t)w->nBodies + 64);
		for (int i = old; i < w->nBodies; ++i) w->owner[i] = w->bodies[i].type == 0 ? 0 : sp_strip_of(w, w->bodies[i].c.x);
		w->capOwner = w->nBodies;
		w->spOwnersDirty = 1;
	}
	if (w->spOwnersDirty)
	{
		sp_resolve(w);
		w->spOwnersDirty = 0;
	}
}

typedef struct { float x; int i; } sp_xi;
static int sp_xi_cmp(const void* a, const void* b)
{
	const sp_xi* p = (const sp_xi*)a;
	const sp_xi* q = (const sp_xi*)b;
	if (p->x != q->x) return p->x < q->x ? -1 : 1;
	return p->i < q->i ? -1 : (p->i > q->i ? 1 : 0);
}

int b2o_shard_spatial(b2o_world* w, int rank, int count, const unsigned char* owners)
{
	if (count < 1 || count > 8 || rank < 0 || rank >= count) return -1;
	w->shardRank = rank;
	w->shardCount = count;
	w->owner = (unsigned char*)realloc(w->owner, (size_t)w->nBodies + 64);
	w->capOwner = w->nBodies;
	memset(w->owner, 0, (size_t)w->nBodies + 64);
	if (owners)
	{
		for (int i = 0; i < w->nBodies; ++i) w->owner[i] = w->bodies[i].type != 0 && owners[i] < count ? owners[i] : 0;
		for (int r = 0; r <= 8; ++r) w->spBounds[r] = 0.0f;
	}
	else
	{
		/* strips of equal body count along x (the rule of b2hip_shard_spatial) */
		sp_xi* xs = (sp_xi*)malloc(sizeof(sp_xi) * (size_t)(w->nBodies + 1));
		int n = 0;
		for (int i = 0; i < w->nBodies; ++i)
		{
			if (w->bodies[i].type == 0 || w->bodies[i].dead) continue;
			xs[n].x = w->bodies[i].c.x; xs[n].i = i; ++n;
		}
		qsort(xs, (size_t)n, sizeof(sp_xi), sp_xi_cmp);
		w->spBounds[0] = -3.0e38f;
		for (int r = 1; r < count; ++r)
		{
			size_t k = (size_t)n * (size_t)r / (size_t)count;
			if (n > 0 && k > (size_t)n - 1) k = (size_t)n - 1;
			w->spBounds[r] = n > 0 ? xs[k].x : 0.0f;
		}
		for (int r = count; r <= 8; ++r) w->spBounds[r] = 3.0e38f;
		for (int k = 0; k < n; ++k) w->owner[xs[k].i] = sp_strip_of(w, xs[k].x);
		free(xs);
	}
	w->spatial = 1;
	w->spOwnersDirty = 1;
	w->spMigrated = 0;
	w->spResolves = 0;
	w->spFailed = 0;
	return 0;
}

void b2o_set_shard_gather(b2o_world* w, int (*fn)(void*, const void*, size_t, void*), void* user) { w->gatherFn = fn; w->gatherUser = user; }
int b2o_spatial_failed(const b2o_world* w) { return w->spFailed; }
int b2o_get_body_owners(const b2o_world* w, int cap, unsigned char* out)
{
	if (!w->spatial) return -1;
	const int n = cap < w->nBodies ? cap : w->nBodies;
	for (int i = 0; i < n; ++i) out[i] = i < w->capOwner ? w->owner[i] : 0;
	return n;
}
void b2o_get_shard_stats(const b2o_world* w, int32_t* out8, long long* out4)
{
	int bodies = 0, proxies = 0, contacts = 0;
	for (int i = 0; i < w->nBodies && w->spatial; ++i)
	{
		if (w->bodies[i].type == 0 || w->bodies[i].dead || w->owner[i] != (unsigned char)w->shardRank) continue;
		++bodies;
		for (int f = w->bodies[i].fixtureHead; f >= 0; f = w->fixtures[f].nextInBody) if (w->fixtures[f].proxyId >= 0) ++proxies;
	}
	for (int s = 0; s < w->nContactSlots && w->spatial; ++s) if (w->contacts[s].alive && !sp_foreign_contact(w, &w->contacts[s])) ++contacts;
	out8[0] = w->shardRank; out8[1] = w->shardCount; out8[2] = bodies; out8[3] = proxies; out8[4] = contacts; out8[5] = 0; out8[6] = 0; out8[7] = 0;
	out4[0] = w->spMigrated; out4[1] = w->spResolves; out4[2] = w->spBytes; out4[3] = 0;
}
