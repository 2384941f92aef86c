// Drop-in header: listener / filter interfaces (reference: Box2D/Dynamics/b2WorldCallbacks.h:31-215).
// The default filter (group / category / mask rule) is evaluated on the device; a user-supplied
// listener is invoked on the host after the device phases (next-row work).
#ifndef B2_WORLD_CALLBACKS_H
#define B2_WORLD_CALLBACKS_H

#include "Box2D/Common/b2Settings.h"
#include "Box2D/Collision/b2Collision.h"

struct b2Vec2;
struct b2Transform;
class b2Fixture;
class b2Body;
class b2Joint;
class b2Contact;

class b2DestructionListener
{
public:
	virtual ~b2DestructionListener() {}
	virtual void SayGoodbye(b2Joint* joint) = 0;
	virtual void SayGoodbye(b2Fixture* fixture) = 0;
};

class b2ContactFilter
{
public:
	virtual ~b2ContactFilter() {}
	virtual bool ShouldCollide(b2Fixture* fixtureA, b2Fixture* fixtureB, uint32 threadId);
};

struct b2ContactImpulse
{
	float32 normalImpulses[b2_maxManifoldPoints];
	float32 tangentImpulses[b2_maxManifoldPoints];
	int32 count;
};

class b2ContactListener
{
public:
	virtual ~b2ContactListener() {}
	virtual void BeginContact(b2Contact* contact) { B2_NOT_USED(contact); }
	virtual void EndContact(b2Contact* contact) { B2_NOT_USED(contact); }
	virtual void PreSolve(b2Contact* contact, const b2Manifold* oldManifold) { B2_NOT_USED(contact); B2_NOT_USED(oldManifold); }
	virtual void PostSolve(b2Contact* contact, const b2ContactImpulse* impulse) { B2_NOT_USED(contact); B2_NOT_USED(impulse); }
	virtual bool BeginContactImmediate(b2Contact* contact, uint32 threadId) = 0;
	virtual bool EndContactImmediate(b2Contact* contact, uint32 threadId) = 0;
	virtual bool PreSolveImmediate(b2Contact* contact, const b2Manifold* oldManifold, uint32 threadId) = 0;
	virtual bool PostSolveImmediate(b2Contact* contact, const b2ContactImpulse* impulse, uint32 threadId) = 0;
};

class b2QueryCallback
{
public:
	virtual ~b2QueryCallback() {}
	virtual bool ReportFixture(b2Fixture* fixture) = 0;
};

class b2RayCastCallback
{
public:
	virtual ~b2RayCastCallback() {}
	virtual float32 ReportFixture(b2Fixture* fixture, const b2Vec2& point, const b2Vec2& normal, float32 fraction) = 0;
};

#endif
