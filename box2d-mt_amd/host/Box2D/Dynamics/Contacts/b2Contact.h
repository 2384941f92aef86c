// Drop-in header: read-only b2Contact view (reference: Box2D/Dynamics/Contacts/b2Contact.h:36-420).
// Contacts live in HBM; b2World::GetContactList() materialises this host view on demand from one
// device-to-host copy (b2hip_get_contacts).
#ifndef B2_CONTACT_H
#define B2_CONTACT_H

#include "Box2D/Common/b2Math.h"
#include "Box2D/Collision/b2Collision.h"
#include "Box2D/Dynamics/b2Fixture.h"

class b2Body;
class b2World;

inline float32 b2MixFriction(float32 friction1, float32 friction2) { return b2Sqrt(friction1 * friction2); }
inline float32 b2MixRestitution(float32 restitution1, float32 restitution2) { return restitution1 > restitution2 ? restitution1 : restitution2; }

struct b2ContactEdge
{
	b2Body* other;
	b2Contact* contact;
	b2ContactEdge* prev;
	b2ContactEdge* next;
};

class b2Contact
{
public:
	b2Manifold* GetManifold() { return &m_manifold; }
	const b2Manifold* GetManifold() const { return &m_manifold; }
	void GetWorldManifold(b2WorldManifold* worldManifold) const;
	bool IsTouching() const { return m_touching; }
	bool IsEnabled() const { return m_enabled; }
	/// Inside b2ContactListener::PreSolve: false leaves the contact out of this step's islands (b2Contact.h:117-123);
	/// the next Collide enables it again. Anywhere else the call has no effect (the view is a copy).
	void SetEnabled(bool flag) { m_enabled = flag; }
	b2Contact* GetNext() { return m_next; }
	const b2Contact* GetNext() const { return m_next; }
	b2Fixture* GetFixtureA() { return m_fixtureA; }
	const b2Fixture* GetFixtureA() const { return m_fixtureA; }
	int32 GetChildIndexA() const { return m_indexA; }
	b2Fixture* GetFixtureB() { return m_fixtureB; }
	const b2Fixture* GetFixtureB() const { return m_fixtureB; }
	int32 GetChildIndexB() const { return m_indexB; }
	/// Friction / restitution / tangent speed (b2Contact.h:129-160). Inside b2ContactListener::PreSolve the setters reach the
	/// solver of this step and stay with the contact (the conveyor belt of the Testbed); anywhere else they change the view only.
	void SetFriction(float32 friction) { m_friction = friction; }
	float32 GetFriction() const { return m_friction; }
	void ResetFriction() { m_friction = b2MixFriction(m_fixtureA->GetFriction(), m_fixtureB->GetFriction()); }
	void SetRestitution(float32 restitution) { m_restitution = restitution; }
	float32 GetRestitution() const { return m_restitution; }
	void ResetRestitution() { m_restitution = b2MixRestitution(m_fixtureA->GetRestitution(), m_fixtureB->GetRestitution()); }
	void SetTangentSpeed(float32 speed) { m_tangentSpeed = speed; }
	float32 GetTangentSpeed() const { return m_tangentSpeed; }

private:
	friend class b2World;
	b2Manifold m_manifold;
	b2Fixture* m_fixtureA;
	b2Fixture* m_fixtureB;
	int32 m_indexA = 0, m_indexB = 0; // child indices (chain shapes)
	b2Contact* m_next;
	float32 m_friction, m_restitution;
	float32 m_tangentSpeed = 0.0f;
	bool m_touching, m_enabled;
};

#endif
