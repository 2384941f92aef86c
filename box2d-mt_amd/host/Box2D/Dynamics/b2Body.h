// Drop-in header: b2Body (reference: Box2D/Dynamics/b2Body.h:36-992). A body is a host handle
// onto row `GetDeviceId()` of the SoA body arrays in HBM; getters read the host mirror that every
// Step() refreshes with one device-to-host copy.
#ifndef B2_BODY_H
#define B2_BODY_H

#include "Box2D/Common/b2Math.h"
#include "Box2D/Collision/Shapes/b2Shape.h"

class b2Fixture;
class b2Joint;
class b2Contact;
class b2World;
struct b2FixtureDef;
struct b2JointEdge;
struct b2ContactEdge;
struct b2JointEdge;

enum b2BodyType
{
	b2_staticBody = 0,
	b2_kinematicBody,
	b2_dynamicBody
};

struct b2BodyDef
{
	b2BodyDef()
	{
		userData = nullptr;
		position.Set(0.0f, 0.0f);
		angle = 0.0f;
		linearVelocity.Set(0.0f, 0.0f);
		angularVelocity = 0.0f;
		linearDamping = 0.0f;
		angularDamping = 0.0f;
		allowSleep = true;
		awake = true;
		fixedRotation = false;
		bullet = false;
		type = b2_staticBody;
		active = true;
		gravityScale = 1.0f;
	}
	b2BodyType type;
	b2Vec2 position;
	float32 angle;
	b2Vec2 linearVelocity;
	float32 angularVelocity;
	float32 linearDamping;
	float32 angularDamping;
	bool allowSleep;
	bool awake;
	bool fixedRotation;
	bool bullet;
	bool active;
	void* userData;
	float32 gravityScale;
};

struct b2ContactEdge;

class b2Body
{
public:
	b2Fixture* CreateFixture(const b2FixtureDef* def);
	b2Fixture* CreateFixture(const b2Shape* shape, float32 density);

	const b2Transform& GetTransform() const;
	const b2Vec2& GetPosition() const;
	float32 GetAngle() const;
	const b2Vec2& GetWorldCenter() const;
	const b2Vec2& GetLocalCenter() const;
	void SetLinearVelocity(const b2Vec2& v);
	const b2Vec2& GetLinearVelocity() const;
	void SetAngularVelocity(float32 omega);
	float32 GetAngularVelocity() const;
	void ApplyForce(const b2Vec2& force, const b2Vec2& point, bool wake);
	void ApplyForceToCenter(const b2Vec2& force, bool wake);
	void ApplyTorque(float32 torque, bool wake);
	void ApplyLinearImpulse(const b2Vec2& impulse, const b2Vec2& point, bool wake);
	void ApplyLinearImpulseToCenter(const b2Vec2& impulse, bool wake);
	void ApplyAngularImpulse(float32 impulse, bool wake);
	/// b2Body.cpp:451-473: pose and sweep are set, the fixtures' proxies follow (a no-op while the world is locked)
	void SetTransform(const b2Vec2& position, float32 angle);
	void SetAwake(bool flag);
	void SetBullet(bool flag);
	void SetActive(bool flag);
	void SetType(b2BodyType type);
	/// b2Body.cpp:238-308; the fixture pointer is dead afterwards
	void DestroyFixture(b2Fixture* fixture);
	/// The body's contact edges, newest contact first (b2Body.h:431-436); valid until the next Step or edit.
	b2ContactEdge* GetContactList();
	const b2ContactEdge* GetContactList() const { return const_cast<b2Body*>(this)->GetContactList(); }
	/// The joints attached to this body, newest first (b2Body.h:426-429)
	b2JointEdge* GetJointList() { return m_jointList; }
	const b2JointEdge* GetJointList() const { return m_jointList; }
	float32 GetMass() const;
	float32 GetInertia() const;
	void GetMassData(b2MassData* data) const;
	/// b2Body.cpp:387-424 (dynamic bodies only; a no-op while the world is locked) and b2Body.cpp:310-385
	void SetMassData(const b2MassData* data);
	void ResetMassData();
	/// b2Body.h:620-648, b2Body.cpp:546-565, b2Body.h:674-688
	void SetLinearDamping(float32 linearDamping);
	void SetAngularDamping(float32 angularDamping);
	void SetGravityScale(float32 scale);
	void SetFixedRotation(bool flag);
	void SetSleepingAllowed(bool flag);
	/// b2Body.h:586-594: velocity of a point attached to the body
	b2Vec2 GetLinearVelocityFromWorldPoint(const b2Vec2& worldPoint) const
	{
		return GetLinearVelocity() + b2Cross(GetAngularVelocity(), worldPoint - GetWorldCenter());
	}
	b2Vec2 GetLinearVelocityFromLocalPoint(const b2Vec2& localPoint) const { return GetLinearVelocityFromWorldPoint(GetWorldPoint(localPoint)); }

	b2Vec2 GetWorldPoint(const b2Vec2& localPoint) const { return b2Mul(GetTransform(), localPoint); }
	b2Vec2 GetWorldVector(const b2Vec2& localVector) const { return b2Mul(GetTransform().q, localVector); }
	b2Vec2 GetLocalPoint(const b2Vec2& worldPoint) const { return b2MulT(GetTransform(), worldPoint); }
	b2Vec2 GetLocalVector(const b2Vec2& worldVector) const { return b2MulT(GetTransform().q, worldVector); }

	float32 GetLinearDamping() const { return m_linearDamping; }
	float32 GetAngularDamping() const { return m_angularDamping; }
	float32 GetGravityScale() const { return m_gravityScale; }
	b2BodyType GetType() const { return m_type; }
	bool IsBullet() const { return m_bullet; }
	bool IsSleepingAllowed() const { return m_allowSleep; }
	bool IsAwake() const;
	bool IsActive() const { return m_active; }
	bool IsFixedRotation() const { return m_fixedRotation; }

	b2Fixture* GetFixtureList() { return m_fixtureList; }
	const b2Fixture* GetFixtureList() const { return m_fixtureList; }
	b2Body* GetNext() { return m_next; }
	const b2Body* GetNext() const { return m_next; }
	void* GetUserData() const { return m_userData; }
	void SetUserData(void* data) { m_userData = data; }
	b2World* GetWorld() { return m_world; }
	const b2World* GetWorld() const { return m_world; }
	/// Row of this body in the device arrays.
	int32 GetDeviceId() const { return m_id; }

private:
	friend class b2World;
	friend class b2Fixture;
	friend class b2Joint;
	friend class b2RevoluteJoint;
	friend class b2Contact;

	b2Body(const b2BodyDef* bd, b2World* world, int32 id);
	~b2Body();

	b2World* m_world;
	int32 m_id;
	b2BodyType m_type;
	bool m_bullet, m_allowSleep, m_active, m_fixedRotation;
	float32 m_linearDamping, m_angularDamping, m_gravityScale;
	b2Body* m_prev;
	b2Body* m_next;
	b2Fixture* m_fixtureList;
	b2JointEdge* m_jointList;
	int32 m_fixtureCount;
	void* m_userData;
	// caches for the reference-returning getters (filled from the world's host mirror)
	mutable b2Transform m_xfCache;
	mutable b2Vec2 m_vecCache[3];
};

#endif
