// Drop-in header: b2Fixture (reference: Box2D/Dynamics/b2Fixture.h:33-366). A fixture is a host
// handle; its collision record lives in HBM under the id returned by b2hip_create_fixture.
#ifndef B2_FIXTURE_H
#define B2_FIXTURE_H

#include "Box2D/Dynamics/b2Body.h"
#include "Box2D/Collision/b2Collision.h"
#include "Box2D/Collision/Shapes/b2Shape.h"

class b2BlockAllocator;
class b2Body;
class b2BroadPhase;
class b2Fixture;

struct b2Filter
{
	b2Filter()
	{
		categoryBits = 0x0001;
		maskBits = 0xFFFF;
		groupIndex = 0;
	}
	uint16 categoryBits;
	uint16 maskBits;
	int16 groupIndex;
};

struct b2FixtureDef
{
	b2FixtureDef()
	{
		shape = nullptr;
		userData = nullptr;
		friction = 0.2f;
		restitution = 0.0f;
		density = 0.0f;
		isSensor = false;
		thickShape = false;
	}
	const b2Shape* shape;
	void* userData;
	float32 friction;
	float32 restitution;
	float32 density;
	bool isSensor;
	b2Filter filter;
	bool thickShape;
};

struct b2FixtureProxy
{
	b2AABB aabb;
	b2Fixture* fixture;
	int32 childIndex;
	int32 proxyId;
};

class b2Fixture
{
public:
	b2Shape::Type GetType() const { return m_shape->GetType(); }
	b2Shape* GetShape() { return m_shape; }
	const b2Shape* GetShape() const { return m_shape; }
	bool IsSensor() const { return m_isSensor; }
	bool IsThickShape() const { return m_isThickShape; }
	const b2Filter& GetFilterData() const { return m_filter; }
	/// b2Fixture.cpp:180-257: the next Collide filters the fixture's contacts again / TOI candidacy is re-evaluated
	void SetFilterData(const b2Filter& filter);
	void Refilter();
	void SetSensor(bool sensor);
	void SetThickShape(bool flag);
	b2Body* GetBody() { return m_body; }
	const b2Body* GetBody() const { return m_body; }
	b2Fixture* GetNext() { return m_next; }
	const b2Fixture* GetNext() const { return m_next; }
	void* GetUserData() const { return m_userData; }
	void SetUserData(void* data) { m_userData = data; }
	bool TestPoint(const b2Vec2& p) const;
	void GetMassData(b2MassData* massData) const { m_shape->ComputeMass(massData, m_density); }
	float32 GetDensity() const { return m_density; }
	float32 GetFriction() const { return m_friction; }
	float32 GetRestitution() const { return m_restitution; }
	/// b2Fixture.h:306-334: the density takes effect at the next b2Body::ResetMassData, friction and restitution in contacts
	/// created from now on (existing contacts keep their mixture)
	void SetDensity(float32 density);
	void SetFriction(float32 friction);
	void SetRestitution(float32 restitution);
	/// b2Fixture.h:296-299
	bool RayCast(b2RayCastOutput* output, const b2RayCastInput& input, int32 childIndex) const;
	/// Fat AABB of the fixture's proxy as the device broad-phase holds it.
	const b2AABB& GetAABB(int32 childIndex) const;
	/// Device id (index into the proxy arrays in HBM).
	int32 GetDeviceId() const { return m_id; }

protected:
	friend class b2Body;
	friend class b2World;
	friend class b2Contact;

	b2Fixture() : m_density(0.0f), m_next(nullptr), m_body(nullptr), m_shape(nullptr), m_friction(0.0f),
		m_restitution(0.0f), m_isSensor(false), m_isThickShape(false), m_userData(nullptr), m_id(-1), m_childCount(1) {}

	float32 m_density;
	b2Fixture* m_next;
	b2Body* m_body;
	b2Shape* m_shape;
	float32 m_friction;
	float32 m_restitution;
	b2Filter m_filter;
	bool m_isSensor;
	bool m_isThickShape;
	void* m_userData;
	int32 m_id;         // device id of child 0; a chain's children follow consecutively
	int32 m_childCount;
	mutable b2AABB m_aabbCache;
};

#endif
