// Drop-in header: the statistics face of the broad-phase (reference: Box2D/Collision/b2BroadPhase.h:37-118 - GetProxyCount,
// GetTreeHeight / Balance / Quality, GetFatAABB, TestOverlap). The broad-phase itself runs on the device (a hashed grid
// rebuilt per step over fat AABBs that follow the reference's MoveProxy rule); these calls are answered from b2World's
// shadow tree over those fat AABBs (b2DynamicTree.h), refitted on demand.
#ifndef B2_BROAD_PHASE_H
#define B2_BROAD_PHASE_H

#include "Box2D/Collision/b2Collision.h"

class b2World;

class b2BroadPhase
{
public:
	enum { e_nullProxy = -1 };
	explicit b2BroadPhase(b2World* world) : m_world(world) {}
	int32 GetProxyCount() const;
	int32 GetTreeHeight() const;
	int32 GetTreeBalance() const;
	float32 GetTreeQuality() const;
	/// proxyId = the device id of the fixture (child) as b2Fixture::GetDeviceId() + child index gives it
	const b2AABB& GetFatAABB(int32 proxyId) const;
	bool TestOverlap(int32 proxyIdA, int32 proxyIdB) const { return b2TestOverlap(GetFatAABB(proxyIdA), GetFatAABB(proxyIdB)); }

private:
	b2World* m_world;
	mutable b2AABB m_scratch[2];
	mutable int m_turn = 0;
};

#endif
