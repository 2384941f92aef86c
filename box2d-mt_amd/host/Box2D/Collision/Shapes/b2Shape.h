// Drop-in Box2D-MT public header (MI355X build): shape base class
// (reference: Box2D/Collision/Shapes/b2Shape.h:28-90).
#ifndef B2_SHAPE_H
#define B2_SHAPE_H

#include "Box2D/Common/b2BlockAllocator.h"
#include "Box2D/Common/b2Math.h"
#include "Box2D/Collision/b2Collision.h"

struct b2MassData
{
	float32 mass;
	b2Vec2 center;
	float32 I;
};

class b2Shape
{
public:
	enum Type { e_circle = 0, e_edge = 1, e_polygon = 2, e_chain = 3, e_typeCount = 4 };

	virtual ~b2Shape() {}
	virtual b2Shape* Clone(b2BlockAllocator* allocator) const = 0;
	Type GetType() const { return m_type; }
	virtual int32 GetChildCount() const = 0;
	virtual bool TestPoint(const b2Transform& xf, const b2Vec2& p) const = 0;
	virtual bool RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& transform, int32 childIndex) const = 0;
	virtual void ComputeAABB(b2AABB* aabb, const b2Transform& xf, int32 childIndex) const = 0;
	virtual void ComputeMass(b2MassData* massData, float32 density) const = 0;

	Type m_type;
	float32 m_radius;
};

#endif
