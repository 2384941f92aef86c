// Drop-in header: circle shape (reference: Box2D/Collision/Shapes/b2CircleShape.h).
#ifndef B2_CIRCLE_SHAPE_H
#define B2_CIRCLE_SHAPE_H

#include "Box2D/Collision/Shapes/b2Shape.h"

class b2CircleShape : public b2Shape
{
public:
	b2CircleShape()
	{
		m_type = e_circle;
		m_radius = 0.0f;
		m_p.SetZero();
	}
	b2Shape* Clone(b2BlockAllocator* allocator) const override;
	int32 GetChildCount() const override { return 1; }
	bool TestPoint(const b2Transform& transform, const b2Vec2& p) const override;
	bool RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& transform, int32 childIndex) const override;
	void ComputeAABB(b2AABB* aabb, const b2Transform& transform, int32 childIndex) const override;
	void ComputeMass(b2MassData* massData, float32 density) const override;
	int32 GetSupport(const b2Vec2& d) const { B2_NOT_USED(d); return 0; }
	const b2Vec2& GetSupportVertex(const b2Vec2& d) const { B2_NOT_USED(d); return m_p; }
	int32 GetVertexCount() const { return 1; }
	const b2Vec2& GetVertex(int32 index) const { B2_NOT_USED(index); return m_p; }

	b2Vec2 m_p;
};

#endif
