// Drop-in header: line segment shape with optional ghost vertices
// (reference: Box2D/Collision/Shapes/b2EdgeShape.h).
#ifndef B2_EDGE_SHAPE_H
#define B2_EDGE_SHAPE_H

#include "Box2D/Collision/Shapes/b2Shape.h"

class b2EdgeShape : public b2Shape
{
public:
	b2EdgeShape()
	{
		m_type = e_edge;
		m_radius = b2_polygonRadius;
		m_vertex0.SetZero();
		m_vertex1.SetZero();
		m_vertex2.SetZero();
		m_vertex3.SetZero();
		m_hasVertex0 = false;
		m_hasVertex3 = false;
	}
	void Set(const b2Vec2& v1, const b2Vec2& v2)
	{
		m_vertex1 = v1;
		m_vertex2 = v2;
		m_hasVertex0 = false;
		m_hasVertex3 = false;
	}
	b2Shape* Clone(b2BlockAllocator* allocator) const override;
	int32 GetChildCount() const override { return 1; }
	bool TestPoint(const b2Transform& transform, const b2Vec2& p) const override;
	bool RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& transform, int32 childIndex) const override;
	void ComputeAABB(b2AABB* aabb, const b2Transform& transform, int32 childIndex) const override;
	void ComputeMass(b2MassData* massData, float32 density) const override;

	b2Vec2 m_vertex1, m_vertex2;
	b2Vec2 m_vertex0, m_vertex3;
	bool m_hasVertex0, m_hasVertex3;
};

#endif
