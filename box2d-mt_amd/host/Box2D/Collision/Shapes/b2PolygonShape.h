// Drop-in header: convex polygon shape (reference: Box2D/Collision/Shapes/b2PolygonShape.h).
#ifndef B2_POLYGON_SHAPE_H
#define B2_POLYGON_SHAPE_H

#include "Box2D/Collision/Shapes/b2Shape.h"

class b2PolygonShape : public b2Shape
{
public:
	b2PolygonShape()
	{
		m_type = e_polygon;
		m_radius = b2_polygonRadius;
		m_count = 0;
		m_centroid.SetZero();
	}
	b2Shape* Clone(b2BlockAllocator* allocator) const override;
	int32 GetChildCount() const override { return 1; }
	void Set(const b2Vec2* points, int32 count);
	void SetAsBox(float32 hx, float32 hy);
	void SetAsBox(float32 hx, float32 hy, const b2Vec2& center, float32 angle);
	bool TestPoint(const b2Transform& transform, const b2Vec2& p) const override;
	bool RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& transform, int32 childIndex) const override;
	void ComputeAABB(b2AABB* aabb, const b2Transform& transform, int32 childIndex) const override;
	void ComputeMass(b2MassData* massData, float32 density) const override;
	int32 GetVertexCount() const { return m_count; }
	const b2Vec2& GetVertex(int32 index) const { b2Assert(0 <= index && index < m_count); return m_vertices[index]; }
	bool Validate() const;

	b2Vec2 m_centroid;
	b2Vec2 m_vertices[b2_maxPolygonVertices];
	b2Vec2 m_normals[b2_maxPolygonVertices];
	int32 m_count;
};

#endif
