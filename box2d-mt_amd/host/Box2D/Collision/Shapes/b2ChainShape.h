// Drop-in header: chain shape, a sequence of segments with ghost-vertex connectivity
// (reference: Box2D/Collision/Shapes/b2ChainShape.h:32-105). Every child segment becomes one device record
// (type "chain child": an edge for the narrow phase, an AABB without the radius for the broad-phase).
#ifndef B2_CHAIN_SHAPE_H
#define B2_CHAIN_SHAPE_H

#include "Box2D/Collision/Shapes/b2Shape.h"

class b2EdgeShape;

class b2ChainShape : public b2Shape
{
public:
	b2ChainShape()
	{
		m_type = e_chain;
		m_radius = b2_polygonRadius;
		m_vertices = nullptr;
		m_count = 0;
		m_prevVertex.SetZero();
		m_nextVertex.SetZero();
		m_hasPrevVertex = false;
		m_hasNextVertex = false;
	}
	~b2ChainShape();
	void Clear();
	void CreateLoop(const b2Vec2* vertices, int32 count);
	void CreateChain(const b2Vec2* vertices, int32 count);
	void SetPrevVertex(const b2Vec2& prevVertex);
	void SetNextVertex(const b2Vec2& nextVertex);
	b2Shape* Clone(b2BlockAllocator* allocator) const override;
	int32 GetChildCount() const override { return m_count - 1; }
	void GetChildEdge(b2EdgeShape* edge, int32 index) const;
	bool TestPoint(const b2Transform& transform, const b2Vec2& p) const override;
	bool RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& transform, int32 childIndex) const override;
	void ComputeAABB(b2AABB* aabb, const b2Transform& transform, int32 childIndex) const override;
	void ComputeMass(b2MassData* massData, float32 density) const override;

	b2Vec2* m_vertices;
	int32 m_count;
	b2Vec2 m_prevVertex, m_nextVertex;
	bool m_hasPrevVertex, m_hasNextVertex;

private:
	void Keep(const b2Vec2* vertices, int32 count, bool closed);
};

#endif
