// Drop-in header: b2Distance (GJK closest points) and b2ShapeCast, with the reference's types
// (Box2D/Collision/b2Distance.h:27-170). Both run on the CPU build of the device headers (csrc/b2d_toi.h - the GJK the TOI
// kernels use - and csrc/b2d_shapecast.h), see host/src/b2_collision.cpp.
#ifndef B2_DISTANCE_H
#define B2_DISTANCE_H

#include "Box2D/Common/b2Math.h"

class b2Shape;

struct b2DistanceProxy
{
	b2DistanceProxy() : m_vertices(nullptr), m_count(0), m_radius(0.0f) {}
	/// child `index` of `shape`; the shape must outlive the proxy
	void Set(const b2Shape* shape, int32 index);
	/// a vertex cloud with a radius; the vertices must outlive the proxy
	void Set(const b2Vec2* vertices, int32 count, float32 radius);
	int32 GetSupport(const b2Vec2& d) const;
	const b2Vec2& GetSupportVertex(const b2Vec2& d) const { return m_vertices[GetSupport(d)]; }
	int32 GetVertexCount() const { return m_count; }
	const b2Vec2& GetVertex(int32 index) const { return m_vertices[index]; }

	b2Vec2 m_buffer[2];
	const b2Vec2* m_vertices;
	int32 m_count;
	float32 m_radius;
};

struct b2SimplexCache
{
	float32 metric;
	uint16 count;
	uint8 indexA[3];
	uint8 indexB[3];
};

struct b2DistanceInput
{
	b2DistanceProxy proxyA;
	b2DistanceProxy proxyB;
	b2Transform transformA;
	b2Transform transformB;
	bool useRadii;
};

struct b2DistanceOutput
{
	b2Vec2 pointA;
	b2Vec2 pointB;
	float32 distance;
	int32 iterations;
};

void b2Distance(b2DistanceOutput* output, b2SimplexCache* cache, const b2DistanceInput* input);

struct b2ShapeCastInput
{
	b2DistanceProxy proxyA;
	b2DistanceProxy proxyB;
	b2Transform transformA;
	b2Transform transformB;
	b2Vec2 translationB;
};

struct b2ShapeCastOutput
{
	b2Vec2 point;
	b2Vec2 normal;
	float32 lambda;
	int32 iterations;
};

bool b2ShapeCast(b2ShapeCastOutput* output, const b2ShapeCastInput* input);

#endif
