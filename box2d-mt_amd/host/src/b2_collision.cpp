// Host-side collision API of the drop-in layer (b2Collision.h:229-256, b2Distance.h, b2TimeOfImpact.h of the reference):
// b2CollidePolygons & co., b2Distance, b2ShapeCast, b2TimeOfImpact, b2TestOverlap(shapes), b2AABB::RayCast.
//
// None of them carries an algorithm of its own: each converts its arguments to the records of the device code and calls the
// CPU build of the routine the kernels run - the manifold functions of csrc/b2d_collide.h (k_collide), the GJK and the
// conservative advancement of csrc/b2d_toi.h (the TOI kernels), the shape cast of csrc/b2d_shapecast.h - so user code on the
// host sees exactly what the device computes. (Those headers are pinned bit for bit against the reference build:
// tests/test_device_math_cpu.py, tests/test_host_api.py.)
#include "Box2D/Collision/b2Collision.h"
#include "Box2D/Collision/b2Distance.h"
#include "Box2D/Collision/b2TimeOfImpact.h"
#include "Box2D/Collision/Shapes/b2ChainShape.h"
#include "Box2D/Collision/Shapes/b2CircleShape.h"
#include "Box2D/Collision/Shapes/b2EdgeShape.h"
#include "Box2D/Collision/Shapes/b2PolygonShape.h"

#include "../../csrc/b2d_shapecast.h"

#include <string.h>

// the 152-byte device record of a shape / of child `child` of a chain (b2_shapes.cpp)
void b2ShapeToRecord(const b2Shape* shape, int32 child, void* record152);

namespace
{
inline V2 toV2(const b2Vec2& v) { return v2(v.x, v.y); }
inline b2Vec2 fromV2(V2 v) { return b2Vec2(v.x, v.y); }
inline Xf toXf(const b2Transform& t)
{
	Xf xf;
	xf.p = toV2(t.p);
	xf.q.s = t.q.s;
	xf.q.c = t.q.c;
	return xf;
}

ShapeRec recordOf(const b2Shape* shape, int32 child = 0)
{
	ShapeRec r;
	b2ShapeToRecord(shape, child, &r);
	return r;
}

void manifoldOut(b2Manifold* out, const Manifold& m)
{
	out->pointCount = m.pointCount;
	// (the routines name the manifold type before they know whether any point survives, like the reference's)
	out->type = (b2Manifold::Type)m.type;
	if (m.pointCount == 0) return;
	out->localNormal = fromV2(m.localNormal);
	out->localPoint = fromV2(m.localPoint);
	for (int32 k = 0; k < m.pointCount; ++k)
	{
		out->points[k].localPoint = fromV2(m.p[k]);
		out->points[k].id.key = m.id[k];
	}
}

template <class A, class B> void collidePair(b2Manifold* out, const A* shapeA, const b2Transform& xfA, const B* shapeB, const b2Transform& xfB)
{
	const ShapeRec a = recordOf(shapeA), b = recordOf(shapeB);
	Manifold m;
	memset(&m, 0, sizeof(m));
	b2dEvaluate(&m, &a, toXf(xfA), &b, toXf(xfB));
	manifoldOut(out, m);
}

// a proxy's vertices as the device code wants them (V2 has the layout of b2Vec2)
GjkProxy proxyOf(const b2DistanceProxy& p)
{
	static_assert(sizeof(V2) == sizeof(b2Vec2), "vertex layout");
	GjkProxy g;
	g.verts = reinterpret_cast<const V2*>(p.m_vertices);
	g.count = p.m_count;
	g.radius = p.m_radius;
	return g;
}

Sweep sweepOf(const b2Sweep& s)
{
	Sweep r;
	r.localCenter = toV2(s.localCenter);
	r.c0 = toV2(s.c0);
	r.c = toV2(s.c);
	r.a0 = s.a0;
	r.a = s.a;
	r.alpha0 = s.alpha0;
	return r;
}
} // namespace

void b2CollideCircles(b2Manifold* manifold, const b2CircleShape* circleA, const b2Transform& xfA, const b2CircleShape* circleB, const b2Transform& xfB)
{
	collidePair(manifold, circleA, xfA, circleB, xfB);
}

void b2CollidePolygonAndCircle(b2Manifold* manifold, const b2PolygonShape* polygonA, const b2Transform& xfA, const b2CircleShape* circleB, const b2Transform& xfB)
{
	collidePair(manifold, polygonA, xfA, circleB, xfB);
}

void b2CollidePolygons(b2Manifold* manifold, const b2PolygonShape* polygonA, const b2Transform& xfA, const b2PolygonShape* polygonB, const b2Transform& xfB)
{
	collidePair(manifold, polygonA, xfA, polygonB, xfB);
}

void b2CollideEdgeAndCircle(b2Manifold* manifold, const b2EdgeShape* edgeA, const b2Transform& xfA, const b2CircleShape* circleB, const b2Transform& xfB)
{
	collidePair(manifold, edgeA, xfA, circleB, xfB);
}

void b2CollideEdgeAndPolygon(b2Manifold* manifold, const b2EdgeShape* edgeA, const b2Transform& xfA, const b2PolygonShape* polygonB, const b2Transform& xfB)
{
	collidePair(manifold, edgeA, xfA, polygonB, xfB);
}

int32 b2ClipSegmentToLine(b2ClipVertex vOut[2], const b2ClipVertex vIn[2], const b2Vec2& normal, float32 offset, int32 vertexIndexA)
{
	ClipVertex in[2], out[2];
	for (int32 k = 0; k < 2; ++k)
	{
		in[k].v = toV2(vIn[k].v);
		in[k].id = vIn[k].id.key;
	}
	const int32 n = b2dClipSegmentToLine(out, in, toV2(normal), offset, vertexIndexA);
	for (int32 k = 0; k < n; ++k)
	{
		vOut[k].v = fromV2(out[k].v);
		vOut[k].id.key = out[k].id;
	}
	return n;
}

// ---- distance proxies ---------------------------------------------------------------------------------------------------------
// b2DistanceProxy::Set (b2Distance.cpp:31-99): a circle is one vertex, an edge two, a polygon its vertices, a chain child its
// two end points (copied: the chain's own array may wrap at the last child)
void b2DistanceProxy::Set(const b2Shape* shape, int32 index)
{
	switch (shape->GetType())
	{
	case b2Shape::e_circle:
	{
		const b2CircleShape* circle = static_cast<const b2CircleShape*>(shape);
		m_vertices = &circle->m_p;
		m_count = 1;
		break;
	}
	case b2Shape::e_polygon:
	{
		const b2PolygonShape* polygon = static_cast<const b2PolygonShape*>(shape);
		m_vertices = polygon->m_vertices;
		m_count = polygon->m_count;
		break;
	}
	case b2Shape::e_chain:
	{
		const b2ChainShape* chain = static_cast<const b2ChainShape*>(shape);
		m_buffer[0] = chain->m_vertices[index];
		m_buffer[1] = chain->m_vertices[index + 1 < chain->m_count ? index + 1 : 0];
		m_vertices = m_buffer;
		m_count = 2;
		break;
	}
	case b2Shape::e_edge:
	{
		const b2EdgeShape* edge = static_cast<const b2EdgeShape*>(shape);
		m_vertices = &edge->m_vertex1; // (m_vertex1, m_vertex2 are adjacent members)
		m_count = 2;
		break;
	}
	default:
		m_vertices = nullptr;
		m_count = 0;
		break;
	}
	m_radius = shape->m_radius;
}

void b2DistanceProxy::Set(const b2Vec2* vertices, int32 count, float32 radius)
{
	m_vertices = vertices;
	m_count = count;
	m_radius = radius;
}

int32 b2DistanceProxy::GetSupport(const b2Vec2& d) const
{
	return b2dSupport(proxyOf(*this), toV2(d));
}

void b2Distance(b2DistanceOutput* output, b2SimplexCache* cache, const b2DistanceInput* input)
{
	GjkCache c;
	c.metric = cache->metric;
	c.count = cache->count;
	for (int32 k = 0; k < 3; ++k)
	{
		c.indexA[k] = cache->indexA[k];
		c.indexB[k] = cache->indexB[k];
	}
	GjkOutput o;
	b2dDistance(o, c, proxyOf(input->proxyA), toXf(input->transformA), proxyOf(input->proxyB), toXf(input->transformB), input->useRadii);
	output->pointA = fromV2(o.pointA);
	output->pointB = fromV2(o.pointB);
	output->distance = o.distance;
	output->iterations = o.iterations;
	cache->metric = c.metric;
	cache->count = (uint16)c.count;
	for (int32 k = 0; k < 3; ++k)
	{
		cache->indexA[k] = (uint8)c.indexA[k];
		cache->indexB[k] = (uint8)c.indexB[k];
	}
}

bool b2ShapeCast(b2ShapeCastOutput* output, const b2ShapeCastInput* input)
{
	ShapeCastResult r;
	const bool hit = b2dShapeCast(&r, proxyOf(input->proxyA), toXf(input->transformA), proxyOf(input->proxyB), toXf(input->transformB),
		toV2(input->translationB));
	output->point = fromV2(r.point);
	output->normal = fromV2(r.normal);
	output->lambda = r.lambda;
	output->iterations = r.iterations;
	return hit;
}

void b2TimeOfImpact(b2TOIOutput* output, const b2TOIInput* input)
{
	float t = input->tMax;
	const int state = b2dTimeOfImpact(&t, proxyOf(input->proxyA), sweepOf(input->sweepA), proxyOf(input->proxyB), sweepOf(input->sweepB), input->tMax);
	output->state = (b2TOIOutput::State)state; // (TOI_UNKNOWN .. TOI_SEPARATED in b2TOIOutput::State's order)
	output->t = t;
}

bool b2TestOverlap(const b2Shape* shapeA, int32 indexA, const b2Shape* shapeB, int32 indexB, const b2Transform& xfA, const b2Transform& xfB)
{
	b2DistanceInput input;
	input.proxyA.Set(shapeA, indexA);
	input.proxyB.Set(shapeB, indexB);
	input.transformA = xfA;
	input.transformB = xfB;
	input.useRadii = true;
	b2SimplexCache cache;
	memset(&cache, 0, sizeof(cache));
	b2DistanceOutput output;
	b2Distance(&output, &cache, &input);
	return output.distance < 10.0f * b2_epsilon;
}

// The segment against the box, axis by axis: the entry parameter is the largest of the near-plane crossings (its axis gives
// the normal), the exit parameter the smallest of the far-plane crossings (b2Collision.cpp:133-198).
bool b2AABB::RayCast(b2RayCastOutput* output, const b2RayCastInput& input) const
{
	const float32 from[2] = { input.p1.x, input.p1.y };
	const float32 step[2] = { input.p2.x - input.p1.x, input.p2.y - input.p1.y };
	const float32 lo[2] = { lowerBound.x, lowerBound.y }, hi[2] = { upperBound.x, upperBound.y };
	float32 enter = -b2_maxFloat, leave = b2_maxFloat;
	float32 face[2] = { 0.0f, 0.0f };
	for (int32 axis = 0; axis < 2; ++axis)
	{
		if (b2Abs(step[axis]) < b2_epsilon)
		{
			if (from[axis] < lo[axis] || hi[axis] < from[axis]) return false; // runs beside the slab
			continue;
		}
		const float32 inv = 1.0f / step[axis];
		float32 nearT = (lo[axis] - from[axis]) * inv;
		float32 farT = (hi[axis] - from[axis]) * inv;
		float32 sign = -1.0f;
		if (nearT > farT)
		{
			const float32 t = nearT;
			nearT = farT;
			farT = t;
			sign = 1.0f;
		}
		if (nearT > enter)
		{
			face[0] = face[1] = 0.0f;
			face[axis] = sign;
			enter = nearT;
		}
		leave = b2Min(leave, farT);
		if (enter > leave) return false;
	}
	if (enter < 0.0f || input.maxFraction < enter) return false; // starts inside, or ends before the box
	output->fraction = enter;
	output->normal.Set(face[0], face[1]);
	return true;
}
