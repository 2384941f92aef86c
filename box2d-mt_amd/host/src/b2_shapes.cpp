// Host-side shape classes of the drop-in API as VIEWS of the 152-byte shape record the device works on.
//
// b2CircleShape / b2EdgeShape / b2PolygonShape / b2ChainShape keep the reference's public members (user code reads and
// writes them: Testbed scenes fill m_vertex0 / m_hasVertex0, read m_vertices, ...), but none of them carries geometry code
// of its own: every query packs the members into a ShapeRec and calls the ONE geometry module of the tree,
// csrc/b2d_shape_geom.h + b2dShapeAABB (the header the kernels and the C ABI use), so that a fixture's mass, its first AABB
// and a ray cast against it are the same floats on the host, in the C ABI and on the device.
// Reference semantics: Box2D/Collision/Shapes/b2{Circle,Edge,Polygon,Chain}Shape.{h,cpp} (cited in b2d_shape_geom.h).
#include "Box2D/Collision/Shapes/b2ChainShape.h"
#include "Box2D/Collision/Shapes/b2CircleShape.h"
#include "Box2D/Collision/Shapes/b2EdgeShape.h"
#include "Box2D/Collision/Shapes/b2PolygonShape.h"

#include "../../csrc/b2d_shape_geom.h"

#include <new>
#include <string.h>

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

ShapeRec blankRecord(int type, float radius)
{
	ShapeRec r;
	memset(&r, 0, sizeof(r));
	r.type = type;
	r.radius = radius;
	return r;
}

ShapeRec recordOf(const b2CircleShape& c)
{
	ShapeRec r = blankRecord(B2D_SHAPE_CIRCLE, c.m_radius);
	r.verts[0] = toV2(c.m_p);
	return r;
}

// segment a-b with optional ghost vertices; `type` is B2D_SHAPE_EDGE or B2D_SHAPE_CHAIN (a chain child: no radius in its AABB)
ShapeRec segmentRecord(int type, float radius, const b2Vec2& a, const b2Vec2& b, const b2Vec2& before, bool hasBefore,
	const b2Vec2& after, bool hasAfter)
{
	ShapeRec r = blankRecord(type, radius);
	r.verts[0] = toV2(a);
	r.verts[1] = toV2(b);
	r.verts[2] = toV2(before);
	r.verts[3] = toV2(after);
	r.count = (hasBefore ? 1 : 0) | (hasAfter ? 2 : 0);
	return r;
}

ShapeRec recordOf(const b2EdgeShape& e)
{
	return segmentRecord(B2D_SHAPE_EDGE, e.m_radius, e.m_vertex1, e.m_vertex2, e.m_vertex0, e.m_hasVertex0, e.m_vertex3, e.m_hasVertex3);
}

ShapeRec recordOf(const b2PolygonShape& p)
{
	ShapeRec r = blankRecord(B2D_SHAPE_POLYGON, p.m_radius);
	r.count = p.m_count;
	r.centroid = toV2(p.m_centroid);
	for (int32 i = 0; i < p.m_count; ++i)
	{
		r.verts[i] = toV2(p.m_vertices[i]);
		r.normals[i] = toV2(p.m_normals[i]);
	}
	return r;
}

void adopt(b2PolygonShape& p, const ShapeRec& r)
{
	p.m_count = r.count;
	p.m_centroid = fromV2(r.centroid);
	for (int32 i = 0; i < r.count; ++i)
	{
		p.m_vertices[i] = fromV2(r.verts[i]);
		p.m_normals[i] = fromV2(r.normals[i]);
	}
}

void boxOf(b2AABB* out, const ShapeRec& r, const b2Transform& t)
{
	const AABB box = b2dShapeAABB(&r, toXf(t));
	out->lowerBound = fromV2(box.lo);
	out->upperBound = fromV2(box.hi);
}

void massOf(b2MassData* out, const ShapeRec& r, float32 density)
{
	const MassProps mp = b2dShapeMass(&r, density);
	out->mass = mp.mass;
	out->center = fromV2(mp.center);
	out->I = mp.inertia;
}

bool castAt(b2RayCastOutput* out, const b2RayCastInput& in, const ShapeRec& r, const b2Transform& t)
{
	RayHit hit;
	if (!b2dShapeRayCast(&r, toXf(t), toV2(in.p1), toV2(in.p2), in.maxFraction, &hit)) return false;
	out->fraction = hit.fraction;
	out->normal = fromV2(hit.normal);
	return true;
}

template <class Shape> b2Shape* cloneInto(b2BlockAllocator* allocator, const Shape& original)
{
	Shape* copy = new (allocator->Allocate(sizeof(Shape))) Shape;
	*copy = original;
	return copy;
}
} // namespace

// ---- the record of any shape / child (used by b2Body::CreateFixture to hand the geometry to the C ABI) -------------------
void b2ShapeToRecord(const b2Shape* shape, int32 child, void* record152)
{
	ShapeRec r = blankRecord(B2D_SHAPE_CIRCLE, 0.0f);
	switch (shape->GetType())
	{
	case b2Shape::e_circle: r = recordOf(*static_cast<const b2CircleShape*>(shape)); break;
	case b2Shape::e_edge: r = recordOf(*static_cast<const b2EdgeShape*>(shape)); break;
	case b2Shape::e_polygon: r = recordOf(*static_cast<const b2PolygonShape*>(shape)); break;
	case b2Shape::e_chain:
	{
		b2EdgeShape link;
		static_cast<const b2ChainShape*>(shape)->GetChildEdge(&link, child);
		r = recordOf(link);
		r.type = B2D_SHAPE_CHAIN;
		break;
	}
	default: break;
	}
	static_assert(sizeof(ShapeRec) == 152, "shape record layout");
	memcpy(record152, &r, sizeof(r));
}

// ---- circle -----------------------------------------------------------------------------------------------------------------
b2Shape* b2CircleShape::Clone(b2BlockAllocator* allocator) const { return cloneInto(allocator, *this); }
bool b2CircleShape::TestPoint(const b2Transform& t, const b2Vec2& p) const
{
	const ShapeRec r = recordOf(*this);
	return b2dShapeTestPoint(&r, toXf(t), toV2(p));
}
bool b2CircleShape::RayCast(b2RayCastOutput* out, const b2RayCastInput& in, const b2Transform& t, int32) const { return castAt(out, in, recordOf(*this), t); }
void b2CircleShape::ComputeAABB(b2AABB* out, const b2Transform& t, int32) const { boxOf(out, recordOf(*this), t); }
void b2CircleShape::ComputeMass(b2MassData* out, float32 density) const { massOf(out, recordOf(*this), density); }

// ---- edge -------------------------------------------------------------------------------------------------------------------
b2Shape* b2EdgeShape::Clone(b2BlockAllocator* allocator) const { return cloneInto(allocator, *this); }
bool b2EdgeShape::TestPoint(const b2Transform&, const b2Vec2&) const { return false; }
bool b2EdgeShape::RayCast(b2RayCastOutput* out, const b2RayCastInput& in, const b2Transform& t, int32) const { return castAt(out, in, recordOf(*this), t); }
void b2EdgeShape::ComputeAABB(b2AABB* out, const b2Transform& t, int32) const { boxOf(out, recordOf(*this), t); }
void b2EdgeShape::ComputeMass(b2MassData* out, float32 density) const { massOf(out, recordOf(*this), density); }

// ---- polygon ----------------------------------------------------------------------------------------------------------------
b2Shape* b2PolygonShape::Clone(b2BlockAllocator* allocator) const { return cloneInto(allocator, *this); }

void b2PolygonShape::SetAsBox(float32 hx, float32 hy)
{
	ShapeRec r;
	b2dPolygonBox(&r, hx, hy);
	adopt(*this, r);
}

void b2PolygonShape::SetAsBox(float32 hx, float32 hy, const b2Vec2& center, float32 angle)
{
	ShapeRec r;
	b2dPolygonBoxAt(&r, hx, hy, toV2(center), angle);
	adopt(*this, r);
}

void b2PolygonShape::Set(const b2Vec2* points, int32 count)
{
	V2 cloud[b2_maxPolygonVertices];
	const int32 n = count < (int32)b2_maxPolygonVertices ? count : (int32)b2_maxPolygonVertices;
	for (int32 i = 0; i < n; ++i) cloud[i] = toV2(points[i]);
	ShapeRec r;
	b2dPolygonFromPoints(&r, cloud, count < 3 ? count : n);
	adopt(*this, r);
}

bool b2PolygonShape::TestPoint(const b2Transform& t, const b2Vec2& p) const
{
	const ShapeRec r = recordOf(*this);
	return b2dShapeTestPoint(&r, toXf(t), toV2(p));
}
bool b2PolygonShape::RayCast(b2RayCastOutput* out, const b2RayCastInput& in, const b2Transform& t, int32) const { return castAt(out, in, recordOf(*this), t); }
void b2PolygonShape::ComputeAABB(b2AABB* out, const b2Transform& t, int32) const { boxOf(out, recordOf(*this), t); }
void b2PolygonShape::ComputeMass(b2MassData* out, float32 density) const { massOf(out, recordOf(*this), density); }
bool b2PolygonShape::Validate() const
{
	const ShapeRec r = recordOf(*this);
	return b2dPolygonConvex(&r);
}

// ---- chain ------------------------------------------------------------------------------------------------------------------
// A chain is a list of vertices; child i is the segment (i, i + 1) with its neighbours as ghost vertices
// (b2ChainShape.cpp:114-147). On the device every child is a fixture-like record of type B2D_SHAPE_CHAIN: an edge for the
// narrow phase and the TOI proxy, an AABB without the radius for the broad-phase (b2ChainShape.cpp:174-189).
b2ChainShape::~b2ChainShape() { Clear(); }

void b2ChainShape::Clear()
{
	b2Free(m_vertices);
	m_vertices = nullptr;
	m_count = 0;
}

void b2ChainShape::Keep(const b2Vec2* vertices, int32 count, bool closed)
{
	m_count = closed ? count + 1 : count;
	m_vertices = (b2Vec2*)b2Alloc(m_count * (int32)sizeof(b2Vec2));
	for (int32 i = 0; i < count; ++i) m_vertices[i] = vertices[i];
	if (closed) m_vertices[count] = m_vertices[0];
	m_hasPrevVertex = m_hasNextVertex = closed;
	if (closed)
	{
		m_prevVertex = m_vertices[m_count - 2];
		m_nextVertex = m_vertices[1];
	}
	else
	{
		m_prevVertex.SetZero();
		m_nextVertex.SetZero();
	}
}

void b2ChainShape::CreateLoop(const b2Vec2* vertices, int32 count)
{
	if (count < 3 || m_vertices != nullptr) return;
	Keep(vertices, count, true);
}

void b2ChainShape::CreateChain(const b2Vec2* vertices, int32 count)
{
	if (count < 2 || m_vertices != nullptr) return;
	Keep(vertices, count, false);
}

void b2ChainShape::SetPrevVertex(const b2Vec2& v)
{
	m_prevVertex = v;
	m_hasPrevVertex = true;
}

void b2ChainShape::SetNextVertex(const b2Vec2& v)
{
	m_nextVertex = v;
	m_hasNextVertex = true;
}

b2Shape* b2ChainShape::Clone(b2BlockAllocator* allocator) const
{
	b2ChainShape* copy = new (allocator->Allocate(sizeof(b2ChainShape))) b2ChainShape;
	copy->m_radius = m_radius;
	if (m_count > 0) copy->Keep(m_vertices, m_count, false);
	copy->m_prevVertex = m_prevVertex;
	copy->m_nextVertex = m_nextVertex;
	copy->m_hasPrevVertex = m_hasPrevVertex;
	copy->m_hasNextVertex = m_hasNextVertex;
	return copy;
}

void b2ChainShape::GetChildEdge(b2EdgeShape* edge, int32 index) const
{
	const int32 last = m_count - 2; // index of the last child
	edge->m_type = b2Shape::e_edge;
	edge->m_radius = m_radius;
	edge->m_vertex1 = m_vertices[index];
	edge->m_vertex2 = m_vertices[index + 1];
	const bool inner0 = index > 0, inner3 = index < last;
	edge->m_vertex0 = inner0 ? m_vertices[index - 1] : m_prevVertex;
	edge->m_hasVertex0 = inner0 || m_hasPrevVertex;
	edge->m_vertex3 = inner3 ? m_vertices[index + 2] : m_nextVertex;
	edge->m_hasVertex3 = inner3 || m_hasNextVertex;
}

bool b2ChainShape::TestPoint(const b2Transform&, const b2Vec2&) const { return false; }

// (the reference wraps the second index at m_count - dead code for a valid child, kept for the same answers on any index)
static ShapeRec chainLink(const b2ChainShape& c, int32 child)
{
	const int32 second = child + 1 == c.m_count ? 0 : child + 1;
	return segmentRecord(B2D_SHAPE_CHAIN, c.m_radius, c.m_vertices[child], c.m_vertices[second], b2Vec2_zero, false, b2Vec2_zero, false);
}

bool b2ChainShape::RayCast(b2RayCastOutput* out, const b2RayCastInput& in, const b2Transform& t, int32 child) const
{
	return castAt(out, in, chainLink(*this, child), t);
}

void b2ChainShape::ComputeAABB(b2AABB* out, const b2Transform& t, int32 child) const { boxOf(out, chainLink(*this, child), t); }

void b2ChainShape::ComputeMass(b2MassData* out, float32) const
{
	out->mass = 0.0f;
	out->center.SetZero();
	out->I = 0.0f;
}
