// Host-side shape classes of the drop-in API. Shapes are user-constructed value objects; what the
// device needs from them (vertices, normals, centroid, radius) is copied into a b2hip_shape when a
// fixture is created. The geometry set-up (hull, normals, centroid, AABB, mass) restates the
// reference's arithmetic in the same operand order so that mass data and initial AABBs match
// bit for bit: b2PolygonShape.cpp:30-67 (SetAsBox), :72-118 (centroid), :120-250 (Set),
// :340-357 (AABB), :359-440 (mass); b2CircleShape.cpp:83-100; b2EdgeShape.cpp:116-138.
#include "Box2D/Collision/Shapes/b2CircleShape.h"
#include "Box2D/Collision/Shapes/b2EdgeShape.h"
#include "Box2D/Collision/Shapes/b2PolygonShape.h"

#include <new>

// ---- circle -------------------------------------------------------------------------------------
b2Shape* b2CircleShape::Clone(b2BlockAllocator* allocator) const
{
	void* mem = allocator->Allocate(sizeof(b2CircleShape));
	b2CircleShape* clone = new (mem) b2CircleShape;
	*clone = *this;
	return clone;
}

bool b2CircleShape::TestPoint(const b2Transform& transform, const b2Vec2& p) const
{
	b2Vec2 center = transform.p + b2Mul(transform.q, m_p);
	b2Vec2 d = p - center;
	return b2Dot(d, d) <= m_radius * m_radius;
}

bool b2CircleShape::RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& transform, int32 childIndex) const
{
	B2_NOT_USED(childIndex);
	b2Vec2 position = transform.p + b2Mul(transform.q, m_p);
	b2Vec2 s = input.p1 - position;
	float32 b = b2Dot(s, s) - m_radius * m_radius;
	b2Vec2 r = input.p2 - input.p1;
	float32 c = b2Dot(s, r);
	float32 rr = b2Dot(r, r);
	float32 sigma = c * c - rr * b;
	if (sigma < 0.0f || rr < b2_epsilon) return false;
	float32 a = -(c + b2Sqrt(sigma));
	if (0.0f <= a && a <= input.maxFraction * rr)
	{
		a /= rr;
		output->fraction = a;
		output->normal = s + a * r;
		output->normal.Normalize();
		return true;
	}
	return false;
}

void b2CircleShape::ComputeAABB(b2AABB* aabb, const b2Transform& transform, int32 childIndex) const
{
	B2_NOT_USED(childIndex);
	b2Vec2 p = transform.p + b2Mul(transform.q, m_p);
	aabb->lowerBound.Set(p.x - m_radius, p.y - m_radius);
	aabb->upperBound.Set(p.x + m_radius, p.y + m_radius);
}

void b2CircleShape::ComputeMass(b2MassData* massData, float32 density) const
{
	massData->mass = density * b2_pi * m_radius * m_radius;
	massData->center = m_p;
	massData->I = massData->mass * (0.5f * m_radius * m_radius + b2Dot(m_p, m_p));
}

// ---- edge ---------------------------------------------------------------------------------------
b2Shape* b2EdgeShape::Clone(b2BlockAllocator* allocator) const
{
	void* mem = allocator->Allocate(sizeof(b2EdgeShape));
	b2EdgeShape* clone = new (mem) b2EdgeShape;
	*clone = *this;
	return clone;
}

bool b2EdgeShape::TestPoint(const b2Transform& transform, const b2Vec2& p) const
{
	B2_NOT_USED(transform);
	B2_NOT_USED(p);
	return false;
}

bool b2EdgeShape::RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& xf, int32 childIndex) const
{
	B2_NOT_USED(childIndex);
	// ray in the edge's frame, intersect with the supporting line, then clamp to the segment
	b2Vec2 p1 = b2MulT(xf.q, input.p1 - xf.p);
	b2Vec2 p2 = b2MulT(xf.q, input.p2 - xf.p);
	b2Vec2 d = p2 - p1;
	b2Vec2 v1 = m_vertex1, v2 = m_vertex2;
	b2Vec2 e = v2 - v1;
	b2Vec2 normal(e.y, -e.x);
	normal.Normalize();
	float32 numerator = b2Dot(normal, v1 - p1);
	float32 denominator = b2Dot(normal, d);
	if (denominator == 0.0f) return false;
	float32 t = numerator / denominator;
	if (t < 0.0f || input.maxFraction < t) return false;
	b2Vec2 q = p1 + t * d;
	float32 rr = b2Dot(e, e);
	if (rr == 0.0f) return false;
	float32 s = b2Dot(q - v1, e) / rr;
	if (s < 0.0f || 1.0f < s) return false;
	output->fraction = t;
	output->normal = numerator > 0.0f ? -b2Mul(xf.q, normal) : b2Mul(xf.q, normal);
	return true;
}

void b2EdgeShape::ComputeAABB(b2AABB* aabb, const b2Transform& xf, int32 childIndex) const
{
	B2_NOT_USED(childIndex);
	b2Vec2 v1 = b2Mul(xf, m_vertex1);
	b2Vec2 v2 = b2Mul(xf, m_vertex2);
	b2Vec2 lower = b2Min(v1, v2);
	b2Vec2 upper = b2Max(v1, v2);
	b2Vec2 r(m_radius, m_radius);
	aabb->lowerBound = lower - r;
	aabb->upperBound = upper + r;
}

void b2EdgeShape::ComputeMass(b2MassData* massData, float32 density) const
{
	B2_NOT_USED(density);
	massData->mass = 0.0f;
	massData->center = 0.5f * (m_vertex1 + m_vertex2);
	massData->I = 0.0f;
}

// ---- polygon ------------------------------------------------------------------------------------
b2Shape* b2PolygonShape::Clone(b2BlockAllocator* allocator) const
{
	void* mem = allocator->Allocate(sizeof(b2PolygonShape));
	b2PolygonShape* clone = new (mem) b2PolygonShape;
	*clone = *this;
	return clone;
}

void b2PolygonShape::SetAsBox(float32 hx, float32 hy)
{
	m_count = 4;
	m_vertices[0].Set(-hx, -hy);
	m_vertices[1].Set(hx, -hy);
	m_vertices[2].Set(hx, hy);
	m_vertices[3].Set(-hx, hy);
	m_normals[0].Set(0.0f, -1.0f);
	m_normals[1].Set(1.0f, 0.0f);
	m_normals[2].Set(0.0f, 1.0f);
	m_normals[3].Set(-1.0f, 0.0f);
	m_centroid.SetZero();
}

void b2PolygonShape::SetAsBox(float32 hx, float32 hy, const b2Vec2& center, float32 angle)
{
	SetAsBox(hx, hy);
	m_centroid = center;
	b2Transform xf;
	xf.p = center;
	xf.q.Set(angle);
	for (int32 i = 0; i < m_count; ++i)
	{
		m_vertices[i] = b2Mul(xf, m_vertices[i]);
		m_normals[i] = b2Mul(xf.q, m_normals[i]);
	}
}

static b2Vec2 PolygonCentroid(const b2Vec2* vs, int32 count)
{
	b2Vec2 c(0.0f, 0.0f);
	float32 area = 0.0f;
	const b2Vec2 pRef(0.0f, 0.0f);
	const float32 inv3 = 1.0f / 3.0f;
	for (int32 i = 0; i < count; ++i)
	{
		b2Vec2 p1 = pRef;
		b2Vec2 p2 = vs[i];
		b2Vec2 p3 = i + 1 < count ? vs[i + 1] : vs[0];
		b2Vec2 e1 = p2 - p1;
		b2Vec2 e2 = p3 - p1;
		float32 D = b2Cross(e1, e2);
		float32 triangleArea = 0.5f * D;
		area += triangleArea;
		c += triangleArea * inv3 * (p1 + p2 + p3);
	}
	c *= 1.0f / area;
	return c;
}

void b2PolygonShape::Set(const b2Vec2* vertices, int32 count)
{
	if (count < 3)
	{
		SetAsBox(1.0f, 1.0f);
		return;
	}
	int32 n = b2Min(count, (int32)b2_maxPolygonVertices);

	// weld near-duplicates
	b2Vec2 ps[b2_maxPolygonVertices];
	int32 kept = 0;
	const float32 weldSq = (0.5f * b2_linearSlop) * (0.5f * b2_linearSlop);
	for (int32 i = 0; i < n; ++i)
	{
		bool unique = true;
		for (int32 j = 0; j < kept && unique; ++j)
		{
			if (b2DistanceSquared(vertices[i], ps[j]) < weldSq) unique = false;
		}
		if (unique) ps[kept++] = vertices[i];
	}
	n = kept;
	if (n < 3)
	{
		SetAsBox(1.0f, 1.0f);
		return;
	}

	// gift wrapping, starting from the right-most (then lowest) point
	int32 start = 0;
	for (int32 i = 1; i < n; ++i)
	{
		if (ps[i].x > ps[start].x || (ps[i].x == ps[start].x && ps[i].y < ps[start].y)) start = i;
	}
	int32 hull[b2_maxPolygonVertices];
	int32 m = 0;
	int32 current = start;
	for (;;)
	{
		hull[m] = current;
		int32 candidate = 0;
		for (int32 j = 1; j < n; ++j)
		{
			if (candidate == current)
			{
				candidate = j;
				continue;
			}
			b2Vec2 r = ps[candidate] - ps[hull[m]];
			b2Vec2 v = ps[j] - ps[hull[m]];
			float32 c = b2Cross(r, v);
			if (c < 0.0f) candidate = j;
			if (c == 0.0f && v.LengthSquared() > r.LengthSquared()) candidate = j;
		}
		++m;
		current = candidate;
		if (candidate == start) break;
	}
	if (m < 3)
	{
		SetAsBox(1.0f, 1.0f);
		return;
	}
	m_count = m;
	for (int32 i = 0; i < m; ++i) m_vertices[i] = ps[hull[i]];
	for (int32 i = 0; i < m; ++i)
	{
		int32 i2 = i + 1 < m ? i + 1 : 0;
		b2Vec2 edge = m_vertices[i2] - m_vertices[i];
		m_normals[i] = b2Cross(edge, 1.0f);
		m_normals[i].Normalize();
	}
	m_centroid = PolygonCentroid(m_vertices, m);
}

bool b2PolygonShape::TestPoint(const b2Transform& xf, const b2Vec2& p) const
{
	b2Vec2 pLocal = b2MulT(xf.q, p - xf.p);
	for (int32 i = 0; i < m_count; ++i)
	{
		if (b2Dot(m_normals[i], pLocal - m_vertices[i]) > 0.0f) return false;
	}
	return true;
}

bool b2PolygonShape::RayCast(b2RayCastOutput* output, const b2RayCastInput& input, const b2Transform& xf, int32 childIndex) const
{
	B2_NOT_USED(childIndex);
	b2Vec2 p1 = b2MulT(xf.q, input.p1 - xf.p);
	b2Vec2 p2 = b2MulT(xf.q, input.p2 - xf.p);
	b2Vec2 d = p2 - p1;
	float32 lower = 0.0f, upper = input.maxFraction;
	int32 index = -1;
	for (int32 i = 0; i < m_count; ++i)
	{
		float32 numerator = b2Dot(m_normals[i], m_vertices[i] - p1);
		float32 denominator = b2Dot(m_normals[i], d);
		if (denominator == 0.0f)
		{
			if (numerator < 0.0f) return false;
		}
		else if (denominator < 0.0f && numerator < lower * denominator)
		{
			lower = numerator / denominator;
			index = i;
		}
		else if (denominator > 0.0f && numerator < upper * denominator)
		{
			upper = numerator / denominator;
		}
		if (upper < lower) return false;
	}
	if (index >= 0)
	{
		output->fraction = lower;
		output->normal = b2Mul(xf.q, m_normals[index]);
		return true;
	}
	return false;
}

void b2PolygonShape::ComputeAABB(b2AABB* aabb, const b2Transform& xf, int32 childIndex) const
{
	B2_NOT_USED(childIndex);
	b2Vec2 lower = b2Mul(xf, m_vertices[0]);
	b2Vec2 upper = lower;
	for (int32 i = 1; i < m_count; ++i)
	{
		b2Vec2 v = b2Mul(xf, m_vertices[i]);
		lower = b2Min(lower, v);
		upper = b2Max(upper, v);
	}
	b2Vec2 r(m_radius, m_radius);
	aabb->lowerBound = lower - r;
	aabb->upperBound = upper + r;
}

void b2PolygonShape::ComputeMass(b2MassData* massData, float32 density) const
{
	b2Vec2 center(0.0f, 0.0f);
	float32 area = 0.0f;
	float32 I = 0.0f;
	b2Vec2 s(0.0f, 0.0f);
	for (int32 i = 0; i < m_count; ++i) s += m_vertices[i];
	s *= 1.0f / m_count;
	const float32 k_inv3 = 1.0f / 3.0f;
	for (int32 i = 0; i < m_count; ++i)
	{
		b2Vec2 e1 = m_vertices[i] - s;
		b2Vec2 e2 = i + 1 < m_count ? m_vertices[i + 1] - s : m_vertices[0] - s;
		float32 D = b2Cross(e1, e2);
		float32 triangleArea = 0.5f * D;
		area += triangleArea;
		center += triangleArea * k_inv3 * (e1 + e2);
		float32 ex1 = e1.x, ey1 = e1.y;
		float32 ex2 = e2.x, ey2 = e2.y;
		float32 intx2 = ex1 * ex1 + ex2 * ex1 + ex2 * ex2;
		float32 inty2 = ey1 * ey1 + ey2 * ey1 + ey2 * ey2;
		I += (0.25f * k_inv3 * D) * (intx2 + inty2);
	}
	massData->mass = density * area;
	center *= 1.0f / area;
	massData->center = center + s;
	massData->I = density * I;
	massData->I += massData->mass * (b2Dot(massData->center, massData->center) - b2Dot(center, center));
}

bool b2PolygonShape::Validate() const
{
	for (int32 i = 0; i < m_count; ++i)
	{
		int32 i2 = i < m_count - 1 ? i + 1 : 0;
		b2Vec2 e = m_vertices[i2] - m_vertices[i];
		for (int32 j = 0; j < m_count; ++j)
		{
			if (j == i || j == i2) continue;
			if (b2Cross(e, m_vertices[j] - m_vertices[i]) < 0.0f) return false;
		}
	}
	return true;
}
