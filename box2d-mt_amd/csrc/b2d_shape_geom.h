// b2d_shape_geom.h - geometry of ONE shape record (ShapeRec, 152 B): construction of a polygon record from a point cloud,
// mass properties, point test, ray cast. (The AABB of a record is b2dShapeAABB in b2d_collide.h.)
//
// One statement of this arithmetic for the whole tree: the C-ABI layer (b2hip.hip: fixture mass, queries) and the drop-in
// host classes (host/src/b2_shapes.cpp: b2PolygonShape / b2CircleShape / b2EdgeShape / b2ChainShape are views of a record)
// both call these functions. Results must equal the reference's bit for bit (fixture mass and the first AABB of a body
// enter every later float), so each routine keeps the reference's OPERAND ORDER - cited per function - with one fp32
// rounding per operation; layout, control flow and naming are this repo's.
#ifndef B2D_SHAPE_GEOM_H
#define B2D_SHAPE_GEOM_H

#include "b2d_collide.h"

struct MassProps
{
	float mass;
	V2 center;
	float inertia; // about the shape's origin
};

struct RayHit
{
	float fraction;
	V2 normal;
};

// ---- polygon records ------------------------------------------------------------------------------------------------------

// Outward unit normals of the hull edges, then the area centroid by a triangle fan about the origin.
// Operand order: b2PolygonShape::Set's normal loop (b2PolygonShape.cpp:232-240) and ComputeCentroid (:72-118).
B2D_HD void b2dPolygonFinish(ShapeRec* s)
{
	const int n = s->count;
	for (int k = 0; k < n; ++k)
	{
		const V2 side = s->verts[k + 1 < n ? k + 1 : 0] - s->verts[k];
		V2 out = b2dCrossVS(side, 1.0f);
		b2dNormalize(out);
		s->normals[k] = out;
	}
	const float third = 1.0f / 3.0f;
	const V2 origin = v2(0.0f, 0.0f);
	V2 weighted = v2(0.0f, 0.0f);
	float total = 0.0f;
	for (int k = 0; k < n; ++k)
	{
		const V2 a = s->verts[k];
		const V2 b = s->verts[k + 1 < n ? k + 1 : 0];
		const float fan = 0.5f * b2dCross(a - origin, b - origin);
		total += fan;
		weighted += fan * third * (origin + a + b);
	}
	weighted *= 1.0f / total;
	s->centroid = weighted;
}

// An axis-aligned box about the origin (b2PolygonShape::SetAsBox, b2PolygonShape.cpp:30-42): corners counter-clockwise
// from (-hx, -hy), axis normals, centroid zero.
B2D_HD void b2dPolygonBox(ShapeRec* s, float hx, float hy)
{
	s->type = B2D_SHAPE_POLYGON;
	s->count = 4;
	s->radius = B2D_POLYGON_RADIUS;
	s->pad = 0.0f;
	for (int k = 0; k < B2D_MAX_POLY_VERTS; ++k) s->verts[k] = s->normals[k] = v2(0.0f, 0.0f);
	const float sx[4] = {-1.0f, 1.0f, 1.0f, -1.0f}, sy[4] = {-1.0f, -1.0f, 1.0f, 1.0f};
	const float nx[4] = {0.0f, 1.0f, 0.0f, -1.0f}, ny[4] = {-1.0f, 0.0f, 1.0f, 0.0f};
	for (int k = 0; k < 4; ++k)
	{
		s->verts[k] = v2(sx[k] < 0.0f ? -hx : hx, sy[k] < 0.0f ? -hy : hy);
		s->normals[k] = v2(nx[k], ny[k]);
	}
	s->centroid = v2(0.0f, 0.0f);
}

// The same box moved to `center` and turned by `angle` (b2PolygonShape.cpp:44-67): vertices through the transform,
// normals through its rotation, centroid = center.
B2D_HD void b2dPolygonBoxAt(ShapeRec* s, float hx, float hy, V2 center, float angle)
{
	b2dPolygonBox(s, hx, hy);
	Xf place;
	place.p = center;
	place.q = b2dRotInline(angle);
	for (int k = 0; k < 4; ++k)
	{
		s->verts[k] = b2dMulXV(place, s->verts[k]);
		s->normals[k] = b2dMulRV(place.q, s->normals[k]);
	}
	s->centroid = center;
}

// Convex hull of up to 8 points as a polygon record (b2PolygonShape::Set, b2PolygonShape.cpp:120-250): points closer than
// half a linear slop to an earlier one are dropped, the hull is wrapped counter-clockwise from the right-most (then
// lowest) point, collinear points keep the farthest. Degenerate input gives the reference's release-build answer, a
// 1 x 1 box. Returns the vertex count.
B2D_HD int b2dPolygonFromPoints(ShapeRec* s, const V2* cloud, int cloudCount)
{
	s->type = B2D_SHAPE_POLYGON;
	s->radius = B2D_POLYGON_RADIUS;
	s->pad = 0.0f;
	V2 pts[B2D_MAX_POLY_VERTS];
	int n = 0;
	if (cloudCount >= 3)
	{
		const int take = cloudCount < B2D_MAX_POLY_VERTS ? cloudCount : B2D_MAX_POLY_VERTS;
		const float tooClose = (0.5f * B2D_LINEAR_SLOP) * (0.5f * B2D_LINEAR_SLOP);
		for (int i = 0; i < take; ++i)
		{
			int twin = -1;
			for (int j = 0; j < n && twin < 0; ++j)
				if (b2dDistanceSquared(cloud[i], pts[j]) < tooClose) twin = j;
			if (twin < 0) pts[n++] = cloud[i];
		}
	}
	int ring[B2D_MAX_POLY_VERTS];
	int m = 0;
	if (n >= 3)
	{
		int first = 0;
		for (int i = 1; i < n; ++i)
		{
			const bool further = pts[i].x > pts[first].x;
			const bool below = pts[i].x == pts[first].x && pts[i].y < pts[first].y;
			if (further || below) first = i;
		}
		int at = first;
		do
		{
			ring[m] = at;
			int next = 0;
			for (int j = 1; j < n; ++j)
			{
				if (next == at)
				{
					next = j;
					continue;
				}
				const V2 toNext = pts[next] - pts[at];
				const V2 toJ = pts[j] - pts[at];
				const float turn = b2dCross(toNext, toJ);
				if (turn < 0.0f) next = j;
				if (turn == 0.0f && b2dDot(toJ, toJ) > b2dDot(toNext, toNext)) next = j;
			}
			++m;
			at = next;
		} while (at != first);
	}
	if (m < 3)
	{
		b2dPolygonBox(s, 1.0f, 1.0f);
		return 4;
	}
	s->count = m;
	for (int k = 0; k < B2D_MAX_POLY_VERTS; ++k) s->verts[k] = s->normals[k] = v2(0.0f, 0.0f);
	for (int k = 0; k < m; ++k) s->verts[k] = pts[ring[k]];
	b2dPolygonFinish(s);
	return m;
}

// b2PolygonShape::Validate (b2PolygonShape.cpp:442-467): no vertex strictly to the right of any edge.
B2D_HD bool b2dPolygonConvex(const ShapeRec* s)
{
	const int n = s->count;
	for (int k = 0; k < n; ++k)
	{
		const int k2 = k + 1 < n ? k + 1 : 0;
		const V2 side = s->verts[k2] - s->verts[k];
		for (int j = 0; j < n; ++j)
			if (j != k && j != k2 && b2dCross(side, s->verts[j] - s->verts[k]) < 0.0f) return false;
	}
	return true;
}

// ---- mass -------------------------------------------------------------------------------------------------------------------

// b2CircleShape::ComputeMass (b2CircleShape.cpp:92-100), b2EdgeShape::ComputeMass (b2EdgeShape.cpp:131-138; a chain
// child is an edge: b2ChainShape.cpp:191-197 reports zero as well), b2PolygonShape::ComputeMass (b2PolygonShape.cpp:359-440:
// triangle fan about the vertex mean, second moments per triangle, parallel-axis shift back to the origin).
B2D_HD MassProps b2dShapeMass(const ShapeRec* s, float density)
{
	MassProps mp;
	if (s->type == B2D_SHAPE_CIRCLE)
	{
		const V2 at = s->verts[0];
		mp.mass = density * B2D_PI * s->radius * s->radius;
		mp.center = at;
		mp.inertia = mp.mass * (0.5f * s->radius * s->radius + b2dDot(at, at));
		return mp;
	}
	if (s->type != B2D_SHAPE_POLYGON)
	{
		mp.mass = 0.0f;
		mp.center = 0.5f * (s->verts[0] + s->verts[1]);
		mp.inertia = 0.0f;
		return mp;
	}
	const int n = s->count;
	V2 mean = v2(0.0f, 0.0f);
	for (int k = 0; k < n; ++k) mean += s->verts[k];
	mean *= 1.0f / n;
	const float third = 1.0f / 3.0f;
	float total = 0.0f, second = 0.0f;
	V2 first = v2(0.0f, 0.0f);
	for (int k = 0; k < n; ++k)
	{
		const V2 p = s->verts[k] - mean;
		const V2 q = s->verts[k + 1 < n ? k + 1 : 0] - mean;
		const float twice = b2dCross(p, q);
		const float fan = 0.5f * twice;
		total += fan;
		first += fan * third * (p + q);
		const float xx = p.x * p.x + q.x * p.x + q.x * q.x;
		const float yy = p.y * p.y + q.y * p.y + q.y * q.y;
		second += (0.25f * third * twice) * (xx + yy);
	}
	mp.mass = density * total;
	first *= 1.0f / total;
	mp.center = first + mean;
	mp.inertia = density * second;
	mp.inertia += mp.mass * (b2dDot(mp.center, mp.center) - b2dDot(first, first));
	return mp;
}

// ---- point test -------------------------------------------------------------------------------------------------------------

// b2CircleShape::TestPoint (b2CircleShape.cpp:38-43), b2PolygonShape::TestPoint (b2PolygonShape.cpp:252-266); edges and
// chain children contain no point (b2EdgeShape.cpp:47-52).
B2D_HD bool b2dShapeTestPoint(const ShapeRec* s, Xf xf, V2 worldPoint)
{
	if (s->type == B2D_SHAPE_CIRCLE)
	{
		const V2 mid = xf.p + b2dMulRV(xf.q, s->verts[0]);
		const V2 off = worldPoint - mid;
		return b2dDot(off, off) <= s->radius * s->radius;
	}
	if (s->type != B2D_SHAPE_POLYGON) return false;
	const V2 local = b2dMulTRV(xf.q, worldPoint - xf.p);
	for (int k = 0; k < s->count; ++k)
		if (b2dDot(s->normals[k], local - s->verts[k]) > 0.0f) return false;
	return true;
}

// ---- ray cast ---------------------------------------------------------------------------------------------------------------

// Ray p1 -> p2 against the record at `xf`, fractions up to maxFraction. Circle: quadratic in the ray parameter
// (b2CircleShape.cpp:49-81). Edge / chain child: supporting line, then the segment's parameter range
// (b2EdgeShape.cpp:58-114). Polygon: the ray clipped by every half-plane in the shape's frame (b2PolygonShape.cpp:268-338).
B2D_HD bool b2dShapeRayCast(const ShapeRec* s, Xf xf, V2 p1, V2 p2, float maxFraction, RayHit* hit)
{
	if (s->type == B2D_SHAPE_CIRCLE)
	{
		const V2 mid = xf.p + b2dMulRV(xf.q, s->verts[0]);
		const V2 from = p1 - mid;
		const float outside = b2dDot(from, from) - s->radius * s->radius;
		const V2 along = p2 - p1;
		const float proj = b2dDot(from, along);
		const float len2 = b2dDot(along, along);
		const float disc = proj * proj - len2 * outside;
		if (disc < 0.0f || len2 < B2D_EPSILON) return false;
		float t = -(proj + b2dSqrt(disc));
		if (!(0.0f <= t && t <= maxFraction * len2)) return false;
		t /= len2;
		hit->fraction = t;
		hit->normal = from + t * along;
		b2dNormalize(hit->normal);
		return true;
	}
	const V2 a = b2dMulTRV(xf.q, p1 - xf.p);
	const V2 b = b2dMulTRV(xf.q, p2 - xf.p);
	const V2 along = b - a;
	if (s->type != B2D_SHAPE_POLYGON)
	{
		const V2 v1 = s->verts[0], v2_ = s->verts[1];
		const V2 seg = v2_ - v1;
		V2 side = v2(seg.y, -seg.x);
		b2dNormalize(side);
		const float gap = b2dDot(side, v1 - a);
		const float rate = b2dDot(side, along);
		if (rate == 0.0f) return false;
		const float t = gap / rate;
		if (t < 0.0f || maxFraction < t) return false;
		const V2 where = a + t * along;
		const float len2 = b2dDot(seg, seg);
		if (len2 == 0.0f) return false;
		const float u = b2dDot(where - v1, seg) / len2;
		if (u < 0.0f || 1.0f < u) return false;
		hit->fraction = t;
		const V2 worldSide = b2dMulRV(xf.q, side);
		hit->normal = gap > 0.0f ? -worldSide : worldSide;
		return true;
	}
	float enter = 0.0f, leave = maxFraction;
	int face = -1;
	for (int k = 0; k < s->count; ++k)
	{
		const float gap = b2dDot(s->normals[k], s->verts[k] - a);
		const float rate = b2dDot(s->normals[k], along);
		if (rate == 0.0f)
		{
			if (gap < 0.0f) return false; // parallel and outside this half-plane
		}
		else if (rate < 0.0f && gap < enter * rate)
		{
			enter = gap / rate;
			face = k;
		}
		else if (rate > 0.0f && gap < leave * rate)
		{
			leave = gap / rate;
		}
		if (leave < enter) return false;
	}
	if (face < 0) return false;
	hit->fraction = enter;
	hit->normal = b2dMulRV(xf.q, s->normals[face]);
	return true;
}

#endif
