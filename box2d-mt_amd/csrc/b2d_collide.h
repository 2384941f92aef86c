// b2d_collide.h - per-contact manifold evaluation (narrow phase) for the MI355X Step() path.
//
// One lane evaluates one contact. Each function follows the tie-breaks and operand order of the
// reference routine it replaces (cited per function) so that manifolds - including feature ids,
// which drive warm-start impulse matching - are bit-identical to the CPU reference.
#ifndef B2D_COLLIDE_H
#define B2D_COLLIDE_H

#include "b2d_math.h"

enum
{
	B2D_SHAPE_CIRCLE = 0, // b2Shape::Type order (b2Shape.h:53-60)
	B2D_SHAPE_EDGE = 1,
	B2D_SHAPE_POLYGON = 2,
	B2D_SHAPE_CHAIN = 3   // ONE child of a b2ChainShape: an edge record (ghost vertices = its neighbours) whose AABB has no radius
};

// an edge or a chain child: the same segment record for the narrow phase, the TOI proxy and the ray cast
#define B2D_IS_SEGMENT(t) ((t) == B2D_SHAPE_EDGE || (t) == B2D_SHAPE_CHAIN)

enum
{
	B2D_MANIFOLD_CIRCLES = 0, // b2Manifold::Type (b2Collision.h:95-100)
	B2D_MANIFOLD_FACE_A = 1,
	B2D_MANIFOLD_FACE_B = 2
};

// One immutable shape record in HBM (152 B). Identical shapes are stored once and shared by every
// fixture that uses them, so box scenes keep their whole shape table in L1/L2.
//   circle : verts[0] = m_p
//   edge   : verts[0] = v1, verts[1] = v2, verts[2] = v0, verts[3] = v3, count bit0 = hasVertex0, bit1 = hasVertex3
//   polygon: verts / normals [count]
struct ShapeRec
{
	int32_t type;
	int32_t count;
	float radius;
	float pad;
	V2 centroid;
	V2 verts[B2D_MAX_POLY_VERTS];
	V2 normals[B2D_MAX_POLY_VERTS];
};

struct Manifold
{
	V2 localNormal;
	V2 localPoint;
	V2 p[2];        // points[i].localPoint
	float ni[2];    // points[i].normalImpulse
	float ti[2];    // points[i].tangentImpulse
	uint32_t id[2]; // points[i].id.key
	int32_t type;
	int32_t pointCount;
};

// b2ContactFeature (b2Collision.h:37-61): {indexA, indexB, typeA, typeB} bytes, little endian key.
#define B2D_CF_VERTEX 0u
#define B2D_CF_FACE 1u
B2D_HD uint32_t b2dMakeId(uint32_t indexA, uint32_t indexB, uint32_t typeA, uint32_t typeB)
{
	return (indexA & 0xffu) | ((indexB & 0xffu) << 8) | (typeA << 16) | (typeB << 24);
}
B2D_HD uint32_t b2dSwapId(uint32_t k)
{
	// swap A<->B features (b2CollidePolygon.cpp:226-233)
	return ((k >> 8) & 0xffu) | ((k & 0xffu) << 8) | (((k >> 24) & 0xffu) << 16) | (((k >> 16) & 0xffu) << 24);
}

struct ClipVertex
{
	V2 v;
	uint32_t id;
};

// b2ClipSegmentToLine (b2Collision.cpp:201-231)
B2D_HD int b2dClipSegmentToLine(ClipVertex vOut[2], const ClipVertex vIn[2], V2 normal, float offset, int vertexIndexA)
{
	int numOut = 0;
	float distance0 = b2dDot(normal, vIn[0].v) - offset;
	float distance1 = b2dDot(normal, vIn[1].v) - offset;
	if (distance0 <= 0.0f) vOut[numOut++] = vIn[0];
	if (distance1 <= 0.0f) vOut[numOut++] = vIn[1];
	if (distance0 * distance1 < 0.0f)
	{
		float interp = distance0 / (distance0 - distance1);
		vOut[numOut].v = vIn[0].v + interp * (vIn[1].v - vIn[0].v);
		// VertexA is hitting edgeB.
		vOut[numOut].id = b2dMakeId((uint32_t)vertexIndexA, (vIn[0].id >> 8) & 0xffu, B2D_CF_VERTEX, B2D_CF_FACE);
		++numOut;
	}
	return numOut;
}

// b2FindMaxSeparation (b2CollidePolygon.cpp:23-62): first max wins (strict >), inner min strict <.
// FIXED > 0: both polygons are known to have FIXED vertices (k_collide's path for two staged box records: the loops unroll,
// the vertex and normal reads are issued together instead of one round trip per loop iteration) - the same operations in the
// same order as with the counts read from the records.
template <int FIXED = 0>
B2D_HD float b2dFindMaxSeparation(int* edgeIndex, const ShapeRec* poly1, Xf xf1, const ShapeRec* poly2, Xf xf2)
{
	const int count1 = FIXED ? FIXED : poly1->count;
	const int count2 = FIXED ? FIXED : poly2->count;
	Xf xf = b2dMulTXX(xf2, xf1);
	int bestIndex = 0;
	float maxSeparation = -B2D_MAXFLOAT;
	for (int i = 0; i < count1; ++i)
	{
		V2 n = b2dMulRV(xf.q, poly1->normals[i]);
		V2 v1 = b2dMulXV(xf, poly1->verts[i]);
		float si = B2D_MAXFLOAT;
		for (int j = 0; j < count2; ++j)
		{
			float sij = b2dDot(n, poly2->verts[j] - v1);
			if (sij < si)
			{
				si = sij;
			}
		}
		if (si > maxSeparation)
		{
			maxSeparation = si;
			bestIndex = i;
		}
	}
	*edgeIndex = bestIndex;
	return maxSeparation;
}

// b2FindIncidentEdge (b2CollidePolygon.cpp:64-107)
template <int FIXED = 0>
B2D_HD void b2dFindIncidentEdge(ClipVertex c[2], const ShapeRec* poly1, Xf xf1, int edge1, const ShapeRec* poly2, Xf xf2)
{
	const int count2 = FIXED ? FIXED : poly2->count;
	V2 normal1 = b2dMulTRV(xf2.q, b2dMulRV(xf1.q, poly1->normals[edge1]));
	int index = 0;
	float minDot = B2D_MAXFLOAT;
	for (int i = 0; i < count2; ++i)
	{
		float dot = b2dDot(normal1, poly2->normals[i]);
		if (dot < minDot)
		{
			minDot = dot;
			index = i;
		}
	}
	int i1 = index;
	int i2 = i1 + 1 < count2 ? i1 + 1 : 0;
	c[0].v = b2dMulXV(xf2, poly2->verts[i1]);
	c[0].id = b2dMakeId((uint32_t)edge1, (uint32_t)i1, B2D_CF_FACE, B2D_CF_VERTEX);
	c[1].v = b2dMulXV(xf2, poly2->verts[i2]);
	c[1].id = b2dMakeId((uint32_t)edge1, (uint32_t)i2, B2D_CF_FACE, B2D_CF_VERTEX);
}

// b2CollidePolygons (b2CollidePolygon.cpp:116-239)
template <int FIXED = 0>
B2D_HD void b2dCollidePolygons(Manifold* m, const ShapeRec* polyA, Xf xfA, const ShapeRec* polyB, Xf xfB)
{
	m->pointCount = 0;
	float totalRadius = polyA->radius + polyB->radius;

	int edgeA = 0;
	float separationA = b2dFindMaxSeparation<FIXED>(&edgeA, polyA, xfA, polyB, xfB);
	if (separationA > totalRadius) return;

	int edgeB = 0;
	float separationB = b2dFindMaxSeparation<FIXED>(&edgeB, polyB, xfB, polyA, xfA);
	if (separationB > totalRadius) return;

	const ShapeRec* poly1;
	const ShapeRec* poly2;
	Xf xf1, xf2;
	int edge1;
	int flip;
	const float k_tol = 0.1f * B2D_LINEAR_SLOP;

	if (separationB > separationA + k_tol)
	{
		poly1 = polyB;
		poly2 = polyA;
		xf1 = xfB;
		xf2 = xfA;
		edge1 = edgeB;
		m->type = B2D_MANIFOLD_FACE_B;
		flip = 1;
	}
	else
	{
		poly1 = polyA;
		poly2 = polyB;
		xf1 = xfA;
		xf2 = xfB;
		edge1 = edgeA;
		m->type = B2D_MANIFOLD_FACE_A;
		flip = 0;
	}

	ClipVertex incidentEdge[2];
	b2dFindIncidentEdge<FIXED>(incidentEdge, poly1, xf1, edge1, poly2, xf2);

	const int count1 = FIXED ? FIXED : poly1->count;
	int iv1 = edge1;
	int iv2 = edge1 + 1 < count1 ? edge1 + 1 : 0;

	V2 v11 = poly1->verts[iv1];
	V2 v12 = poly1->verts[iv2];

	V2 localTangent = v12 - v11;
	b2dNormalize(localTangent);

	V2 localNormal = b2dCrossVS(localTangent, 1.0f);
	V2 planePoint = 0.5f * (v11 + v12);

	V2 tangent = b2dMulRV(xf1.q, localTangent);
	V2 normal = b2dCrossVS(tangent, 1.0f);

	v11 = b2dMulXV(xf1, v11);
	v12 = b2dMulXV(xf1, v12);

	float frontOffset = b2dDot(normal, v11);
	float sideOffset1 = -b2dDot(tangent, v11) + totalRadius;
	float sideOffset2 = b2dDot(tangent, v12) + totalRadius;

	ClipVertex clipPoints1[2];
	ClipVertex clipPoints2[2];
	int np;

	np = b2dClipSegmentToLine(clipPoints1, incidentEdge, -tangent, sideOffset1, iv1);
	if (np < 2) return;

	np = b2dClipSegmentToLine(clipPoints2, clipPoints1, tangent, sideOffset2, iv2);
	if (np < 2) return;

	m->localNormal = localNormal;
	m->localPoint = planePoint;

	int pointCount = 0;
	for (int i = 0; i < B2D_MAX_MANIFOLD_POINTS; ++i)
	{
		float separation = b2dDot(normal, clipPoints2[i].v) - frontOffset;
		if (separation <= totalRadius)
		{
			m->p[pointCount] = b2dMulTXV(xf2, clipPoints2[i].v);
			m->id[pointCount] = flip ? b2dSwapId(clipPoints2[i].id) : clipPoints2[i].id;
			++pointCount;
		}
	}
	m->pointCount = pointCount;
}

// b2CollideCircles (b2CollideCircle.cpp:23-49)
B2D_HD void b2dCollideCircles(Manifold* m, const ShapeRec* circleA, Xf xfA, const ShapeRec* circleB, Xf xfB)
{
	m->pointCount = 0;
	V2 pA = b2dMulXV(xfA, circleA->verts[0]);
	V2 pB = b2dMulXV(xfB, circleB->verts[0]);
	V2 d = pB - pA;
	float distSqr = b2dDot(d, d);
	float rA = circleA->radius, rB = circleB->radius;
	float radius = rA + rB;
	if (distSqr > radius * radius) return;
	m->type = B2D_MANIFOLD_CIRCLES;
	m->localPoint = circleA->verts[0];
	m->localNormal = v2(0.0f, 0.0f);
	m->pointCount = 1;
	m->p[0] = circleB->verts[0];
	m->id[0] = 0;
}

// b2CollidePolygonAndCircle (b2CollideCircle.cpp:51-154)
B2D_HD void b2dCollidePolygonAndCircle(Manifold* m, const ShapeRec* polygonA, Xf xfA, const ShapeRec* circleB, Xf xfB)
{
	m->pointCount = 0;
	V2 c = b2dMulXV(xfB, circleB->verts[0]);
	V2 cLocal = b2dMulTXV(xfA, c);

	int normalIndex = 0;
	float separation = -B2D_MAXFLOAT;
	float radius = polygonA->radius + circleB->radius;
	int vertexCount = polygonA->count;

	for (int i = 0; i < vertexCount; ++i)
	{
		float s = b2dDot(polygonA->normals[i], cLocal - polygonA->verts[i]);
		if (s > radius) return;
		if (s > separation)
		{
			separation = s;
			normalIndex = i;
		}
	}

	int vertIndex1 = normalIndex;
	int vertIndex2 = vertIndex1 + 1 < vertexCount ? vertIndex1 + 1 : 0;
	V2 v1 = polygonA->verts[vertIndex1];
	V2 v2_ = polygonA->verts[vertIndex2];

	if (separation < B2D_EPSILON)
	{
		m->pointCount = 1;
		m->type = B2D_MANIFOLD_FACE_A;
		m->localNormal = polygonA->normals[normalIndex];
		m->localPoint = 0.5f * (v1 + v2_);
		m->p[0] = circleB->verts[0];
		m->id[0] = 0;
		return;
	}

	float u1 = b2dDot(cLocal - v1, v2_ - v1);
	float u2 = b2dDot(cLocal - v2_, v1 - v2_);
	if (u1 <= 0.0f)
	{
		if (b2dDistanceSquared(cLocal, v1) > radius * radius) return;
		m->pointCount = 1;
		m->type = B2D_MANIFOLD_FACE_A;
		m->localNormal = cLocal - v1;
		b2dNormalize(m->localNormal);
		m->localPoint = v1;
		m->p[0] = circleB->verts[0];
		m->id[0] = 0;
	}
	else if (u2 <= 0.0f)
	{
		if (b2dDistanceSquared(cLocal, v2_) > radius * radius) return;
		m->pointCount = 1;
		m->type = B2D_MANIFOLD_FACE_A;
		m->localNormal = cLocal - v2_;
		b2dNormalize(m->localNormal);
		m->localPoint = v2_;
		m->p[0] = circleB->verts[0];
		m->id[0] = 0;
	}
	else
	{
		V2 faceCenter = 0.5f * (v1 + v2_);
		float s = b2dDot(cLocal - faceCenter, polygonA->normals[vertIndex1]);
		if (s > radius) return;
		m->pointCount = 1;
		m->type = B2D_MANIFOLD_FACE_A;
		m->localNormal = polygonA->normals[vertIndex1];
		m->localPoint = faceCenter;
		m->p[0] = circleB->verts[0];
		m->id[0] = 0;
	}
}

// b2CollideEdgeAndCircle (b2CollideEdge.cpp:27-152)
B2D_HD void b2dCollideEdgeAndCircle(Manifold* m, const ShapeRec* edgeA, Xf xfA, const ShapeRec* circleB, Xf xfB)
{
	m->pointCount = 0;
	V2 Q = b2dMulTXV(xfA, b2dMulXV(xfB, circleB->verts[0]));
	V2 A = edgeA->verts[0], B = edgeA->verts[1];
	V2 e = B - A;
	float u = b2dDot(e, B - Q);
	float v = b2dDot(e, Q - A);
	float radius = edgeA->radius + circleB->radius;
	bool hasVertex0 = (edgeA->count & 1) != 0;
	bool hasVertex3 = (edgeA->count & 2) != 0;

	// Region A
	if (v <= 0.0f)
	{
		V2 P = A;
		V2 d = Q - P;
		float dd = b2dDot(d, d);
		if (dd > radius * radius) return;
		if (hasVertex0)
		{
			V2 A1 = edgeA->verts[2];
			V2 B1 = A;
			V2 e1 = B1 - A1;
			float u1 = b2dDot(e1, B1 - Q);
			if (u1 > 0.0f) return;
		}
		m->pointCount = 1;
		m->type = B2D_MANIFOLD_CIRCLES;
		m->localNormal = v2(0.0f, 0.0f);
		m->localPoint = P;
		m->id[0] = b2dMakeId(0, 0, B2D_CF_VERTEX, B2D_CF_VERTEX);
		m->p[0] = circleB->verts[0];
		return;
	}

	// Region B
	if (u <= 0.0f)
	{
		V2 P = B;
		V2 d = Q - P;
		float dd = b2dDot(d, d);
		if (dd > radius * radius) return;
		if (hasVertex3)
		{
			V2 B2 = edgeA->verts[3];
			V2 A2 = B;
			V2 e2 = B2 - A2;
			float v2b = b2dDot(e2, Q - A2);
			if (v2b > 0.0f) return;
		}
		m->pointCount = 1;
		m->type = B2D_MANIFOLD_CIRCLES;
		m->localNormal = v2(0.0f, 0.0f);
		m->localPoint = P;
		m->id[0] = b2dMakeId(1, 0, B2D_CF_VERTEX, B2D_CF_VERTEX);
		m->p[0] = circleB->verts[0];
		return;
	}

	// Region AB
	float den = b2dDot(e, e);
	V2 P = (1.0f / den) * (u * A + v * B);
	V2 d = Q - P;
	float dd = b2dDot(d, d);
	if (dd > radius * radius) return;

	V2 n = v2(-e.y, e.x);
	if (b2dDot(n, Q - A) < 0.0f)
	{
		n = v2(-n.x, -n.y);
	}
	b2dNormalize(n);

	m->pointCount = 1;
	m->type = B2D_MANIFOLD_FACE_A;
	m->localNormal = n;
	m->localPoint = A;
	m->id[0] = b2dMakeId(0, 0, B2D_CF_FACE, B2D_CF_VERTEX);
	m->p[0] = circleB->verts[0];
}

// b2EPCollider::Collide (b2CollideEdge.cpp:230-618) with ComputeEdgeSeparation (:620-638) and
// ComputePolygonSeparation (:640-690).
B2D_HD void b2dCollideEdgeAndPolygon(Manifold* m, const ShapeRec* edgeA, Xf xfA, const ShapeRec* polygonB, Xf xfB)
{
	Xf xf = b2dMulTXX(xfA, xfB);
	V2 centroidB = b2dMulXV(xf, polygonB->centroid);

	V2 v0 = edgeA->verts[2];
	V2 v1 = edgeA->verts[0];
	V2 v2e = edgeA->verts[1];
	V2 v3 = edgeA->verts[3];
	bool hasVertex0 = (edgeA->count & 1) != 0;
	bool hasVertex3 = (edgeA->count & 2) != 0;

	V2 edge1 = v2e - v1;
	b2dNormalize(edge1);
	V2 normal1 = v2(edge1.y, -edge1.x);
	float offset1 = b2dDot(normal1, centroidB - v1);
	float offset0 = 0.0f, offset2 = 0.0f;
	bool convex1 = false, convex2 = false;
	V2 normal0 = v2(0.0f, 0.0f), normal2 = v2(0.0f, 0.0f);

	if (hasVertex0)
	{
		V2 edge0 = v1 - v0;
		b2dNormalize(edge0);
		normal0 = v2(edge0.y, -edge0.x);
		convex1 = b2dCross(edge0, edge1) >= 0.0f;
		offset0 = b2dDot(normal0, centroidB - v0);
	}
	if (hasVertex3)
	{
		V2 edge2 = v3 - v2e;
		b2dNormalize(edge2);
		normal2 = v2(edge2.y, -edge2.x);
		convex2 = b2dCross(edge1, edge2) > 0.0f;
		offset2 = b2dDot(normal2, centroidB - v2e);
	}

	bool front;
	V2 normal, lowerLimit, upperLimit;
	if (hasVertex0 && hasVertex3)
	{
		if (convex1 && convex2)
		{
			front = offset0 >= 0.0f || offset1 >= 0.0f || offset2 >= 0.0f;
			if (front) { normal = normal1; lowerLimit = normal0; upperLimit = normal2; }
			else { normal = -normal1; lowerLimit = -normal1; upperLimit = -normal1; }
		}
		else if (convex1)
		{
			front = offset0 >= 0.0f || (offset1 >= 0.0f && offset2 >= 0.0f);
			if (front) { normal = normal1; lowerLimit = normal0; upperLimit = normal1; }
			else { normal = -normal1; lowerLimit = -normal2; upperLimit = -normal1; }
		}
		else if (convex2)
		{
			front = offset2 >= 0.0f || (offset0 >= 0.0f && offset1 >= 0.0f);
			if (front) { normal = normal1; lowerLimit = normal1; upperLimit = normal2; }
			else { normal = -normal1; lowerLimit = -normal1; upperLimit = -normal0; }
		}
		else
		{
			front = offset0 >= 0.0f && offset1 >= 0.0f && offset2 >= 0.0f;
			if (front) { normal = normal1; lowerLimit = normal1; upperLimit = normal1; }
			else { normal = -normal1; lowerLimit = -normal2; upperLimit = -normal0; }
		}
	}
	else if (hasVertex0)
	{
		if (convex1)
		{
			front = offset0 >= 0.0f || offset1 >= 0.0f;
			if (front) { normal = normal1; lowerLimit = normal0; upperLimit = -normal1; }
			else { normal = -normal1; lowerLimit = normal1; upperLimit = -normal1; }
		}
		else
		{
			front = offset0 >= 0.0f && offset1 >= 0.0f;
			if (front) { normal = normal1; lowerLimit = normal1; upperLimit = -normal1; }
			else { normal = -normal1; lowerLimit = normal1; upperLimit = -normal0; }
		}
	}
	else if (hasVertex3)
	{
		if (convex2)
		{
			front = offset1 >= 0.0f || offset2 >= 0.0f;
			if (front) { normal = normal1; lowerLimit = -normal1; upperLimit = normal2; }
			else { normal = -normal1; lowerLimit = -normal1; upperLimit = normal1; }
		}
		else
		{
			front = offset1 >= 0.0f && offset2 >= 0.0f;
			if (front) { normal = normal1; lowerLimit = -normal1; upperLimit = normal1; }
			else { normal = -normal1; lowerLimit = -normal2; upperLimit = normal1; }
		}
	}
	else
	{
		front = offset1 >= 0.0f;
		if (front) { normal = normal1; lowerLimit = -normal1; upperLimit = -normal1; }
		else { normal = -normal1; lowerLimit = normal1; upperLimit = normal1; }
	}

	// Polygon B in frame A.
	V2 pbV[B2D_MAX_POLY_VERTS];
	V2 pbN[B2D_MAX_POLY_VERTS];
	int pbCount = polygonB->count;
	for (int i = 0; i < pbCount; ++i)
	{
		pbV[i] = b2dMulXV(xf, polygonB->verts[i]);
		pbN[i] = b2dMulRV(xf.q, polygonB->normals[i]);
	}

	float radius = polygonB->radius + edgeA->radius;
	m->pointCount = 0;

	// ComputeEdgeSeparation
	float edgeSeparation = B2D_MAXFLOAT;
	for (int i = 0; i < pbCount; ++i)
	{
		float s = b2dDot(normal, pbV[i] - v1);
		if (s < edgeSeparation)
		{
			edgeSeparation = s;
		}
	}
	// (edge axis type is always e_edgeA)
	if (edgeSeparation > radius) return;

	// ComputePolygonSeparation
	int polyType = 0; // 0 = unknown, 2 = edgeB
	int polyIndex = -1;
	float polySeparation = -B2D_MAXFLOAT;
	{
		V2 perp = v2(-normal.y, normal.x);
		for (int i = 0; i < pbCount; ++i)
		{
			V2 n = -pbN[i];
			float s1 = b2dDot(n, pbV[i] - v1);
			float s2 = b2dDot(n, pbV[i] - v2e);
			float s = b2dMin(s1, s2);
			if (s > radius)
			{
				polyType = 2;
				polyIndex = i;
				polySeparation = s;
				break;
			}
			if (b2dDot(n, perp) >= 0.0f)
			{
				if (b2dDot(n - upperLimit, normal) < -B2D_ANGULAR_SLOP) continue;
			}
			else
			{
				if (b2dDot(n - lowerLimit, normal) < -B2D_ANGULAR_SLOP) continue;
			}
			if (s > polySeparation)
			{
				polyType = 2;
				polyIndex = i;
				polySeparation = s;
			}
		}
	}
	if (polyType != 0 && polySeparation > radius) return;

	const float k_relativeTol = 0.98f;
	const float k_absoluteTol = 0.001f;
	bool primaryIsEdgeA;
	if (polyType == 0)
	{
		primaryIsEdgeA = true;
	}
	else if (polySeparation > k_relativeTol * edgeSeparation + k_absoluteTol)
	{
		primaryIsEdgeA = false;
	}
	else
	{
		primaryIsEdgeA = true;
	}

	ClipVertex ie[2];
	int rf_i1, rf_i2;
	V2 rf_v1, rf_v2, rf_normal;
	if (primaryIsEdgeA)
	{
		m->type = B2D_MANIFOLD_FACE_A;
		int bestIndex = 0;
		float bestValue = b2dDot(normal, pbN[0]);
		for (int i = 1; i < pbCount; ++i)
		{
			float value = b2dDot(normal, pbN[i]);
			if (value < bestValue)
			{
				bestValue = value;
				bestIndex = i;
			}
		}
		int i1 = bestIndex;
		int i2 = i1 + 1 < pbCount ? i1 + 1 : 0;
		ie[0].v = pbV[i1];
		ie[0].id = b2dMakeId(0, (uint32_t)i1, B2D_CF_FACE, B2D_CF_VERTEX);
		ie[1].v = pbV[i2];
		ie[1].id = b2dMakeId(0, (uint32_t)i2, B2D_CF_FACE, B2D_CF_VERTEX);
		if (front)
		{
			rf_i1 = 0;
			rf_i2 = 1;
			rf_v1 = v1;
			rf_v2 = v2e;
			rf_normal = normal1;
		}
		else
		{
			rf_i1 = 1;
			rf_i2 = 0;
			rf_v1 = v2e;
			rf_v2 = v1;
			rf_normal = -normal1;
		}
	}
	else
	{
		m->type = B2D_MANIFOLD_FACE_B;
		ie[0].v = v1;
		ie[0].id = b2dMakeId(0, (uint32_t)polyIndex, B2D_CF_VERTEX, B2D_CF_FACE);
		ie[1].v = v2e;
		ie[1].id = b2dMakeId(0, (uint32_t)polyIndex, B2D_CF_VERTEX, B2D_CF_FACE);
		rf_i1 = polyIndex;
		rf_i2 = rf_i1 + 1 < pbCount ? rf_i1 + 1 : 0;
		rf_v1 = pbV[rf_i1];
		rf_v2 = pbV[rf_i2];
		rf_normal = pbN[rf_i1];
	}

	V2 sideNormal1 = v2(rf_normal.y, -rf_normal.x);
	V2 sideNormal2 = -sideNormal1;
	float sideOffset1 = b2dDot(sideNormal1, rf_v1);
	float sideOffset2 = b2dDot(sideNormal2, rf_v2);

	ClipVertex clipPoints1[2];
	ClipVertex clipPoints2[2];
	int np;
	np = b2dClipSegmentToLine(clipPoints1, ie, sideNormal1, sideOffset1, rf_i1);
	if (np < B2D_MAX_MANIFOLD_POINTS) return;
	np = b2dClipSegmentToLine(clipPoints2, clipPoints1, sideNormal2, sideOffset2, rf_i2);
	if (np < B2D_MAX_MANIFOLD_POINTS) return;

	if (primaryIsEdgeA)
	{
		m->localNormal = rf_normal;
		m->localPoint = rf_v1;
	}
	else
	{
		m->localNormal = polygonB->normals[rf_i1];
		m->localPoint = polygonB->verts[rf_i1];
	}

	int pointCount = 0;
	for (int i = 0; i < B2D_MAX_MANIFOLD_POINTS; ++i)
	{
		float separation = b2dDot(rf_normal, clipPoints2[i].v - rf_v1);
		if (separation <= radius)
		{
			if (primaryIsEdgeA)
			{
				m->p[pointCount] = b2dMulTXV(xf, clipPoints2[i].v);
				m->id[pointCount] = clipPoints2[i].id;
			}
			else
			{
				m->p[pointCount] = clipPoints2[i].v;
				m->id[pointCount] = b2dSwapId(clipPoints2[i].id);
			}
			++pointCount;
		}
	}
	m->pointCount = pointCount;
}

// Evaluate dispatch: the (typeA, typeB) combinations b2Contact::Create can produce
// (b2Contact.cpp:42-52, 72-98): A/B already ordered so that typeA >= typeB ... see b2dOrderTypes.
B2D_HD void b2dEvaluate(Manifold* m, const ShapeRec* sA, Xf xfA, const ShapeRec* sB, Xf xfB)
{
	// chain children collide as the edge b2ChainShape::GetChildEdge hands out (b2ChainAndPolygonContact.cpp:45-53,
	// b2ChainAndCircleContact.cpp:45-53)
	int tA = sA->type == B2D_SHAPE_CHAIN ? B2D_SHAPE_EDGE : sA->type, tB = sB->type;
	if (tA == B2D_SHAPE_POLYGON && tB == B2D_SHAPE_POLYGON)
		b2dCollidePolygons(m, sA, xfA, sB, xfB);
	else if (tA == B2D_SHAPE_POLYGON && tB == B2D_SHAPE_CIRCLE)
		b2dCollidePolygonAndCircle(m, sA, xfA, sB, xfB);
	else if (tA == B2D_SHAPE_CIRCLE && tB == B2D_SHAPE_CIRCLE)
		b2dCollideCircles(m, sA, xfA, sB, xfB);
	else if (tA == B2D_SHAPE_EDGE && tB == B2D_SHAPE_POLYGON)
		b2dCollideEdgeAndPolygon(m, sA, xfA, sB, xfB);
	else if (tA == B2D_SHAPE_EDGE && tB == B2D_SHAPE_CIRCLE)
		b2dCollideEdgeAndCircle(m, sA, xfA, sB, xfB);
	else
		m->pointCount = 0;
}

// Which fixture becomes "A": b2Contact::Create swaps when the register entry is not primary
// (b2Contact.cpp:82-93). Primary entries: circle-circle, polygon-circle, polygon-polygon,
// edge-circle, edge-polygon. Returns 1 if (type1, type2) must be swapped, 0 if kept, -1 if no contact type.
B2D_HD int b2dContactSwap(int type1, int type2)
{
	// chain-circle and chain-polygon are primary like their edge forms (b2Contact.cpp:47-52); chain-chain, chain-edge: none
	if (type1 == B2D_SHAPE_CHAIN) type1 = B2D_SHAPE_EDGE;
	if (type2 == B2D_SHAPE_CHAIN) type2 = B2D_SHAPE_EDGE;
	if (type1 == B2D_SHAPE_CIRCLE && type2 == B2D_SHAPE_CIRCLE) return 0;
	if (type1 == B2D_SHAPE_POLYGON && type2 == B2D_SHAPE_CIRCLE) return 0;
	if (type1 == B2D_SHAPE_CIRCLE && type2 == B2D_SHAPE_POLYGON) return 1;
	if (type1 == B2D_SHAPE_POLYGON && type2 == B2D_SHAPE_POLYGON) return 0;
	if (type1 == B2D_SHAPE_EDGE && type2 == B2D_SHAPE_CIRCLE) return 0;
	if (type1 == B2D_SHAPE_CIRCLE && type2 == B2D_SHAPE_EDGE) return 1;
	if (type1 == B2D_SHAPE_EDGE && type2 == B2D_SHAPE_POLYGON) return 0;
	if (type1 == B2D_SHAPE_POLYGON && type2 == B2D_SHAPE_EDGE) return 1;
	return -1; // edge-edge, chain-* not on the device path
}

// Shape AABBs: b2PolygonShape::ComputeAABB (b2PolygonShape.cpp:340-357),
// b2CircleShape::ComputeAABB (b2CircleShape.cpp:83-90), b2EdgeShape::ComputeAABB (b2EdgeShape.cpp:116-129)
B2D_HD AABB b2dShapeAABB(const ShapeRec* s, Xf xf)
{
	AABB r;
	if (s->type == B2D_SHAPE_CIRCLE)
	{
		V2 q = b2dMulRV(xf.q, s->verts[0]);
		V2 p = v2(xf.p.x + q.x, xf.p.y + q.y);
		r.lo = v2(p.x - s->radius, p.y - s->radius);
		r.hi = v2(p.x + s->radius, p.y + s->radius);
		return r;
	}
	if (s->type == B2D_SHAPE_CHAIN)
	{
		// b2ChainShape::ComputeAABB (b2ChainShape.cpp:174-189): the child's two vertices, NO radius
		V2 a = b2dMulXV(xf, s->verts[0]);
		V2 b = b2dMulXV(xf, s->verts[1]);
		r.lo = b2dMinV(a, b);
		r.hi = b2dMaxV(a, b);
		return r;
	}
	if (s->type == B2D_SHAPE_EDGE)
	{
		V2 a = b2dMulXV(xf, s->verts[0]);
		V2 b = b2dMulXV(xf, s->verts[1]);
		V2 lower = b2dMinV(a, b);
		V2 upper = b2dMaxV(a, b);
		r.lo = v2(lower.x - s->radius, lower.y - s->radius);
		r.hi = v2(upper.x + s->radius, upper.y + s->radius);
		return r;
	}
	V2 lower = b2dMulXV(xf, s->verts[0]);
	V2 upper = lower;
	for (int i = 1; i < s->count; ++i)
	{
		V2 v = b2dMulXV(xf, s->verts[i]);
		lower = b2dMinV(lower, v);
		upper = b2dMaxV(upper, v);
	}
	r.lo = v2(lower.x - s->radius, lower.y - s->radius);
	r.hi = v2(upper.x + s->radius, upper.y + s->radius);
	return r;
}

#endif
