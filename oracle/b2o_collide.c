/* b2o_collide.c - CPU oracle, narrow phase: plain-C restatement of the reference's manifold routines.
 * TEST INFRASTRUCTURE (see b2o.h). Each function cites the reference lines it follows; operand order
 * is kept so that results are bit-identical to the reference built without FP contraction. */
#include "b2o_internal.h"

#include <string.h>

/* b2ClipSegmentToLine  Box2D/Collision/b2Collision.cpp:201-231 */
static int clip_segment(clipv out[2], const clipv in[2], vec2 normal, float offset, int vertexIndexA)
{
	int n = 0;
	float d0 = v_dot(normal, in[0].v) - offset;
	float d1 = v_dot(normal, in[1].v) - offset;
	if (d0 <= 0.0f) out[n++] = in[0];
	if (d1 <= 0.0f) out[n++] = in[1];
	if (d0 * d1 < 0.0f)
	{
		float interp = d0 / (d0 - d1);
		out[n].v = v_add(in[0].v, v_scale(interp, v_sub(in[1].v, in[0].v)));
		out[n].id = make_id((uint32_t)vertexIndexA, (in[0].id >> 8) & 0xffu, CF_VERTEX, CF_FACE);
		++n;
	}
	return n;
}

/* b2FindMaxSeparation  b2CollidePolygon.cpp:23-62 */
static float find_max_separation(int* edgeIndex, const b2o_shape* p1, xform xf1, const b2o_shape* p2, xform xf2)
{
	xform xf = xf_mul_t(xf2, xf1);
	int best = 0;
	float maxSep = -B2O_MAXFLOAT;
	for (int i = 0; i < p1->count; ++i)
	{
		vec2 n = r_mul(xf.q, shape_normal(p1, i));
		vec2 v1 = xf_mul(xf, shape_vert(p1, i));
		float si = B2O_MAXFLOAT;
		for (int j = 0; j < p2->count; ++j)
		{
			float sij = v_dot(n, v_sub(shape_vert(p2, j), v1));
			if (sij < si) si = sij;
		}
		if (si > maxSep)
		{
			maxSep = si;
			best = i;
		}
	}
	*edgeIndex = best;
	return maxSep;
}

/* b2FindIncidentEdge  b2CollidePolygon.cpp:64-107 */
static void find_incident_edge(clipv c[2], const b2o_shape* p1, xform xf1, int edge1, const b2o_shape* p2, xform xf2)
{
	vec2 normal1 = r_mul_t(xf2.q, r_mul(xf1.q, shape_normal(p1, edge1)));
	int index = 0;
	float minDot = B2O_MAXFLOAT;
	for (int i = 0; i < p2->count; ++i)
	{
		float d = v_dot(normal1, shape_normal(p2, i));
		if (d < minDot)
		{
			minDot = d;
			index = i;
		}
	}
	int i1 = index;
	int i2 = i1 + 1 < p2->count ? i1 + 1 : 0;
	c[0].v = xf_mul(xf2, shape_vert(p2, i1));
	c[0].id = make_id((uint32_t)edge1, (uint32_t)i1, CF_FACE, CF_VERTEX);
	c[1].v = xf_mul(xf2, shape_vert(p2, i2));
	c[1].id = make_id((uint32_t)edge1, (uint32_t)i2, CF_FACE, CF_VERTEX);
}

/* b2CollidePolygons  b2CollidePolygon.cpp:116-239 */
void b2o_collide_polygons(manifold* m, const b2o_shape* polyA, xform xfA, const b2o_shape* polyB, xform xfB)
{
	m->pointCount = 0;
	float totalRadius = polyA->radius + polyB->radius;
	int edgeA = 0;
	float sepA = find_max_separation(&edgeA, polyA, xfA, polyB, xfB);
	if (sepA > totalRadius) return;
	int edgeB = 0;
	float sepB = find_max_separation(&edgeB, polyB, xfB, polyA, xfA);
	if (sepB > totalRadius) return;

	const b2o_shape *poly1, *poly2;
	xform xf1, xf2;
	int edge1, flip;
	const float k_tol = 0.1f * B2O_LINEAR_SLOP;
	if (sepB > sepA + k_tol)
	{
		poly1 = polyB; poly2 = polyA; xf1 = xfB; xf2 = xfA; edge1 = edgeB;
		m->type = MANIFOLD_FACE_B;
		flip = 1;
	}
	else
	{
		poly1 = polyA; poly2 = polyB; xf1 = xfA; xf2 = xfB; edge1 = edgeA;
		m->type = MANIFOLD_FACE_A;
		flip = 0;
	}
	clipv incident[2];
	find_incident_edge(incident, poly1, xf1, edge1, poly2, xf2);

	int iv1 = edge1;
	int iv2 = edge1 + 1 < poly1->count ? edge1 + 1 : 0;
	vec2 v11 = shape_vert(poly1, iv1);
	vec2 v12 = shape_vert(poly1, iv2);
	vec2 localTangent = v_sub(v12, v11);
	v_normalize(&localTangent);
	vec2 localNormal = v_cross_vs(localTangent, 1.0f);
	vec2 planePoint = v_scale(0.5f, v_add(v11, v12));
	vec2 tangent = r_mul(xf1.q, localTangent);
	vec2 normal = v_cross_vs(tangent, 1.0f);
	v11 = xf_mul(xf1, v11);
	v12 = xf_mul(xf1, v12);
	float frontOffset = v_dot(normal, v11);
	float sideOffset1 = -v_dot(tangent, v11) + totalRadius;
	float sideOffset2 = v_dot(tangent, v12) + totalRadius;

	clipv cp1[2], cp2[2];
	int np = clip_segment(cp1, incident, v_neg(tangent), sideOffset1, iv1);
	if (np < 2) return;
	np = clip_segment(cp2, cp1, tangent, sideOffset2, iv2);
	if (np < 2) return;

	m->localNormal = localNormal;
	m->localPoint = planePoint;
	int pc = 0;
	for (int i = 0; i < 2; ++i)
	{
		float separation = v_dot(normal, cp2[i].v) - frontOffset;
		if (separation <= totalRadius)
		{
			m->p[pc] = xf_mul_tv(xf2, cp2[i].v);
			m->id[pc] = flip ? swap_id(cp2[i].id) : cp2[i].id;
			++pc;
		}
	}
	m->pointCount = pc;
}

/* b2CollideCircles  b2CollideCircle.cpp:23-49 */
void b2o_collide_circles(manifold* m, const b2o_shape* cA, xform xfA, const b2o_shape* cB, xform xfB)
{
	m->pointCount = 0;
	vec2 pA = xf_mul(xfA, shape_vert(cA, 0));
	vec2 pB = xf_mul(xfB, shape_vert(cB, 0));
	vec2 d = v_sub(pB, pA);
	float distSqr = v_dot(d, d);
	float radius = cA->radius + cB->radius;
	if (distSqr > radius * radius) return;
	m->type = MANIFOLD_CIRCLES;
	m->localPoint = shape_vert(cA, 0);
	m->localNormal = v_make(0.0f, 0.0f);
	m->pointCount = 1;
	m->p[0] = shape_vert(cB, 0);
	m->id[0] = 0;
}

/* b2CollidePolygonAndCircle  b2CollideCircle.cpp:51-154 */
void b2o_collide_polygon_circle(manifold* m, const b2o_shape* polyA, xform xfA, const b2o_shape* cB, xform xfB)
{
	m->pointCount = 0;
	vec2 c = xf_mul(xfB, shape_vert(cB, 0));
	vec2 cLocal = xf_mul_tv(xfA, c);
	int normalIndex = 0;
	float separation = -B2O_MAXFLOAT;
	float radius = polyA->radius + cB->radius;
	int n = polyA->count;
	for (int i = 0; i < n; ++i)
	{
		float s = v_dot(shape_normal(polyA, i), v_sub(cLocal, shape_vert(polyA, i)));
		if (s > radius) return;
		if (s > separation)
		{
			separation = s;
			normalIndex = i;
		}
	}
	int vi1 = normalIndex;
	int vi2 = vi1 + 1 < n ? vi1 + 1 : 0;
	vec2 v1 = shape_vert(polyA, vi1);
	vec2 v2 = shape_vert(polyA, vi2);
	if (separation < B2O_EPSILON)
	{
		m->pointCount = 1;
		m->type = MANIFOLD_FACE_A;
		m->localNormal = shape_normal(polyA, normalIndex);
		m->localPoint = v_scale(0.5f, v_add(v1, v2));
		m->p[0] = shape_vert(cB, 0);
		m->id[0] = 0;
		return;
	}
	float u1 = v_dot(v_sub(cLocal, v1), v_sub(v2, v1));
	float u2 = v_dot(v_sub(cLocal, v2), v_sub(v1, v2));
	if (u1 <= 0.0f)
	{
		if (v_dist_sq(cLocal, v1) > radius * radius) return;
		m->pointCount = 1;
		m->type = MANIFOLD_FACE_A;
		m->localNormal = v_sub(cLocal, v1);
		v_normalize(&m->localNormal);
		m->localPoint = v1;
		m->p[0] = shape_vert(cB, 0);
		m->id[0] = 0;
	}
	else if (u2 <= 0.0f)
	{
		if (v_dist_sq(cLocal, v2) > radius * radius) return;
		m->pointCount = 1;
		m->type = MANIFOLD_FACE_A;
		m->localNormal = v_sub(cLocal, v2);
		v_normalize(&m->localNormal);
		m->localPoint = v2;
		m->p[0] = shape_vert(cB, 0);
		m->id[0] = 0;
	}
	else
	{
		vec2 faceCenter = v_scale(0.5f, v_add(v1, v2));
		float s = v_dot(v_sub(cLocal, faceCenter), shape_normal(polyA, vi1));
		if (s > radius) return;
		m->pointCount = 1;
		m->type = MANIFOLD_FACE_A;
		m->localNormal = shape_normal(polyA, vi1);
		m->localPoint = faceCenter;
		m->p[0] = shape_vert(cB, 0);
		m->id[0] = 0;
	}
}

/* b2CollideEdgeAndCircle  b2CollideEdge.cpp:27-152 */
void b2o_collide_edge_circle(manifold* m, const b2o_shape* edgeA, xform xfA, const b2o_shape* cB, xform xfB)
{
	m->pointCount = 0;
	vec2 Q = xf_mul_tv(xfA, xf_mul(xfB, shape_vert(cB, 0)));
	vec2 A = shape_vert(edgeA, 0), B = shape_vert(edgeA, 1);
	vec2 e = v_sub(B, A);
	float u = v_dot(e, v_sub(B, Q));
	float v = v_dot(e, v_sub(Q, A));
	float radius = edgeA->radius + cB->radius;
	int has0 = edgeA->count & 1, has3 = edgeA->count & 2;
	if (v <= 0.0f)
	{
		vec2 P = A;
		vec2 d = v_sub(Q, P);
		if (v_dot(d, d) > radius * radius) return;
		if (has0)
		{
			vec2 A1 = shape_vert(edgeA, 2);
			vec2 e1 = v_sub(A, A1);
			float u1 = v_dot(e1, v_sub(A, Q));
			if (u1 > 0.0f) return;
		}
		m->pointCount = 1;
		m->type = MANIFOLD_CIRCLES;
		m->localNormal = v_make(0.0f, 0.0f);
		m->localPoint = P;
		m->id[0] = make_id(0, 0, CF_VERTEX, CF_VERTEX);
		m->p[0] = shape_vert(cB, 0);
		return;
	}
	if (u <= 0.0f)
	{
		vec2 P = B;
		vec2 d = v_sub(Q, P);
		if (v_dot(d, d) > radius * radius) return;
		if (has3)
		{
			vec2 B2 = shape_vert(edgeA, 3);
			vec2 e2 = v_sub(B2, B);
			float v2 = v_dot(e2, v_sub(Q, B));
			if (v2 > 0.0f) return;
		}
		m->pointCount = 1;
		m->type = MANIFOLD_CIRCLES;
		m->localNormal = v_make(0.0f, 0.0f);
		m->localPoint = P;
		m->id[0] = make_id(1, 0, CF_VERTEX, CF_VERTEX);
		m->p[0] = shape_vert(cB, 0);
		return;
	}
	float den = v_dot(e, e);
	vec2 P = v_scale(1.0f / den, v_add(v_scale(u, A), v_scale(v, B)));
	vec2 d = v_sub(Q, P);
	if (v_dot(d, d) > radius * radius) return;
	vec2 n = v_make(-e.y, e.x);
	if (v_dot(n, v_sub(Q, A)) < 0.0f) n = v_make(-n.x, -n.y);
	v_normalize(&n);
	m->pointCount = 1;
	m->type = MANIFOLD_FACE_A;
	m->localNormal = n;
	m->localPoint = A;
	m->id[0] = make_id(0, 0, CF_FACE, CF_VERTEX);
	m->p[0] = shape_vert(cB, 0);
}

/* b2EPCollider::Collide  b2CollideEdge.cpp:230-618 (+ ComputeEdgeSeparation :620-638, ComputePolygonSeparation :640-690) */
void b2o_collide_edge_polygon(manifold* m, const b2o_shape* edgeA, xform xfA, const b2o_shape* polyB, xform xfB)
{
	xform xf = xf_mul_t(xfA, xfB);
	vec2 centroidB = xf_mul(xf, v_make(polyB->centroid[0], polyB->centroid[1]));
	vec2 v0 = shape_vert(edgeA, 2), v1 = shape_vert(edgeA, 0), v2 = shape_vert(edgeA, 1), v3 = shape_vert(edgeA, 3);
	int has0 = edgeA->count & 1, has3 = edgeA->count & 2;

	vec2 edge1 = v_sub(v2, v1);
	v_normalize(&edge1);
	vec2 normal1 = v_make(edge1.y, -edge1.x);
	float offset1 = v_dot(normal1, v_sub(centroidB, v1));
	float offset0 = 0.0f, offset2 = 0.0f;
	int convex1 = 0, convex2 = 0;
	vec2 normal0 = v_make(0, 0), normal2 = v_make(0, 0);
	if (has0)
	{
		vec2 edge0 = v_sub(v1, v0);
		v_normalize(&edge0);
		normal0 = v_make(edge0.y, -edge0.x);
		convex1 = v_cross(edge0, edge1) >= 0.0f;
		offset0 = v_dot(normal0, v_sub(centroidB, v0));
	}
	if (has3)
	{
		vec2 edge2 = v_sub(v3, v2);
		v_normalize(&edge2);
		normal2 = v_make(edge2.y, -edge2.x);
		convex2 = v_cross(edge1, edge2) > 0.0f;
		offset2 = v_dot(normal2, v_sub(centroidB, v2));
	}

	int front;
	vec2 normal, lower, upper;
	vec2 n1 = normal1, nn1 = v_neg(normal1);
	if (has0 && has3)
	{
		if (convex1 && convex2)
		{
			front = offset0 >= 0.0f || offset1 >= 0.0f || offset2 >= 0.0f;
			if (front) { normal = n1; lower = normal0; upper = normal2; }
			else { normal = nn1; lower = nn1; upper = nn1; }
		}
		else if (convex1)
		{
			front = offset0 >= 0.0f || (offset1 >= 0.0f && offset2 >= 0.0f);
			if (front) { normal = n1; lower = normal0; upper = n1; }
			else { normal = nn1; lower = v_neg(normal2); upper = nn1; }
		}
		else if (convex2)
		{
			front = offset2 >= 0.0f || (offset0 >= 0.0f && offset1 >= 0.0f);
			if (front) { normal = n1; lower = n1; upper = normal2; }
			else { normal = nn1; lower = nn1; upper = v_neg(normal0); }
		}
		else
		{
			front = offset0 >= 0.0f && offset1 >= 0.0f && offset2 >= 0.0f;
			if (front) { normal = n1; lower = n1; upper = n1; }
			else { normal = nn1; lower = v_neg(normal2); upper = v_neg(normal0); }
		}
	}
	else if (has0)
	{
		if (convex1)
		{
			front = offset0 >= 0.0f || offset1 >= 0.0f;
			if (front) { normal = n1; lower = normal0; upper = nn1; }
			else { normal = nn1; lower = n1; upper = nn1; }
		}
		else
		{
			front = offset0 >= 0.0f && offset1 >= 0.0f;
			if (front) { normal = n1; lower = n1; upper = nn1; }
			else { normal = nn1; lower = n1; upper = v_neg(normal0); }
		}
	}
	else if (has3)
	{
		if (convex2)
		{
			front = offset1 >= 0.0f || offset2 >= 0.0f;
			if (front) { normal = n1; lower = nn1; upper = normal2; }
			else { normal = nn1; lower = nn1; upper = n1; }
		}
		else
		{
			front = offset1 >= 0.0f && offset2 >= 0.0f;
			if (front) { normal = n1; lower = nn1; upper = n1; }
			else { normal = nn1; lower = v_neg(normal2); upper = n1; }
		}
	}
	else
	{
		front = offset1 >= 0.0f;
		if (front) { normal = n1; lower = nn1; upper = nn1; }
		else { normal = nn1; lower = n1; upper = n1; }
	}

	vec2 pbV[8], pbN[8];
	int pbCount = polyB->count;
	for (int i = 0; i < pbCount; ++i)
	{
		pbV[i] = xf_mul(xf, shape_vert(polyB, i));
		pbN[i] = r_mul(xf.q, shape_normal(polyB, i));
	}
	float radius = polyB->radius + edgeA->radius;
	m->pointCount = 0;

	float edgeSep = B2O_MAXFLOAT;
	for (int i = 0; i < pbCount; ++i)
	{
		float s = v_dot(normal, v_sub(pbV[i], v1));
		if (s < edgeSep) edgeSep = s;
	}
	if (edgeSep > radius) return;

	int polyKnown = 0, polyIndex = -1;
	float polySep = -B2O_MAXFLOAT;
	vec2 perp = v_make(-normal.y, normal.x);
	for (int i = 0; i < pbCount; ++i)
	{
		vec2 n = v_neg(pbN[i]);
		float s1 = v_dot(n, v_sub(pbV[i], v1));
		float s2 = v_dot(n, v_sub(pbV[i], v2));
		float s = f_min(s1, s2);
		if (s > radius)
		{
			polyKnown = 1;
			polyIndex = i;
			polySep = s;
			break;
		}
		if (v_dot(n, perp) >= 0.0f)
		{
			if (v_dot(v_sub(n, upper), normal) < -B2O_ANGULAR_SLOP) continue;
		}
		else
		{
			if (v_dot(v_sub(n, lower), normal) < -B2O_ANGULAR_SLOP) continue;
		}
		if (s > polySep)
		{
			polyKnown = 1;
			polyIndex = i;
			polySep = s;
		}
	}
	if (polyKnown && polySep > radius) return;

	int primaryEdgeA;
	if (!polyKnown) primaryEdgeA = 1;
	else if (polySep > 0.98f * edgeSep + 0.001f) primaryEdgeA = 0;
	else primaryEdgeA = 1;

	clipv ie[2];
	int rf_i1, rf_i2;
	vec2 rf_v1, rf_v2, rf_normal;
	if (primaryEdgeA)
	{
		m->type = MANIFOLD_FACE_A;
		int best = 0;
		float bestValue = v_dot(normal, pbN[0]);
		for (int i = 1; i < pbCount; ++i)
		{
			float value = v_dot(normal, pbN[i]);
			if (value < bestValue)
			{
				bestValue = value;
				best = i;
			}
		}
		int i1 = best, i2 = i1 + 1 < pbCount ? i1 + 1 : 0;
		ie[0].v = pbV[i1];
		ie[0].id = make_id(0, (uint32_t)i1, CF_FACE, CF_VERTEX);
		ie[1].v = pbV[i2];
		ie[1].id = make_id(0, (uint32_t)i2, CF_FACE, CF_VERTEX);
		if (front) { rf_i1 = 0; rf_i2 = 1; rf_v1 = v1; rf_v2 = v2; rf_normal = normal1; }
		else { rf_i1 = 1; rf_i2 = 0; rf_v1 = v2; rf_v2 = v1; rf_normal = v_neg(normal1); }
	}
	else
	{
		m->type = MANIFOLD_FACE_B;
		ie[0].v = v1;
		ie[0].id = make_id(0, (uint32_t)polyIndex, CF_VERTEX, CF_FACE);
		ie[1].v = v2;
		ie[1].id = make_id(0, (uint32_t)polyIndex, CF_VERTEX, CF_FACE);
		rf_i1 = polyIndex;
		rf_i2 = rf_i1 + 1 < pbCount ? rf_i1 + 1 : 0;
		rf_v1 = pbV[rf_i1];
		rf_v2 = pbV[rf_i2];
		rf_normal = pbN[rf_i1];
	}
	vec2 side1 = v_make(rf_normal.y, -rf_normal.x);
	vec2 side2 = v_neg(side1);
	float sideOffset1 = v_dot(side1, rf_v1);
	float sideOffset2 = v_dot(side2, rf_v2);
	clipv cp1[2], cp2[2];
	int np = clip_segment(cp1, ie, side1, sideOffset1, rf_i1);
	if (np < 2) return;
	np = clip_segment(cp2, cp1, side2, sideOffset2, rf_i2);
	if (np < 2) return;
	if (primaryEdgeA)
	{
		m->localNormal = rf_normal;
		m->localPoint = rf_v1;
	}
	else
	{
		m->localNormal = shape_normal(polyB, rf_i1);
		m->localPoint = shape_vert(polyB, rf_i1);
	}
	int pc = 0;
	for (int i = 0; i < 2; ++i)
	{
		float separation = v_dot(rf_normal, v_sub(cp2[i].v, rf_v1));
		if (separation <= radius)
		{
			if (primaryEdgeA)
			{
				m->p[pc] = xf_mul_tv(xf, cp2[i].v);
				m->id[pc] = cp2[i].id;
			}
			else
			{
				m->p[pc] = cp2[i].v;
				m->id[pc] = swap_id(cp2[i].id);
			}
			++pc;
		}
	}
	m->pointCount = pc;
}

/* Type-pair dispatch of b2Contact::Evaluate overrides (b2Contact.cpp:42-52). */
void b2o_evaluate(manifold* m, const b2o_shape* sA, xform xfA, const b2o_shape* sB, xform xfB)
{
	if (sA->type == SHAPE_POLYGON && sB->type == SHAPE_POLYGON) b2o_collide_polygons(m, sA, xfA, sB, xfB);
	else if (sA->type == SHAPE_POLYGON && sB->type == SHAPE_CIRCLE) b2o_collide_polygon_circle(m, sA, xfA, sB, xfB);
	else if (sA->type == SHAPE_CIRCLE && sB->type == SHAPE_CIRCLE) b2o_collide_circles(m, sA, xfA, sB, xfB);
	/* chain children: b2ChainAndPolygonContact.cpp:45-53, b2ChainAndCircleContact.cpp:45-53 evaluate the child edge */
	else if (SHAPE_IS_SEGMENT(sA->type) && sB->type == SHAPE_POLYGON) b2o_collide_edge_polygon(m, sA, xfA, sB, xfB);
	else if (SHAPE_IS_SEGMENT(sA->type) && sB->type == SHAPE_CIRCLE) b2o_collide_edge_circle(m, sA, xfA, sB, xfB);
	else m->pointCount = 0;
}

void b2o_collide(const b2o_shape* shapeA, const float* xfA3, const b2o_shape* shapeB, const float* xfB3, float* out)
{
	xform a, b;
	a.p = v_make(xfA3[0], xfA3[1]);
	a.q = r_make(xfA3[2]);
	b.p = v_make(xfB3[0], xfB3[1]);
	b.q = r_make(xfB3[2]);
	manifold m;
	memset(&m, 0, sizeof(m));
	b2o_evaluate(&m, shapeA, a, shapeB, b);
	for (int i = 0; i < 16; ++i) out[i] = 0.0f;
	out[0] = (float)m.type;
	out[1] = (float)m.pointCount;
	if (m.pointCount == 0) return;
	out[2] = m.localNormal.x;
	out[3] = m.localNormal.y;
	out[4] = m.localPoint.x;
	out[5] = m.localPoint.y;
	for (int k = 0; k < m.pointCount; ++k)
	{
		float* q = out + 6 + 5 * k;
		q[0] = m.p[k].x;
		q[1] = m.p[k].y;
		memcpy(q + 4, &m.id[k], 4);
	}
}
