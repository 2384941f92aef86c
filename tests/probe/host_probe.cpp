// CPU compile of the device math headers (box2d-mt_amd/csrc/b2d_*.h) for bit-level checks against
// oracle/_ref WITHOUT a GPU.  TEST INFRASTRUCTURE: nothing in the product links this file.
#include "b2d_solver.h"
#include "b2d_toi.h"
#include <math.h>

extern "C"
{

void probe_sincos(int n, const float* a, float* s, float* c)
{
	for (int i = 0; i < n; ++i)
	{
		s[i] = b2dSin(a[i]);
		c[i] = b2dCos(a[i]);
	}
}

// exhaustive-ish check against this machine's libm over [lo, hi] bit patterns; returns mismatches
long probe_sincos_vs_libm(unsigned lo, unsigned hi, unsigned stride)
{
	long bad = 0;
	for (unsigned long u = lo; u <= hi; u += stride)
	{
		float f = b2dAsFloat((unsigned)u);
		if (f != f || f - f != 0.0f) continue;
		if (b2dAsUint(sinf(f)) != b2dAsUint(b2dSin(f))) ++bad;
		if (b2dAsUint(cosf(f)) != b2dAsUint(b2dCos(f))) ++bad;
		// the fused pair (one argument reduction for both) is what b2Rot::Set uses on the device
		float s2, c2;
		b2dSinCos(f, &s2, &c2);
		if (b2dAsUint(s2) != b2dAsUint(sinf(f)) || b2dAsUint(c2) != b2dAsUint(cosf(f))) ++bad;
	}
	return bad;
}

// shape = ShapeRec as 38 floats/ints: type, count, radius, pad, centroid(2), verts(16), normals(16)
// xf = px, py, angle ; out = 16 floats in the harness manifold layout
void probe_collide(const void* shapeA, const float* xfA, const void* shapeB, const float* xfB, float* out)
{
	const ShapeRec* sA = (const ShapeRec*)shapeA;
	const ShapeRec* sB = (const ShapeRec*)shapeB;
	Xf a, b;
	a.p = v2(xfA[0], xfA[1]);
	a.q = b2dRot(xfA[2]);
	b.p = v2(xfB[0], xfB[1]);
	b.q = b2dRot(xfB[2]);
	Manifold m;
	memset(&m, 0, sizeof(m));
	b2dEvaluate(&m, sA, a, sB, b);
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

void probe_shape_aabb(const void* shape, const float* xf, float* out4)
{
	Xf a;
	a.p = v2(xf[0], xf[1]);
	a.q = b2dRot(xf[2]);
	AABB r = b2dShapeAABB((const ShapeRec*)shape, a);
	out4[0] = r.lo.x;
	out4[1] = r.lo.y;
	out4[2] = r.hi.x;
	out4[3] = r.hi.y;
}

// Same layouts as the harness probes b2h_probe_distance / b2h_probe_toi (box2d-mt_amd/harness/harness.cpp).
static Sweep SweepFrom9(const float* s9)
{
	Sweep s;
	s.localCenter = v2(s9[0], s9[1]);
	s.c0 = v2(s9[2], s9[3]);
	s.c = v2(s9[4], s9[5]);
	s.a0 = s9[6];
	s.a = s9[7];
	s.alpha0 = s9[8];
	return s;
}

void probe_distance(int countA, const float* vertsA, float radiusA, const float* xfA, int countB, const float* vertsB,
	float radiusB, const float* xfB, int useRadii, float* out6)
{
	GjkProxy pA = { (const V2*)vertsA, countA, radiusA }, pB = { (const V2*)vertsB, countB, radiusB };
	Xf a, b;
	a.p = v2(xfA[0], xfA[1]);
	a.q = b2dRot(xfA[2]);
	b.p = v2(xfB[0], xfB[1]);
	b.q = b2dRot(xfB[2]);
	GjkCache cache;
	memset(&cache, 0, sizeof(cache));
	GjkOutput out;
	b2dDistance(out, cache, pA, a, pB, b, useRadii != 0);
	out6[0] = out.pointA.x; out6[1] = out.pointA.y;
	out6[2] = out.pointB.x; out6[3] = out.pointB.y;
	out6[4] = out.distance;
	out6[5] = (float)out.iterations;
}

// ownIdBlock (b2d_math.h) against the integer remainder: every block count up to `nbMax`, every `stride`-th body id below
// `bodies` plus the ids whose hash lies at the very top of its 24 bits; returns the number of mismatches
long probe_own_id_block_check(int nbMax, unsigned bodies, unsigned stride)
{
	long bad = 0;
	for (int nb = 1; nb <= nbMax; ++nb)
	{
		for (unsigned body = 0; body < bodies; body += stride)
		{
			const unsigned x = body * 2654435761u >> 8;
			if (ownIdBlock((int)body, nb) != 1 + (int)(x % (unsigned)nb)) ++bad;
		}
	}
	// (x * rcp(nb) rounds up past an integer only near the top of the 24 bits: all ids whose hash is there)
	for (unsigned body = 0; body < bodies; ++body)
	{
		const unsigned x = body * 2654435761u >> 8;
		if (x < 0xf00000u) continue;
		for (int nb = 1; nb <= nbMax; ++nb) if (ownIdBlock((int)body, nb) != 1 + (int)(x % (unsigned)nb)) ++bad;
	}
	return bad;
}

void probe_toi(int countA, const float* vertsA, float radiusA, const float* sweepA9, int countB, const float* vertsB,
	float radiusB, const float* sweepB9, float tMax, float* out2)
{
	GjkProxy pA = { (const V2*)vertsA, countA, radiusA }, pB = { (const V2*)vertsB, countB, radiusB };
	Sweep sA = SweepFrom9(sweepA9), sB = SweepFrom9(sweepB9);
	float t = 0.0f;
	int state = b2dTimeOfImpact(&t, pA, sA, pB, sB, tMax);
	out2[0] = (float)state;
	out2[1] = t;
}

} // extern "C"
