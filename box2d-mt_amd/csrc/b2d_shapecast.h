// b2d_shapecast.h - linear cast of one convex proxy against another (b2ShapeCast, b2Distance.cpp:605-745 of the reference:
// van den Bergen's GJK ray cast). Not on the Step() path - the drop-in host API offers it to user code
// (Testbed/Tests/ShapeCast.h) - but it shares the simplex machinery of b2d_toi.h, so it lives beside it.
//
// Shape B travels along `travel` while A rests. The cast keeps a clip parameter lambda in [0, 1] and a simplex of the
// Minkowski difference (B shifted by lambda * travel) - A; each round takes the support point against the current closest
// vector v, clips lambda against the plane through it, and re-solves the simplex for the new closest vector, until |v|
// is the target distance sigma (the two skins just touch). Tolerances and operand order follow the reference so that the
// host API returns the reference's floats.
#ifndef B2D_SHAPECAST_H
#define B2D_SHAPECAST_H

#include "b2d_toi.h"

struct ShapeCastResult
{
	V2 point, normal;
	float lambda;
	int iterations;
};

// closest point of the simplex to the origin (b2Simplex::GetClosestPoint, b2Distance.cpp:196-212)
B2D_HD V2 b2dSimplexClosest(const Simplex& s)
{
	if (s.count == 1) return s.v1.w;
	if (s.count == 2) return s.v1.a * s.v1.w + s.v2.a * s.v2.w;
	return v2(0.0f, 0.0f);
}

B2D_HD bool b2dShapeCast(ShapeCastResult* out, const GjkProxy& pA, Xf xfA, const GjkProxy& pB, Xf xfB, V2 travel)
{
	out->iterations = 0;
	out->lambda = 1.0f;
	out->normal = v2(0.0f, 0.0f);
	out->point = v2(0.0f, 0.0f);
	const float skinA = b2dMax(pA.radius, B2D_POLYGON_RADIUS), skinB = b2dMax(pB.radius, B2D_POLYGON_RADIUS);
	const float skins = skinA + skinB;
	const float sigma = b2dMax(B2D_POLYGON_RADIUS, skins - B2D_POLYGON_RADIUS);
	const float tolerance = 0.5f * B2D_LINEAR_SLOP;
	V2 n = v2(0.0f, 0.0f);
	float lambda = 0.0f;
	Simplex s;
	memset(&s, 0, sizeof(s));
	// start from the pair of points that face each other along the travel
	int ia = b2dSupport(pA, b2dMulTRV(xfA.q, -travel));
	V2 wA = b2dMulXV(xfA, pA.verts[ia]);
	int ib = b2dSupport(pB, b2dMulTRV(xfB.q, travel));
	V2 wB = b2dMulXV(xfB, pB.verts[ib]);
	V2 v = wA - wB;
	int iter = 0;
	while (iter < 20 && b2dAbs(b2dLength(v) - sigma) > tolerance)
	{
		out->iterations += 1;
		ia = b2dSupport(pA, b2dMulTRV(xfA.q, -v));
		wA = b2dMulXV(xfA, pA.verts[ia]);
		ib = b2dSupport(pB, b2dMulTRV(xfB.q, v));
		wB = b2dMulXV(xfB, pB.verts[ib]);
		const V2 p = wA - wB;
		b2dNormalize(v); // (-v is the plane normal at p)
		const float vp = b2dDot(v, p), vr = b2dDot(v, travel);
		if (vp - sigma > lambda * vr)
		{
			if (vr <= 0.0f) return false; // moving away from the plane: never reaches it
			lambda = (vp - sigma) / vr;
			if (lambda > 1.0f) return false;
			n = -v;
			s.count = 0;
		}
		// the simplex works with B - A, B taken at the clip point (the plane itself stays in unshifted space)
		SimplexVertex& nv = s.count == 0 ? s.v1 : (s.count == 1 ? s.v2 : s.v3);
		nv.indexA = ib;
		nv.wA = wB + lambda * travel;
		nv.indexB = ia;
		nv.wB = wA;
		nv.w = nv.wB - nv.wA;
		nv.a = 1.0f;
		s.count += 1;
		if (s.count == 2) b2dSimplexSolve2(s);
		else if (s.count == 3) b2dSimplexSolve3(s);
		if (s.count == 3) return false; // the origin is inside: overlap
		v = b2dSimplexClosest(s);
		++iter;
	}
	// witness points: the simplex's first side is shape B (at the clip point), its second side shape A
	V2 onA;
	if (s.count == 1) onA = s.v1.wB;
	else if (s.count == 2) onA = s.v1.a * s.v1.wB + s.v2.a * s.v2.wB;
	else if (s.count == 3) onA = (s.v1.a * s.v1.wA + s.v2.a * s.v2.wA) + s.v3.a * s.v3.wA;
	else onA = v2(0.0f, 0.0f);
	if (b2dDot(v, v) > 0.0f)
	{
		n = -v;
		b2dNormalize(n);
	}
	out->point = onA + skinA * n;
	out->normal = n;
	out->lambda = lambda;
	out->iterations = iter;
	return true;
}

#endif
