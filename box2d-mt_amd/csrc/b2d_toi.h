// b2d_toi.h - continuous collision math of the MI355X Step() path: GJK distance and the
// conservative-advancement time of impact, one lane per contact.
//
// Same operand order as the reference (one fp32 rounding per operation, -ffp-contract=off):
//   b2Distance       Box2D/Collision/b2Distance.cpp:103-604
//   b2TimeOfImpact   Box2D/Collision/b2TimeOfImpact.cpp:45-486
//   b2Sweep          Box2D/Common/b2Math.h:679-705
// Everything lives in registers / scalar locals: a simplex is 3 vertices x 7 words and the proxies
// point straight at the ShapeRec vertex arrays in HBM (read through the scalar / L1 path; the
// records are shared by every contact of a shape, so they stay cache resident).
// Compiled by hipcc for the kernels and by g++ for the CPU probe (tests/probe/host_probe.cpp).
#ifndef B2D_TOI_H
#define B2D_TOI_H

#include "b2d_collide.h"

#define B2D_MAX_SUB_STEPS 8
#define B2D_MAX_TOI_CONTACTS 32
#define B2D_MAX_TOI_BODIES 64

// b2Sweep  b2Math.h:362-382
struct Sweep
{
	V2 localCenter, c0, c;
	float a0, a, alpha0;
};

// b2Sweep::GetTransform  b2Math.h:679-687
B2D_HD Xf b2dSweepXf(const Sweep& s, float beta)
{
	Xf xf;
	const float om = 1.0f - beta;
	xf.p = om * s.c0 + beta * s.c;
	const float angle = om * s.a0 + beta * s.a;
	xf.q = b2dRot(angle);
	xf.p -= b2dMulRV(xf.q, s.localCenter);
	return xf;
}

// b2Sweep::Advance  b2Math.h:689-696
B2D_HD void b2dSweepAdvance(Sweep& s, float alpha)
{
	const float beta = (alpha - s.alpha0) / (1.0f - s.alpha0);
	s.c0 += beta * (s.c - s.c0);
	s.a0 += beta * (s.a - s.a0);
	s.alpha0 = alpha;
}

// b2Sweep::Normalize  b2Math.h:699-705
B2D_HD void b2dSweepNormalize(Sweep& s)
{
	const float twoPi = 2.0f * B2D_PI;
	const float d = twoPi * __builtin_floorf(s.a0 / twoPi);
	s.a0 -= d;
	s.a -= d;
}

// b2DistanceProxy  b2Distance.h:30-57 / b2Distance.cpp:31-83: circle = 1 vertex, edge = 2, polygon = count
struct GjkProxy
{
	const V2* verts;
	int count;
	float radius;
};

B2D_HD GjkProxy b2dProxy(const ShapeRec* s)
{
	GjkProxy p;
	p.verts = s->verts;
	p.radius = s->radius;
	p.count = s->type == B2D_SHAPE_CIRCLE ? 1 : (B2D_IS_SEGMENT(s->type) ? 2 : s->count);
	return p;
}

// b2DistanceProxy::GetSupport  b2Distance.h:132-147: the first maximum wins
B2D_HD int b2dSupport(const GjkProxy& p, V2 d)
{
	int best = 0;
	float bestValue = b2dDot(p.verts[0], d);
	for (int i = 1; i < p.count; ++i)
	{
		const float value = b2dDot(p.verts[i], d);
		if (value > bestValue)
		{
			best = i;
			bestValue = value;
		}
	}
	return best;
}

// b2SimplexCache  b2Distance.h:61-67
struct GjkCache
{
	float metric;
	int count;
	int indexA[3], indexB[3];
};

struct SimplexVertex
{
	V2 wA, wB, w;
	float a;
	int indexA, indexB;
};

struct Simplex
{
	SimplexVertex v1, v2, v3;
	int count;
};

B2D_HD void b2dSimplexVertex(SimplexVertex& v, const GjkProxy& pA, Xf xfA, int ia, const GjkProxy& pB, Xf xfB, int ib)
{
	v.indexA = ia;
	v.indexB = ib;
	v.wA = b2dMulXV(xfA, pA.verts[ia]);
	v.wB = b2dMulXV(xfB, pB.verts[ib]);
	v.w = v.wB - v.wA;
}

// b2Simplex::GetMetric  b2Distance.cpp:241-264
B2D_HD float b2dSimplexMetric(const Simplex& s)
{
	if (s.count == 2) return b2dLength(s.v1.w - s.v2.w);
	if (s.count == 3) return b2dCross(s.v2.w - s.v1.w, s.v3.w - s.v1.w);
	return 0.0f;
}

// b2Simplex::Solve2  b2Distance.cpp:297-330
B2D_HD void b2dSimplexSolve2(Simplex& s)
{
	const V2 w1 = s.v1.w, w2 = s.v2.w;
	const V2 e12 = w2 - w1;
	const float d12_2 = -b2dDot(w1, e12);
	if (d12_2 <= 0.0f)
	{
		s.v1.a = 1.0f;
		s.count = 1;
		return;
	}
	const float d12_1 = b2dDot(w2, e12);
	if (d12_1 <= 0.0f)
	{
		s.v2.a = 1.0f;
		s.count = 1;
		s.v1 = s.v2;
		return;
	}
	const float inv = 1.0f / (d12_1 + d12_2);
	s.v1.a = d12_1 * inv;
	s.v2.a = d12_2 * inv;
	s.count = 2;
}

// b2Simplex::Solve3  b2Distance.cpp:338-440
B2D_HD void b2dSimplexSolve3(Simplex& s)
{
	const V2 w1 = s.v1.w, w2 = s.v2.w, w3 = s.v3.w;
	const V2 e12 = w2 - w1;
	const float d12_1 = b2dDot(w2, e12);
	const float d12_2 = -b2dDot(w1, e12);
	const V2 e13 = w3 - w1;
	const float d13_1 = b2dDot(w3, e13);
	const float d13_2 = -b2dDot(w1, e13);
	const V2 e23 = w3 - w2;
	const float d23_1 = b2dDot(w3, e23);
	const float d23_2 = -b2dDot(w2, e23);
	const float n123 = b2dCross(e12, e13);
	const float d123_1 = n123 * b2dCross(w2, w3);
	const float d123_2 = n123 * b2dCross(w3, w1);
	const float d123_3 = n123 * b2dCross(w1, w2);
	if (d12_2 <= 0.0f && d13_2 <= 0.0f)
	{
		s.v1.a = 1.0f;
		s.count = 1;
		return;
	}
	if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f)
	{
		const float inv = 1.0f / (d12_1 + d12_2);
		s.v1.a = d12_1 * inv;
		s.v2.a = d12_2 * inv;
		s.count = 2;
		return;
	}
	if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f)
	{
		const float inv = 1.0f / (d13_1 + d13_2);
		s.v1.a = d13_1 * inv;
		s.v3.a = d13_2 * inv;
		s.count = 2;
		s.v2 = s.v3;
		return;
	}
	if (d12_1 <= 0.0f && d23_2 <= 0.0f)
	{
		s.v2.a = 1.0f;
		s.count = 1;
		s.v1 = s.v2;
		return;
	}
	if (d13_1 <= 0.0f && d23_1 <= 0.0f)
	{
		s.v3.a = 1.0f;
		s.count = 1;
		s.v1 = s.v3;
		return;
	}
	if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f)
	{
		const float inv = 1.0f / (d23_1 + d23_2);
		s.v2.a = d23_1 * inv;
		s.v3.a = d23_2 * inv;
		s.count = 2;
		s.v1 = s.v3;
		return;
	}
	const float inv = 1.0f / (d123_1 + d123_2 + d123_3);
	s.v1.a = d123_1 * inv;
	s.v2.a = d123_2 * inv;
	s.v3.a = d123_3 * inv;
	s.count = 3;
}

struct GjkOutput
{
	V2 pointA, pointB;
	float distance;
	int iterations;
};

// b2Distance  b2Distance.cpp:444-604 (ReadCache :105-158, WriteCache :160-170, GetSearchDirection :172-199,
// GetWitnessPoints :214-239 inlined)
B2D_HD void b2dDistance(GjkOutput& out, GjkCache& cache, const GjkProxy& pA, Xf xfA, const GjkProxy& pB, Xf xfB, bool useRadii)
{
	Simplex s;
	memset(&s, 0, sizeof(s));
	s.count = cache.count;
	if (s.count > 0) b2dSimplexVertex(s.v1, pA, xfA, cache.indexA[0], pB, xfB, cache.indexB[0]);
	if (s.count > 1) b2dSimplexVertex(s.v2, pA, xfA, cache.indexA[1], pB, xfB, cache.indexB[1]);
	if (s.count > 2) b2dSimplexVertex(s.v3, pA, xfA, cache.indexA[2], pB, xfB, cache.indexB[2]);
	if (s.count > 1)
	{
		const float metric1 = cache.metric;
		const float metric2 = b2dSimplexMetric(s);
		if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < B2D_EPSILON) s.count = 0;
	}
	if (s.count == 0)
	{
		b2dSimplexVertex(s.v1, pA, xfA, 0, pB, xfB, 0);
		s.v1.a = 1.0f;
		s.count = 1;
	}
	int saveA[3], saveB[3];
	int iter = 0;
	while (iter < 20)
	{
		const int saveCount = s.count;
		saveA[0] = s.v1.indexA; saveB[0] = s.v1.indexB;
		saveA[1] = s.v2.indexA; saveB[1] = s.v2.indexB;
		saveA[2] = s.v3.indexA; saveB[2] = s.v3.indexB;
		if (s.count == 2) b2dSimplexSolve2(s);
		else if (s.count == 3) b2dSimplexSolve3(s);
		if (s.count == 3) break;
		V2 d;
		if (s.count == 1)
		{
			d = -s.v1.w;
		}
		else
		{
			const V2 e12 = s.v2.w - s.v1.w;
			const float sgn = b2dCross(e12, -s.v1.w);
			d = sgn > 0.0f ? b2dCrossSV(1.0f, e12) : b2dCrossVS(e12, 1.0f);
		}
		if (b2dDot(d, d) < B2D_EPSILON * B2D_EPSILON) break;
		const int ia = b2dSupport(pA, b2dMulTRV(xfA.q, -d));
		const int ib = b2dSupport(pB, b2dMulTRV(xfB.q, d));
		SimplexVertex nv;
		nv.a = 0.0f;
		b2dSimplexVertex(nv, pA, xfA, ia, pB, xfB, ib);
		if (s.count == 1) s.v2 = nv; else s.v3 = nv;
		++iter;
		bool duplicate = false;
		for (int i = 0; i < saveCount; ++i)
		{
			if (ia == saveA[i] && ib == saveB[i]) duplicate = true;
		}
		if (duplicate) break;
		++s.count;
	}
	if (s.count == 1)
	{
		out.pointA = s.v1.wA;
		out.pointB = s.v1.wB;
	}
	else if (s.count == 2)
	{
		out.pointA = s.v1.a * s.v1.wA + s.v2.a * s.v2.wA;
		out.pointB = s.v1.a * s.v1.wB + s.v2.a * s.v2.wB;
	}
	else
	{
		out.pointA = (s.v1.a * s.v1.wA + s.v2.a * s.v2.wA) + s.v3.a * s.v3.wA;
		out.pointB = out.pointA;
	}
	out.distance = b2dLength(out.pointA - out.pointB);
	out.iterations = iter;
	cache.metric = b2dSimplexMetric(s);
	cache.count = s.count;
	cache.indexA[0] = s.v1.indexA; cache.indexB[0] = s.v1.indexB;
	cache.indexA[1] = s.v2.indexA; cache.indexB[1] = s.v2.indexB;
	cache.indexA[2] = s.v3.indexA; cache.indexB[2] = s.v3.indexB;
	if (useRadii)
	{
		const float rA = pA.radius, rB = pB.radius;
		if (out.distance > rA + rB && out.distance > B2D_EPSILON)
		{
			out.distance -= rA + rB;
			V2 normal = out.pointB - out.pointA;
			b2dNormalize(normal);
			out.pointA += rA * normal;
			out.pointB -= rB * normal;
		}
		else
		{
			const V2 p = 0.5f * (out.pointA + out.pointB);
			out.pointA = p;
			out.pointB = p;
			out.distance = 0.0f;
		}
	}
}

// ---- b2SeparationFunction  b2TimeOfImpact.cpp:36-251 ------------------------------------------------
#define SEP_POINTS 0
#define SEP_FACE_A 1
#define SEP_FACE_B 2

struct SepFn
{
	int type;
	V2 localPoint, axis;
};

// Initialize :45-129
B2D_HD void b2dSepInit(SepFn& f, const GjkCache& cache, const GjkProxy& pA, const Sweep& sweepA, const GjkProxy& pB,
	const Sweep& sweepB, float t1)
{
	const Xf xfA = b2dSweepXf(sweepA, t1);
	const Xf xfB = b2dSweepXf(sweepB, t1);
	f.localPoint = v2(0.0f, 0.0f);
	if (cache.count == 1)
	{
		f.type = SEP_POINTS;
		const V2 pointA = b2dMulXV(xfA, pA.verts[cache.indexA[0]]);
		const V2 pointB = b2dMulXV(xfB, pB.verts[cache.indexB[0]]);
		f.axis = pointB - pointA;
		b2dNormalize(f.axis);
	}
	else if (cache.indexA[0] == cache.indexA[1])
	{
		f.type = SEP_FACE_B;
		const V2 b1 = pB.verts[cache.indexB[0]];
		const V2 b2 = pB.verts[cache.indexB[1]];
		f.axis = b2dCrossVS(b2 - b1, 1.0f);
		b2dNormalize(f.axis);
		const V2 normal = b2dMulRV(xfB.q, f.axis);
		f.localPoint = 0.5f * (b1 + b2);
		const V2 pointB = b2dMulXV(xfB, f.localPoint);
		const V2 pointA = b2dMulXV(xfA, pA.verts[cache.indexA[0]]);
		const float s = b2dDot(pointA - pointB, normal);
		if (s < 0.0f) f.axis = -f.axis;
	}
	else
	{
		f.type = SEP_FACE_A;
		const V2 a1 = pA.verts[cache.indexA[0]];
		const V2 a2 = pA.verts[cache.indexA[1]];
		f.axis = b2dCrossVS(a2 - a1, 1.0f);
		b2dNormalize(f.axis);
		const V2 normal = b2dMulRV(xfA.q, f.axis);
		f.localPoint = 0.5f * (a1 + a2);
		const V2 pointA = b2dMulXV(xfA, f.localPoint);
		const V2 pointB = b2dMulXV(xfB, pB.verts[cache.indexB[0]]);
		const float s = b2dDot(pointB - pointA, normal);
		if (s < 0.0f) f.axis = -f.axis;
	}
}

// Evaluate :193-245 on given transforms
B2D_HD float b2dSepEvalXf(const SepFn& f, const GjkProxy& pA, Xf xfA, const GjkProxy& pB, Xf xfB, int indexA, int indexB)
{
	if (f.type == SEP_POINTS)
	{
		const V2 pointA = b2dMulXV(xfA, pA.verts[indexA]);
		const V2 pointB = b2dMulXV(xfB, pB.verts[indexB]);
		return b2dDot(pointB - pointA, f.axis);
	}
	if (f.type == SEP_FACE_A)
	{
		const V2 normal = b2dMulRV(xfA.q, f.axis);
		const V2 pointA = b2dMulXV(xfA, f.localPoint);
		const V2 pointB = b2dMulXV(xfB, pB.verts[indexB]);
		return b2dDot(pointB - pointA, normal);
	}
	const V2 normal = b2dMulRV(xfB.q, f.axis);
	const V2 pointB = b2dMulXV(xfB, f.localPoint);
	const V2 pointA = b2dMulXV(xfA, pA.verts[indexA]);
	return b2dDot(pointA - pointB, normal);
}

// FindMinSeparation :132-190
B2D_HD float b2dSepFindMin(const SepFn& f, const GjkProxy& pA, const Sweep& sweepA, const GjkProxy& pB, const Sweep& sweepB,
	int* indexA, int* indexB, float t)
{
	const Xf xfA = b2dSweepXf(sweepA, t);
	const Xf xfB = b2dSweepXf(sweepB, t);
	if (f.type == SEP_POINTS)
	{
		*indexA = b2dSupport(pA, b2dMulTRV(xfA.q, f.axis));
		*indexB = b2dSupport(pB, b2dMulTRV(xfB.q, -f.axis));
	}
	else if (f.type == SEP_FACE_A)
	{
		const V2 normal = b2dMulRV(xfA.q, f.axis);
		*indexA = -1;
		*indexB = b2dSupport(pB, b2dMulTRV(xfB.q, -normal));
	}
	else
	{
		const V2 normal = b2dMulRV(xfB.q, f.axis);
		*indexB = -1;
		*indexA = b2dSupport(pA, b2dMulTRV(xfA.q, -normal));
	}
	return b2dSepEvalXf(f, pA, xfA, pB, xfB, *indexA, *indexB);
}

B2D_HD float b2dSepEval(const SepFn& f, const GjkProxy& pA, const Sweep& sweepA, const GjkProxy& pB, const Sweep& sweepB,
	int indexA, int indexB, float t)
{
	return b2dSepEvalXf(f, pA, b2dSweepXf(sweepA, t), pB, b2dSweepXf(sweepB, t), indexA, indexB);
}

// b2TOIOutput::State  b2TimeOfImpact.h:39-49
#define TOI_UNKNOWN 0
#define TOI_FAILED 1
#define TOI_OVERLAPPED 2
#define TOI_TOUCHING 3
#define TOI_SEPARATED 4

// b2TimeOfImpact  b2TimeOfImpact.cpp:253-486. Returns the state, *tOut = output.t
B2D_HD int b2dTimeOfImpact(float* tOut, const GjkProxy& pA, const Sweep& sweepAIn, const GjkProxy& pB, const Sweep& sweepBIn, float tMax)
{
	int state = TOI_UNKNOWN;
	float tResult = tMax;
	Sweep sweepA = sweepAIn, sweepB = sweepBIn;
	b2dSweepNormalize(sweepA);
	b2dSweepNormalize(sweepB);
	const float totalRadius = pA.radius + pB.radius;
	const float target = b2dMax(B2D_LINEAR_SLOP, totalRadius - 3.0f * B2D_LINEAR_SLOP);
	const float tolerance = 0.25f * B2D_LINEAR_SLOP;
	float t1 = 0.0f;
	int iter = 0;
	GjkCache cache;
	cache.count = 0;
	cache.metric = 0.0f;
	for (int k = 0; k < 3; ++k) cache.indexA[k] = cache.indexB[k] = 0;
	for (;;)
	{
		const Xf xfA = b2dSweepXf(sweepA, t1);
		const Xf xfB = b2dSweepXf(sweepB, t1);
		GjkOutput dist;
		b2dDistance(dist, cache, pA, xfA, pB, xfB, false);
		if (dist.distance <= 0.0f)
		{
			state = TOI_OVERLAPPED;
			tResult = 0.0f;
			break;
		}
		if (dist.distance < target + tolerance)
		{
			state = TOI_TOUCHING;
			tResult = t1;
			break;
		}
		SepFn fcn;
		b2dSepInit(fcn, cache, pA, sweepA, pB, sweepB, t1);
		bool done = false;
		float t2 = tMax;
		int pushBackIter = 0;
		for (;;)
		{
			int indexA, indexB;
			float s2 = b2dSepFindMin(fcn, pA, sweepA, pB, sweepB, &indexA, &indexB, t2);
			if (s2 > target + tolerance)
			{
				state = TOI_SEPARATED;
				tResult = tMax;
				done = true;
				break;
			}
			if (s2 > target - tolerance)
			{
				t1 = t2;
				break;
			}
			float s1 = b2dSepEval(fcn, pA, sweepA, pB, sweepB, indexA, indexB, t1);
			if (s1 < target - tolerance)
			{
				state = TOI_FAILED;
				tResult = t1;
				done = true;
				break;
			}
			if (s1 <= target + tolerance)
			{
				state = TOI_TOUCHING;
				tResult = t1;
				done = true;
				break;
			}
			int rootIter = 0;
			float a1 = t1, a2 = t2;
			for (;;)
			{
				float t;
				if (rootIter & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
				else t = 0.5f * (a1 + a2);
				++rootIter;
				const float s = b2dSepEval(fcn, pA, sweepA, pB, sweepB, indexA, indexB, t);
				if (b2dAbs(s - target) < tolerance)
				{
					t2 = t;
					break;
				}
				if (s > target)
				{
					a1 = t;
					s1 = s;
				}
				else
				{
					a2 = t;
					s2 = s;
				}
				if (rootIter == 50) break;
			}
			++pushBackIter;
			if (pushBackIter == B2D_MAX_POLY_VERTS) break;
		}
		++iter;
		if (done) break;
		if (iter == 20)
		{
			state = TOI_FAILED;
			tResult = t1;
			break;
		}
	}
	*tOut = tResult;
	return state;
}

#endif
