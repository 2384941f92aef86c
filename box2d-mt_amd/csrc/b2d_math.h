// b2d_math.h - scalar/vector primitives of the MI355X Step() path.
//
// Every function evaluates its expression in the SAME operand order as the reference primitive it
// stands for (Box2D/Common/b2Math.h, cited per function), with one fp32 rounding per operation:
// the translation unit is built with -ffp-contract=off and IEEE divide/sqrt, so a lane that walks
// an island in the reference's order reproduces the reference's floats bit for bit.
//
// B2D_HD expands to __host__ __device__ under hipcc; the CPU test shim (tests/host_probe.cpp)
// compiles the same header with g++ to pin the math against oracle/_ref without a GPU.
#ifndef B2D_MATH_H
#define B2D_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define B2D_HD __host__ __device__ __forceinline__
#define B2D_D __device__ __forceinline__
#else
#define B2D_HD static inline
#define B2D_D static inline
#endif

// --- tuning constants (Box2D/Common/b2Settings.h:47-174) -------------------------------------
// (in this header so that the CPU build can check it exhaustively: tests/test_device_math_cpu.py)
// The block (+ 1) a body takes by its own id: hash(body) mod blocks. The remainder goes by way of a float quotient with BOTH
// corrections, not through `%`: with a divisor the compiler knows to be small it expands `%` into its 24-bit form, and
// inside k_block_census that form returned 0xffffff for x = 0xc1f9f3, nb = 11 (a quotient one too large, the remainder - 1,
// uncorrected) where the same expression in k_color_check and k_color_fill returned 10 - one body of 622 was home in no
// block, its neighbour's cut constraint waited for a hand-over that never came (found with B2HIP_HANDOVER_WHY, round 5).
B2D_HD int ownIdBlock(int body, int nb)
{
	const uint32_t x = (uint32_t)body * 2654435761u >> 8; // (< 2^24: exact as a float)
	const uint32_t q = (uint32_t)((float)x / (float)nb);
	int r = (int)x - (int)(q * (uint32_t)nb);
	if (r < 0) r += nb; else if (r >= nb) r -= nb;
	return 1 + r;
}

#define B2D_EPSILON 1.192092896e-07f
#define B2D_MAXFLOAT 3.402823466e+38f
#define B2D_PI 3.14159265359f
#define B2D_MAX_MANIFOLD_POINTS 2
#define B2D_MAX_POLY_VERTS 8
#define B2D_AABB_EXTENSION 0.1f
#define B2D_AABB_MULTIPLIER 2.0f
#define B2D_LINEAR_SLOP 0.005f
#define B2D_ANGULAR_SLOP (2.0f / 180.0f * B2D_PI)
#define B2D_POLYGON_RADIUS (2.0f * B2D_LINEAR_SLOP)
#define B2D_VELOCITY_THRESHOLD 1.0f
#define B2D_MAX_LINEAR_CORRECTION 0.2f
#define B2D_MAX_TRANSLATION 2.0f
#define B2D_MAX_TRANSLATION_SQ (B2D_MAX_TRANSLATION * B2D_MAX_TRANSLATION)
#define B2D_MAX_ROTATION (0.5f * B2D_PI)
#define B2D_MAX_ROTATION_SQ (B2D_MAX_ROTATION * B2D_MAX_ROTATION)
#define B2D_BAUMGARTE 0.2f
#define B2D_TOI_BAUMGARTE 0.75f
#define B2D_TIME_TO_SLEEP 0.5f
#define B2D_LINEAR_SLEEP_TOL 0.01f
#define B2D_ANGULAR_SLEEP_TOL (2.0f / 180.0f * B2D_PI)

struct V2
{
	float x, y;
};

struct Rot
{
	float s, c;
};

struct Xf
{
	V2 p;
	Rot q;
};

B2D_HD V2 v2(float x, float y)
{
	V2 r;
	r.x = x;
	r.y = y;
	return r;
}

// b2Math.h:602-644 - the ternary forms, not fminf/fmaxf/fabsf (NaN and signed-zero behaviour).
B2D_HD float b2dMin(float a, float b) { return a < b ? a : b; }
B2D_HD float b2dMax(float a, float b) { return a > b ? a : b; }
B2D_HD float b2dAbs(float a) { return a > 0.0f ? a : -a; }
B2D_HD float b2dClamp(float a, float lo, float hi) { return b2dMax(lo, b2dMin(a, hi)); }
B2D_HD V2 b2dMinV(V2 a, V2 b) { return v2(b2dMin(a.x, b.x), b2dMin(a.y, b.y)); }
B2D_HD V2 b2dMaxV(V2 a, V2 b) { return v2(b2dMax(a.x, b.x), b2dMax(a.y, b.y)); }

B2D_HD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
B2D_HD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
B2D_HD V2 operator-(V2 a) { return v2(-a.x, -a.y); }
B2D_HD V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
B2D_HD void operator+=(V2& a, V2 b) { a.x += b.x; a.y += b.y; }
B2D_HD void operator-=(V2& a, V2 b) { a.x -= b.x; a.y -= b.y; }
B2D_HD void operator*=(V2& a, float s) { a.x *= s; a.y *= s; }

// b2Math.h:388-411
B2D_HD float b2dDot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
B2D_HD float b2dCross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
B2D_HD V2 b2dCrossVS(V2 a, float s) { return v2(s * a.y, -s * a.x); }
B2D_HD V2 b2dCrossSV(float s, V2 a) { return v2(-s * a.y, s * a.x); }

// Correctly rounded sqrt on both sides: on gfx950 __builtin_sqrtf lowers to v_sqrt_f32 plus the FMA
// fix-up sequence (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt). NOT __fsqrt_rn, which HIP
// maps to the approximate native sqrt unless OCML_BASIC_ROUNDED_OPERATIONS is defined.
B2D_HD float b2dSqrt(float x)
{
	return __builtin_sqrtf(x);
}

// b2Vec2::Length / Normalize (b2Math.h:84-108): below epsilon the vector is left untouched.
B2D_HD float b2dLength(V2 a) { return b2dSqrt(a.x * a.x + a.y * a.y); }
B2D_HD float b2dNormalize(V2& a)
{
	float length = b2dLength(a);
	if (length < B2D_EPSILON)
	{
		return 0.0f;
	}
	float invLength = 1.0f / length;
	a.x *= invLength;
	a.y *= invLength;
	return length;
}

B2D_HD float b2dDistanceSquared(V2 a, V2 b)
{
	V2 c = a - b;
	return b2dDot(c, c);
}

// --- sinf / cosf -----------------------------------------------------------------------------
// b2Rot::Set (b2Math.h:294-299) calls libm sinf/cosf. The oracle runs on glibc 2.35 x86-64, whose
// sinf/cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h; the FMA ifunc variant) evaluate
// a double-precision polynomial after a double reduction. This is that algorithm restated with the
// one fused operation the FMA build has in the reduction; it was compared against libm for all 2^32
// float inputs on the build machine: 0 mismatches for both functions (tests/test_oracle_sincos.py
// re-checks a dense sample).
struct B2dSinCosTab
{
	double c0, c1, c2, c3, c4, s1, s2, s3;
};

B2D_HD double b2dFma(double a, double b, double c) { return __builtin_fma(a, b, c); }

B2D_HD uint32_t b2dAsUint(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	return u;
}

B2D_HD float b2dAsFloat(uint32_t u)
{
	float f;
	memcpy(&f, &u, 4);
	return f;
}

B2D_HD uint32_t b2dAbsTop12(float x) { return (b2dAsUint(x) >> 20) & 0x7ff; }

B2D_HD float b2dSinCosPoly(double x, double x2, int neg, int n)
{
	// neg selects the second table row (computes -sin / -cos "for free").
	const double c0 = neg ? -0x1p0 : 0x1p0;
	const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
	const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
	const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
	const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
	const double s1 = -0x1.555545995a603p-3;
	const double s2 = 0x1.1107605230bc4p-7;
	const double s3 = -0x1.994eb3774cf24p-13;
	if ((n & 1) == 0)
	{
		double x3 = x * x2;
		double sa = s2 + x2 * s3;
		double x7 = x3 * x2;
		double s = x + x3 * s1;
		return (float)(s + x7 * sa);
	}
	else
	{
		double x4 = x2 * x2;
		double cb = c3 + x2 * c4;
		double ca = c0 + x2 * c1;
		double x6 = x4 * x2;
		double c = ca + x4 * c2;
		return (float)(c + x6 * cb);
	}
}

B2D_HD double b2dReduceFast(double x, int* np)
{
	double r = x * 0x1.45F306DC9C883p+23;
	int n = ((int32_t)r + 0x800000) >> 24;
	*np = n;
	return b2dFma(-(double)n, 0x1.921FB54442D18p0, x);
}

B2D_HD uint32_t b2dInvPio4(int i)
{
	switch (i)
	{
	case 0: return 0xa2u; case 1: return 0xa2f9u; case 2: return 0xa2f983u; case 3: return 0xa2f9836eu;
	case 4: return 0xf9836e4eu; case 5: return 0x836e4e44u; case 6: return 0x6e4e4415u; case 7: return 0x4e441529u;
	case 8: return 0x441529fcu; case 9: return 0x1529fc27u; case 10: return 0x29fc2757u; case 11: return 0xfc2757d1u;
	case 12: return 0x2757d1f5u; case 13: return 0x57d1f534u; case 14: return 0xd1f534ddu; case 15: return 0xf534ddc0u;
	case 16: return 0x34ddc0dbu; case 17: return 0xddc0db62u; case 18: return 0xc0db6295u; case 19: return 0xdb629599u;
	case 20: return 0x6295993cu; case 21: return 0x95993c43u; case 22: return 0x993c4390u; default: return 0x3c439041u;
	}
}

B2D_HD double b2dReduceLarge(uint32_t xi, int* np)
{
	int base = (int)((xi >> 26) & 15);
	int shift = (int)((xi >> 23) & 7);
	uint64_t n, res0, res1, res2;
	xi = (xi & 0xffffff) | 0x800000;
	xi <<= shift;
	res0 = (uint32_t)(xi * b2dInvPio4(base));
	res1 = (uint64_t)xi * b2dInvPio4(base + 4);
	res2 = (uint64_t)xi * b2dInvPio4(base + 8);
	res0 = (res2 >> 32) | (res0 << 32);
	res0 += res1;
	n = (res0 + (1ULL << 61)) >> 62;
	res0 -= n << 62;
	double x = (double)(int64_t)res0;
	*np = (int)n;
	return x * 0x1.921FB54442D18p-62;
}

// which = 0 -> sinf(y), which = 1 -> cosf(y)
B2D_HD float b2dSinCosImpl(float y, int which)
{
	double x = (double)y;
	int n;
	if (b2dAbsTop12(y) < b2dAbsTop12(0x1.921fb6p-1f))
	{
		if (b2dAbsTop12(y) < b2dAbsTop12(0x1p-12f))
		{
			return which ? 1.0f : y;
		}
		return b2dSinCosPoly(x, x * x, 0, which);
	}
	else if (b2dAbsTop12(y) < b2dAbsTop12(120.0f))
	{
		x = b2dReduceFast(x, &n);
		double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
		return b2dSinCosPoly(x * s, x * x, (n & 2) != 0, n ^ which);
	}
	else if (b2dAbsTop12(y) < 0x7f8u)
	{
		uint32_t xi = b2dAsUint(y);
		int sign = (int)(xi >> 31);
		x = b2dReduceLarge(xi, &n);
		int m = n + sign;
		double s = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
		return b2dSinCosPoly(x * s, x * x, (m & 2) != 0, n ^ which);
	}
	return y - y;
}

B2D_HD float b2dSin(float y) { return b2dSinCosImpl(y, 0); }
B2D_HD float b2dCos(float y) { return b2dSinCosImpl(y, 1); }

// |y| >= 120, infinities and NaN: the table-driven reduction. Cold (a body would have to spin 19 turns without its angle
// ever being renormalised) and big: kept out of line so that the callers' loops stay small.
struct SinCosPair { float s, c; };
#if defined(__HIPCC__)
__host__ __device__ __noinline__
#endif
static SinCosPair b2dSinCosLarge(float y)
{
	SinCosPair r;
	if (b2dAbsTop12(y) < 0x7f8u)
	{
		int n;
		uint32_t xi = b2dAsUint(y);
		int sign = (int)(xi >> 31);
		double x = b2dReduceLarge(xi, &n);
		int m = n + sign;
		const double s = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
		const double xs = x * s, x2 = x * x;
		r.s = b2dSinCosPoly(xs, x2, (m & 2) != 0, n);
		r.c = b2dSinCosPoly(xs, x2, (m & 2) != 0, n ^ 1);
		return r;
	}
	r.s = r.c = y - y;
	return r;
}

// sinf(y) and cosf(y) together: ONE argument reduction feeds both polynomials, and ONE straight-line path serves every
// |y| < 120. glibc's sincosf has three ranges below 120 - |y| < 2^-12 (returns y and 1), |y| < pi/4 (no reduction) and the
// rest (reduce_fast) - but the third computes exactly what the first two return: for |y| < pi/4 reduce_fast finds n = 0 and
// fma(-0, pi/2, x) = x, and for |y| < 2^-12 the polynomials round to y and 1 (x^3/6 and x^2/2 are below half an ulp of the
// results). Every operation is the one b2dSin / b2dCos perform on the same operands; the pair is checked bit for bit
// against libm's sinf / cosf for ALL 2^32 inputs (tests/test_device_math_cpu.py, exhaustive variant in tools/probe_sincos_all.py).
// The straight line matters on the device: the position solver calls this four times per two-point contact, one wave per
// SIMD, and with three diverging ranges inlined at every site the loop was 25 KB of code.
B2D_HD void b2dSinCos(float y, float* sinOut, float* cosOut)
{
	if (b2dAbsTop12(y) < b2dAbsTop12(120.0f))
	{
		int n;
		const double x = b2dReduceFast((double)y, &n);
		const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
		const double xs = x * s, x2 = x * x;
		const bool neg = (n & 2) != 0;
		// both polynomials once (b2dSinCosPoly's two branches), then dealt by the parity of n
		const double x3 = xs * x2;
		const double sa = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
		const double x7 = x3 * x2;
		const double sp = xs + x3 * -0x1.555545995a603p-3;
		const float sinPoly = (float)(sp + x7 * sa);
		const double c0 = neg ? -0x1p0 : 0x1p0;
		const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
		const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
		const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
		const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
		const double x4 = x2 * x2;
		const double cb = c3 + x2 * c4;
		const double ca = c0 + x2 * c1;
		const double x6 = x4 * x2;
		const double cp = ca + x4 * c2;
		const float cosPoly = (float)(cp + x6 * cb);
		// (|y| < 2^-12 returns y and 1 outright in glibc; the polynomials agree except for the sign of sin(-0): a select, not a branch)
		const bool tiny = b2dAbsTop12(y) < b2dAbsTop12(0x1p-12f);
		*sinOut = tiny ? y : ((n & 1) ? cosPoly : sinPoly);
		*cosOut = tiny ? 1.0f : ((n & 1) ? sinPoly : cosPoly);
		return;
	}
	const SinCosPair big = b2dSinCosLarge(y);
	*sinOut = big.s;
	*cosOut = big.c;
}

// b2Rot::Set (b2Math.h:294-299). On the device this is ONE out-of-line copy: the glibc-exact sinf/cosf pair is ~3 KB of
// code, and inlined at every transform (ten sites in the time-of-impact root finder alone) it made the TOI kernels
// 60-200 KB - far beyond the 64 KB instruction cache, where a wave pays ~2 us per KB of cold code it walks through.
#if defined(__HIPCC__)
__device__ __noinline__ static Rot b2dRotOutOfLine(float angle)
{
	Rot q;
	b2dSinCos(angle, &q.s, &q.c);
	return q;
}
#endif

// The inlined form, for the one place where throughput matters more than code size (k_solve_small's position sweeps).
B2D_HD Rot b2dRotInline(float angle)
{
	Rot q;
	b2dSinCos(angle, &q.s, &q.c);
	return q;
}

B2D_HD Rot b2dRot(float angle)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return b2dRotOutOfLine(angle);
#else
	Rot q;
	b2dSinCos(angle, &q.s, &q.c);
	return q;
#endif
}

// b2Math.h:553-600
B2D_HD V2 b2dMulRV(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
B2D_HD V2 b2dMulTRV(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
B2D_HD V2 b2dMulXV(Xf T, V2 v)
{
	float x = (T.q.c * v.x - T.q.s * v.y) + T.p.x;
	float y = (T.q.s * v.x + T.q.c * v.y) + T.p.y;
	return v2(x, y);
}
B2D_HD V2 b2dMulTXV(Xf T, V2 v)
{
	float px = v.x - T.p.x;
	float py = v.y - T.p.y;
	float x = (T.q.c * px + T.q.s * py);
	float y = (-T.q.s * px + T.q.c * py);
	return v2(x, y);
}
// b2MulT(q, r) for rotations (b2Math.h:541-551): qT * r
B2D_HD Rot b2dMulTRR(Rot q, Rot r)
{
	Rot o;
	o.s = q.c * r.s - q.s * r.c;
	o.c = q.c * r.c + q.s * r.s;
	return o;
}
// b2MulT(A, B) for transforms (b2Math.h:594-600)
B2D_HD Xf b2dMulTXX(Xf A, Xf B)
{
	Xf C;
	C.q = b2dMulTRR(A.q, B.q);
	C.p = b2dMulTRV(A.q, B.p - A.p);
	return C;
}

// Body transform from (center, angle, localCenter): b2Body::SynchronizeTransform (b2Body.h:958-962)
B2D_HD Xf b2dXfFromSweep(V2 c, float a, V2 localCenter)
{
	Xf xf;
	xf.q = b2dRot(a);
	xf.p = c - b2dMulRV(xf.q, localCenter);
	return xf;
}

struct AABB
{
	V2 lo, hi;
};

// b2TestOverlap(AABB) (b2Collision.h:273-286): touching counts as overlap.
B2D_HD bool b2dAabbOverlap(AABB a, AABB b)
{
	V2 d1 = b.lo - a.hi;
	V2 d2 = a.lo - b.hi;
	if (d1.x > 0.0f || d1.y > 0.0f) return false;
	if (d2.x > 0.0f || d2.y > 0.0f) return false;
	return true;
}

// b2AABB::Contains (b2Collision.h:202-210)
B2D_HD bool b2dAabbContains(AABB a, AABB b)
{
	bool result = true;
	result = result && a.lo.x <= b.lo.x;
	result = result && a.lo.y <= b.lo.y;
	result = result && b.hi.x <= a.hi.x;
	result = result && b.hi.y <= a.hi.y;
	return result;
}

B2D_HD AABB b2dAabbCombine(AABB a, AABB b)
{
	AABB r;
	r.lo = b2dMinV(a.lo, b.lo);
	r.hi = b2dMaxV(a.hi, b.hi);
	return r;
}

#endif
