/* b2o_internal.h - CPU oracle internals (TEST INFRASTRUCTURE, see b2o.h). Tiny fp32 vector helpers
 * written so that every expression rounds once per operation in the reference's operand order
 * (Box2D/Common/b2Math.h); build with -ffp-contract=off. */
#ifndef B2O_INTERNAL_H
#define B2O_INTERNAL_H

#include "b2o.h"

#include <math.h>
#include <stdint.h>

#define B2O_EPSILON 1.192092896e-07f
#define B2O_MAXFLOAT 3.402823466e+38f
#define B2O_PI 3.14159265359f
#define B2O_AABB_EXTENSION 0.1f
#define B2O_AABB_MULTIPLIER 2.0f
#define B2O_LINEAR_SLOP 0.005f
#define B2O_ANGULAR_SLOP (2.0f / 180.0f * B2O_PI)
#define B2O_VELOCITY_THRESHOLD 1.0f
#define B2O_MAX_LINEAR_CORRECTION 0.2f
#define B2O_MAX_TRANSLATION 2.0f
#define B2O_MAX_TRANSLATION_SQ (B2O_MAX_TRANSLATION * B2O_MAX_TRANSLATION)
#define B2O_MAX_ROTATION (0.5f * B2O_PI)
#define B2O_MAX_ROTATION_SQ (B2O_MAX_ROTATION * B2O_MAX_ROTATION)
#define B2O_BAUMGARTE 0.2f
#define B2O_TIME_TO_SLEEP 0.5f
#define B2O_LINEAR_SLEEP_TOL 0.01f
#define B2O_ANGULAR_SLEEP_TOL (2.0f / 180.0f * B2O_PI)

/* SHAPE_CHAIN = one child of a b2ChainShape: the edge b2ChainShape::GetChildEdge hands out (b2ChainShape.cpp:114-147) for the
 * narrow phase and the TOI proxy; its AABB has no radius (b2ChainShape.cpp:174-189) */
enum { SHAPE_CIRCLE = 0, SHAPE_EDGE = 1, SHAPE_POLYGON = 2, SHAPE_CHAIN = 3 };
#define SHAPE_IS_SEGMENT(t) ((t) == SHAPE_EDGE || (t) == SHAPE_CHAIN)
enum { MANIFOLD_CIRCLES = 0, MANIFOLD_FACE_A = 1, MANIFOLD_FACE_B = 2 };
#define CF_VERTEX 0u
#define CF_FACE 1u

typedef struct { float x, y; } vec2;
typedef struct { float s, c; } rot;
typedef struct { vec2 p; rot q; } xform;
typedef struct { vec2 v; uint32_t id; } clipv;

typedef struct
{
	vec2 localNormal, localPoint;
	vec2 p[2];
	float ni[2], ti[2];
	uint32_t id[2];
	int type, pointCount;
} manifold;

static inline vec2 v_make(float x, float y) { vec2 r; r.x = x; r.y = y; return r; }
static inline vec2 v_add(vec2 a, vec2 b) { return v_make(a.x + b.x, a.y + b.y); }
static inline vec2 v_sub(vec2 a, vec2 b) { return v_make(a.x - b.x, a.y - b.y); }
static inline vec2 v_neg(vec2 a) { return v_make(-a.x, -a.y); }
static inline vec2 v_scale(float s, vec2 a) { return v_make(s * a.x, s * a.y); }
static inline float v_dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float v_cross(vec2 a, vec2 b) { return a.x * b.y - a.y * b.x; }
static inline vec2 v_cross_vs(vec2 a, float s) { return v_make(s * a.y, -s * a.x); }
static inline vec2 v_cross_sv(float s, vec2 a) { return v_make(-s * a.y, s * a.x); }
static inline float f_min(float a, float b) { return a < b ? a : b; }
static inline float f_max(float a, float b) { return a > b ? a : b; }
static inline float f_abs(float a) { return a > 0.0f ? a : -a; }
static inline float f_clamp(float a, float lo, float hi) { return f_max(lo, f_min(a, hi)); }
static inline vec2 v_min(vec2 a, vec2 b) { return v_make(f_min(a.x, b.x), f_min(a.y, b.y)); }
static inline vec2 v_max(vec2 a, vec2 b) { return v_make(f_max(a.x, b.x), f_max(a.y, b.y)); }
static inline float v_length(vec2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
static inline float v_dist_sq(vec2 a, vec2 b) { vec2 c = v_sub(a, b); return v_dot(c, c); }
static inline float v_normalize(vec2* a)
{
	float length = v_length(*a);
	if (length < B2O_EPSILON) return 0.0f;
	float inv = 1.0f / length;
	a->x *= inv;
	a->y *= inv;
	return length;
}
/* b2Rot::Set: libm sinf / cosf, exactly what the reference calls (b2Math.h:294-299) */
static inline rot r_make(float angle) { rot q; q.s = sinf(angle); q.c = cosf(angle); return q; }
static inline vec2 r_mul(rot q, vec2 v) { return v_make(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
static inline vec2 r_mul_t(rot q, vec2 v) { return v_make(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
static inline vec2 xf_mul(xform T, vec2 v)
{
	float x = (T.q.c * v.x - T.q.s * v.y) + T.p.x;
	float y = (T.q.s * v.x + T.q.c * v.y) + T.p.y;
	return v_make(x, y);
}
static inline vec2 xf_mul_tv(xform T, vec2 v)
{
	float px = v.x - T.p.x;
	float py = v.y - T.p.y;
	return v_make(T.q.c * px + T.q.s * py, -T.q.s * px + T.q.c * py);
}
static inline xform xf_mul_t(xform A, xform B)
{
	xform C;
	C.q.s = A.q.c * B.q.s - A.q.s * B.q.c;
	C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
	C.p = r_mul_t(A.q, v_sub(B.p, A.p));
	return C;
}

static inline vec2 shape_vert(const b2o_shape* s, int i) { return v_make(s->verts[2 * i], s->verts[2 * i + 1]); }
static inline vec2 shape_normal(const b2o_shape* s, int i) { return v_make(s->normals[2 * i], s->normals[2 * i + 1]); }
static inline uint32_t make_id(uint32_t ia, uint32_t ib, uint32_t ta, uint32_t tb)
{
	return (ia & 0xffu) | ((ib & 0xffu) << 8) | (ta << 16) | (tb << 24);
}
static inline uint32_t swap_id(uint32_t k)
{
	return ((k >> 8) & 0xffu) | ((k & 0xffu) << 8) | (((k >> 24) & 0xffu) << 16) | (((k >> 16) & 0xffu) << 24);
}

void b2o_evaluate(manifold* m, const b2o_shape* sA, xform xfA, const b2o_shape* sB, xform xfB);
void b2o_collide_polygons(manifold* m, const b2o_shape* polyA, xform xfA, const b2o_shape* polyB, xform xfB);
void b2o_collide_circles(manifold* m, const b2o_shape* cA, xform xfA, const b2o_shape* cB, xform xfB);
void b2o_collide_polygon_circle(manifold* m, const b2o_shape* polyA, xform xfA, const b2o_shape* cB, xform xfB);
void b2o_collide_edge_circle(manifold* m, const b2o_shape* edgeA, xform xfA, const b2o_shape* cB, xform xfB);
void b2o_collide_edge_polygon(manifold* m, const b2o_shape* edgeA, xform xfA, const b2o_shape* polyB, xform xfB);

#endif
