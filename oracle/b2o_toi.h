/* b2o_toi.h - CPU oracle: GJK distance + time of impact (TEST INFRASTRUCTURE, see b2o.h) */
#ifndef B2O_TOI_H
#define B2O_TOI_H

#include "b2o_internal.h"

/* b2Sweep  Box2D/Common/b2Math.h:362-382 */
typedef struct
{
	vec2 localCenter, c0, c;
	float a0, a, alpha0;
} sweep_t;

/* b2DistanceProxy  b2Distance.h:30-57 */
typedef struct
{
	const float* verts;
	int count;
	float radius;
} gjk_proxy;

/* b2SimplexCache  b2Distance.h:61-67 */
typedef struct
{
	float metric;
	int count;
	uint8_t indexA[3], indexB[3];
} gjk_cache;

typedef struct
{
	vec2 pointA, pointB;
	float distance;
	int iterations;
} gjk_output;

enum { TOI_UNKNOWN = 0, TOI_FAILED = 1, TOI_OVERLAPPED = 2, TOI_TOUCHING = 3, TOI_SEPARATED = 4 };

typedef struct
{
	int state;
	float t;
	int iterations;
} toi_output;

xform b2o_sweep_xf(const sweep_t* s, float beta);
void b2o_sweep_advance(sweep_t* s, float alpha);
void b2o_proxy_set(gjk_proxy* p, const b2o_shape* s);
void b2o_distance(gjk_output* out, gjk_cache* cache, const gjk_proxy* pA, xform xfA, const gjk_proxy* pB, xform xfB, int useRadii);
void b2o_time_of_impact(toi_output* out, const gjk_proxy* pA, const sweep_t* sweepA, const gjk_proxy* pB, const sweep_t* sweepB,
	float tMax);

#endif
