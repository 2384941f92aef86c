// Drop-in header: b2TimeOfImpact with the reference's types (Box2D/Collision/b2TimeOfImpact.h:25-58), computed by the CPU
// build of the conservative-advancement routine the TOI kernels use (csrc/b2d_toi.h).
#ifndef B2_TIME_OF_IMPACT_H
#define B2_TIME_OF_IMPACT_H

#include "Box2D/Common/b2Math.h"
#include "Box2D/Collision/b2Distance.h"

struct b2TOIInput
{
	b2DistanceProxy proxyA;
	b2DistanceProxy proxyB;
	b2Sweep sweepA;
	b2Sweep sweepB;
	float32 tMax;
};

struct b2TOIOutput
{
	enum State { e_unknown, e_failed, e_overlapped, e_touching, e_separated };
	State state;
	float32 t;
};

void b2TimeOfImpact(b2TOIOutput* output, const b2TOIInput* input);

#endif
