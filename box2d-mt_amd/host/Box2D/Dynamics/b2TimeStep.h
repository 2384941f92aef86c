// Drop-in header: the profile block callers read through b2World::GetProfile() (reference interface:
// Box2D/Dynamics/b2TimeStep.h:25-40, b2AddProfile :76-91). Milliseconds. In the MI355X build the thirteen values come
// from HIP events around the device phases (b2hip_get_profile fills them in this declaration order).
#ifndef B2_TIME_STEP_H
#define B2_TIME_STEP_H

#include "Box2D/Common/b2Math.h"

struct b2Profile
{
	float32 step, collide;                                           // whole step; contact update
	float32 solve, solveTraversal, solveInit, solveVelocity, solvePosition; // island build and solver phases
	float32 solveTOI, solveTOIFindMinContact;                        // continuous collision
	float32 broadphase, broadphaseSyncFixtures, broadphaseFindContacts;
	float32 locking;                                                 // always 0 here: no locks on the device path
};

static_assert(sizeof(b2Profile) == 13 * sizeof(float32), "b2Profile is thirteen floats, in b2hip_get_profile's order");

// dest += scale * src, field by field (the Testbed averages profiles with it).
inline void b2AddProfile(b2Profile& dest, const b2Profile& src, float32 scale)
{
	float32* d = &dest.step;
	const float32* s = &src.step;
	for (int i = 0; i < 13; ++i) d[i] += scale * s[i];
}

#endif
