// Drop-in header: profile block (reference: Box2D/Dynamics/b2TimeStep.h:25-40). Times in ms.
// In the MI355X build the fields are filled from HIP events around the device phases.
#ifndef B2_TIME_STEP_H
#define B2_TIME_STEP_H

#include "Box2D/Common/b2Math.h"

struct b2Profile
{
	float32 step;
	float32 collide;
	float32 solve;
	float32 solveTraversal;
	float32 solveInit;
	float32 solveVelocity;
	float32 solvePosition;
	float32 solveTOI;
	float32 solveTOIFindMinContact;
	float32 broadphase;
	float32 broadphaseSyncFixtures;
	float32 broadphaseFindContacts;
	float32 locking;
};

struct b2TimeStep
{
	float32 dt;
	float32 inv_dt;
	float32 dtRatio;
	int32 velocityIterations;
	int32 positionIterations;
	bool warmStarting;
};

struct b2Position
{
	b2Vec2 c;
	float32 a;
};

struct b2Velocity
{
	b2Vec2 v;
	float32 w;
};

#endif
