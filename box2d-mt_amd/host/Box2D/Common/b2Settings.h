// Drop-in Box2D-MT public header (MI355X build): scalar types and tuning constants.
// Mirrors the names and values of the reference's Box2D/Common/b2Settings.h:34-196 so that user
// code written against the reference compiles unchanged; the values are also what the HIP kernels
// use (box2d-mt_amd/csrc/b2d_math.h).
#ifndef B2_SETTINGS_H
#define B2_SETTINGS_H

#include <stddef.h>
#include <assert.h>
#include <float.h>

#define B2_NOT_USED(x) ((void)(x))
#define b2Assert(A) assert(A)

typedef signed char int8;
typedef signed short int16;
typedef signed int int32;
typedef unsigned char uint8;
typedef unsigned short uint16;
typedef unsigned int uint32;
typedef float float32;
typedef double float64;

#define b2_maxFloat FLT_MAX
#define b2_epsilon FLT_EPSILON
#define b2_pi 3.14159265359f

// collision
#define b2_maxManifoldPoints 2
#define b2_maxPolygonVertices 8
#define b2_aabbExtension 0.1f
#define b2_aabbMultiplier 2.0f
#define b2_linearSlop 0.005f
#define b2_angularSlop (2.0f / 180.0f * b2_pi)
#define b2_polygonRadius (2.0f * b2_linearSlop)
#define b2_maxSubSteps 8

// dynamics
#define b2_maxTOIContacts 32
#define b2_velocityThreshold 1.0f
#define b2_maxLinearCorrection 0.2f
#define b2_maxAngularCorrection (8.0f / 180.0f * b2_pi)
#define b2_maxTranslation 2.0f
#define b2_maxTranslationSquared (b2_maxTranslation * b2_maxTranslation)
#define b2_maxRotation (0.5f * b2_pi)
#define b2_maxRotationSquared (b2_maxRotation * b2_maxRotation)
#define b2_baumgarte 0.2f
#define b2_toiBaugarte 0.75f

// sleep
#define b2_timeToSleep 0.5f
#define b2_linearSleepTolerance 0.01f
#define b2_angularSleepTolerance (2.0f / 180.0f * b2_pi)

// multithreading (plugin surface limits)
#define b2_cacheLineSize 64
#define b2_maxThreads 8
#define b2_maxRangeSubTasks 8
#define b2_maxIslandsPerSolveTask 16
#define b2_maxWorldStepTaskGroups 1

void* b2Alloc(int32 size);
void b2Free(void* mem);
void b2Log(const char* string, ...);

struct b2Version
{
	int32 major;
	int32 minor;
	int32 revision;
};

extern b2Version b2_version;
extern b2Version b2_mtVersion;

#endif
