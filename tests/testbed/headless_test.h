// Headless stand-in for Testbed/Framework/Test.h (TEST INFRASTRUCTURE): just enough of the Test base class, Settings, the
// debug-draw / camera globals and the GLFW key names for the reference's scene headers (Testbed/Tests/*.h) to compile
// UNMODIFIED against a Box2D API - the reference's own or the drop-in one of box2d-mt_amd/host. Nothing is drawn.
// Mirrors the members and virtuals the scenes use (Test.h:176-282, Test.cpp:35-140, 275-330 for what Step does).
#ifndef HEADLESS_TEST_H
#define HEADLESS_TEST_H

#include "Box2D/Box2D.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

class Test;
typedef Test* TestCreateFcn();

#define RAND_LIMIT 32767
#define DRAW_STRING_NEW_LINE 16

// GLFW key names the scenes switch on (values as in glfw3.h)
enum
{
	GLFW_KEY_A = 65, GLFW_KEY_B = 66, GLFW_KEY_C = 67, GLFW_KEY_D = 68, GLFW_KEY_E = 69, GLFW_KEY_F = 70, GLFW_KEY_G = 71,
	GLFW_KEY_H = 72, GLFW_KEY_I = 73, GLFW_KEY_J = 74, GLFW_KEY_K = 75, GLFW_KEY_L = 76, GLFW_KEY_M = 77, GLFW_KEY_N = 78,
	GLFW_KEY_O = 79, GLFW_KEY_P = 80, GLFW_KEY_Q = 81, GLFW_KEY_R = 82, GLFW_KEY_S = 83, GLFW_KEY_T = 84, GLFW_KEY_U = 85,
	GLFW_KEY_V = 86, GLFW_KEY_W = 87, GLFW_KEY_X = 88, GLFW_KEY_Y = 89, GLFW_KEY_Z = 90, GLFW_KEY_COMMA = 44, GLFW_KEY_PERIOD = 46,
	GLFW_KEY_0 = 48, GLFW_KEY_1 = 49, GLFW_KEY_2 = 50, GLFW_KEY_3 = 51, GLFW_KEY_4 = 52, GLFW_KEY_5 = 53, GLFW_KEY_6 = 54, GLFW_KEY_7 = 55,
	GLFW_KEY_8 = 56, GLFW_KEY_9 = 57, GLFW_KEY_SPACE = 32, GLFW_KEY_MINUS = 45, GLFW_KEY_EQUAL = 61, GLFW_KEY_LEFT_BRACKET = 91, GLFW_KEY_RIGHT_BRACKET = 93
};

inline float32 RandomFloat()
{
	float32 r = (float32)(rand() & (RAND_LIMIT));
	r /= RAND_LIMIT;
	r = 2.0f * r - 1.0f;
	return r;
}

inline float32 RandomFloat(float32 lo, float32 hi)
{
	float32 r = (float32)(rand() & (RAND_LIMIT));
	r /= RAND_LIMIT;
	r = (hi - lo) * r + lo;
	return r;
}

struct Settings
{
	Settings() : hz(60.0f), velocityIterations(8), positionIterations(3), threadCount(1), drawContactPoints(false),
		enableWarmStarting(true), enableContinuous(true), enableSubStepping(false), enableSleep(true), pause(false), singleStep(false) {}
	float32 hz;
	int32 velocityIterations, positionIterations, threadCount;
	bool drawContactPoints, enableWarmStarting, enableContinuous, enableSubStepping, enableSleep, pause, singleStep;
};

struct Camera
{
	Camera() : m_center(0.0f, 20.0f), m_zoom(1.0f) {}
	b2Vec2 m_center;
	float32 m_zoom;
};

// DebugDraw: every call is accepted and dropped
struct DebugDraw : public b2Draw
{
	void DrawPolygon(const b2Vec2*, int32, const b2Color&) override {}
	void DrawSolidPolygon(const b2Vec2*, int32, const b2Color&) override {}
	void DrawCircle(const b2Vec2&, float32, const b2Color&) override {}
	void DrawSolidCircle(const b2Vec2&, float32, const b2Vec2&, const b2Color&) override {}
	void DrawSegment(const b2Vec2&, const b2Vec2&, const b2Color&) override {}
	void DrawTransform(const b2Transform&) override {}
	void DrawPoint(const b2Vec2&, float32, const b2Color&) override {}
	void DrawString(int, int, const char*, ...) {}
	void DrawString(const b2Vec2&, const char*, ...) {}
	void DrawAABB(b2AABB*, const b2Color&) {}
	void Flush() {}
};

extern DebugDraw g_debugDraw;
extern Camera g_camera;

enum class TestResult { NONE = 0, PASS, FAIL };

// what Test::PreSolveImmediate keeps of every manifold point of a step, per calling thread (Test.h:141-153, 263-264): scenes
// read it after the step (CollisionProcessing.h: which bodies touched)
const int32 k_maxContactPoints = 8192;
struct ContactPoint
{
	b2Fixture* fixtureA;
	b2Fixture* fixtureB;
	b2Vec2 normal;
	b2Vec2 position;
	b2PointState state;
	float32 normalImpulse;
	float32 tangentImpulse;
	float32 separation;
};

class Test : public b2ContactListener
{
public:
	Test() : m_threadPoolExec(b2ThreadPoolOptions())
	{
		m_world = new b2World(b2Vec2(0.0f, -10.0f));
		m_world->SetContactListener(this);
		m_textLine = 30;
		m_stepCount = 0;
		m_timeStep = 0.0f;
		m_mouseJoint = nullptr;
		m_bomb = nullptr;
		m_visible = false;
		for (uint32 t = 0; t < b2_maxThreads; ++t) m_pointCount[t] = 0;
		b2BodyDef bodyDef;
		m_groundBody = m_world->CreateBody(&bodyDef);
	}
	virtual ~Test() { delete m_world; }

	virtual void Step(Settings* settings)
	{
		const float32 timeStep = settings->hz > 0.0f ? 1.0f / settings->hz : 0.0f;
		m_timeStep = timeStep;
		for (uint32 t = 0; t < b2_maxThreads; ++t) m_pointCount[t] = 0; // (Test::Step, Test.cpp:292-295)
		m_world->SetAllowSleeping(settings->enableSleep);
		m_world->SetWarmStarting(settings->enableWarmStarting);
		m_world->SetContinuousPhysics(settings->enableContinuous);
		m_world->SetSubStepping(settings->enableSubStepping);
		m_world->Step(timeStep, settings->velocityIterations, settings->positionIterations, m_threadPoolExec);
		if (timeStep > 0.0f) ++m_stepCount;
	}
	virtual void Keyboard(int key) { B2_NOT_USED(key); }
	virtual void KeyboardUp(int key) { B2_NOT_USED(key); }
	virtual void MouseDown(const b2Vec2& p) { B2_NOT_USED(p); }
	virtual void MouseUp(const b2Vec2& p) { B2_NOT_USED(p); }
	virtual void JointDestroyed(b2Joint* joint) { B2_NOT_USED(joint); }
	void DrawTitle(const char*) {}
	void LaunchBomb() {}

	bool BeginContactImmediate(b2Contact*, uint32) override { return false; }
	bool EndContactImmediate(b2Contact*, uint32) override { return false; }
	// Test.cpp:73-110: every point of every updated manifold goes into m_points of the calling thread
	bool PreSolveImmediate(b2Contact* contact, const b2Manifold* oldManifold, uint32 threadId) override
	{
		const b2Manifold* now = contact->GetManifold();
		if (now->pointCount == 0) return false;
		b2PointState before[b2_maxManifoldPoints], after[b2_maxManifoldPoints];
		b2GetPointStates(before, after, oldManifold, now);
		b2WorldManifold wm;
		contact->GetWorldManifold(&wm);
		for (int32 i = 0; i < now->pointCount && m_pointCount[threadId] < k_maxContactPoints; ++i)
		{
			ContactPoint& cp = m_points[threadId][m_pointCount[threadId]++];
			cp.fixtureA = contact->GetFixtureA();
			cp.fixtureB = contact->GetFixtureB();
			cp.position = wm.points[i];
			cp.normal = wm.normal;
			cp.state = after[i];
			cp.normalImpulse = now->points[i].normalImpulse;
			cp.tangentImpulse = now->points[i].tangentImpulse;
			cp.separation = wm.separations[i];
		}
		return false;
	}
	bool PostSolveImmediate(b2Contact*, const b2ContactImpulse*, uint32) override { return false; }
	void BeginContact(b2Contact*) override {}
	void EndContact(b2Contact*) override {}
	void PreSolve(b2Contact*, const b2Manifold*) override {}
	void PostSolve(b2Contact*, const b2ContactImpulse*) override {}
	virtual TestResult TestPassed() const { return TestResult::NONE; }

	b2ThreadPoolTaskExecutor* GetExecutor() { return &m_threadPoolExec; }
	b2World* GetWorld() { return m_world; }

protected:
	b2Body* m_groundBody;
	int32 m_textLine;
	b2World* m_world;
	b2Body* m_bomb;
	b2MouseJoint* m_mouseJoint;
	bool m_visible;
	b2Vec2 m_mouseWorld;
	int32 m_stepCount;
	float32 m_timeStep;
	b2ThreadPoolTaskExecutor m_threadPoolExec;
	ContactPoint m_points[b2_maxThreads][k_maxContactPoints];
	int32 m_pointCount[b2_maxThreads];
};

#endif
