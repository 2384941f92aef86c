// Headless runner of the REFERENCE's own Testbed scene headers (included from /root/reference where they lie, unmodified)
// on top of a Box2D API chosen by the include path: the drop-in one (box2d-mt_amd/host, -> libtestbed_amd.so) or the
// reference's (-> libtestbed_ref.so). TEST INFRASTRUCTURE; the built library goes to oracle/_ref/ (git-ignored, travels).
// Follows TestMT.cpp:4-48: construct the scene, step it `steps` times with default Settings, ask TestPassed().
#include "headless_test.h"

DebugDraw g_debugDraw;
Camera g_camera;

#include "Testbed/Tests/SleepCollideTest.h"
#include "Testbed/Tests/TunnelingTest.h"
#include "Testbed/Tests/DuplicateProxyTest.h"
#include "Testbed/Tests/ManyBodies.h"
#include "Testbed/Tests/MultithreadDemo.h"
#include "Testbed/Tests/Car.h"
#include "Testbed/Tests/Pyramid.h"
#include "Testbed/Tests/Tumbler.h"
#include "Testbed/Tests/SleepCollidePerf.h"

struct Entry { const char* name; TestCreateFcn* create; };
static const Entry kEntries[] = {
	{ "SleepCollideTest", SleepCollideTest::Create }, { "TunnelingTest", TunnelingTest::Create },
	{ "QueryTest", QueryTest::Create }, { "ManyBodies6", ManyBodies6::Create }, { "ManyBodies1", ManyBodies1::Create },
	{ "MultithreadDemo", MultithreadDemo::Create }, { "Car", Car::Create }, { "Pyramid", Pyramid::Create },
	{ "Tumbler", Tumbler::Create }, { "SleepCollidePerf", SleepCollidePerf::Create },
};

extern "C"
{

// Runs scene `name` for `steps` steps. Returns TestPassed() (0 none, 1 pass, 2 fail) or -1 for an unknown name; out6 =
// body count, contact count, sum of |x| + |y| over the bodies, max |v|, 1 if every body state is finite, awake bodies.
static void Summarise(Test* t, double* out6);

// Same, with the six summary figures after EVERY step (trace = steps x 6 doubles): where two backends part.
int testbed_trace(const char* name, int steps, double* trace)
{
	for (const Entry& e : kEntries)
	{
		if (strcmp(e.name, name) != 0) continue;
		srand(0);
		Test* t = e.create();
		Settings settings;
		for (int i = 0; i < steps; ++i)
		{
			t->Step(&settings);
			Summarise(t, trace + 6 * i);
		}
		const int res = (int)t->TestPassed();
		delete t;
		return res;
	}
	return -1;
}

// Debugging aid: the body list (world order) after `steps` steps: x, y, angle, vx, vy, w, awake, type per body. Returns the
// body count (at most cap rows are written).
int testbed_states(const char* name, int steps, double* rows8, int cap)
{
	for (const Entry& e : kEntries)
	{
		if (strcmp(e.name, name) != 0) continue;
		srand(0);
		Test* t = e.create();
		Settings settings;
		for (int i = 0; i < steps; ++i) t->Step(&settings);
		int n = 0;
		for (b2Body* b = t->GetWorld()->GetBodyList(); b; b = b->GetNext(), ++n)
		{
			if (n >= cap) continue;
			double* o = rows8 + 8 * n;
			o[0] = b->GetPosition().x; o[1] = b->GetPosition().y; o[2] = b->GetAngle();
			o[3] = b->GetLinearVelocity().x; o[4] = b->GetLinearVelocity().y; o[5] = b->GetAngularVelocity();
			o[6] = b->IsAwake() ? 1.0 : 0.0; o[7] = (double)b->GetType();
		}
		delete t;
		return n;
	}
	return -1;
}

int testbed_run(const char* name, int steps, double* out6)
{
	for (const Entry& e : kEntries)
	{
		if (strcmp(e.name, name) != 0) continue;
		srand(0);
		Test* t = e.create();
		Settings settings;
		for (int i = 0; i < steps; ++i) t->Step(&settings);
		if (out6) Summarise(t, out6);
		const int res = (int)t->TestPassed();
		delete t;
		return res;
	}
	return -1;
}

}

static void Summarise(Test* t, double* out6)
{
	{
		{
		double sum = 0.0, vmax = 0.0;
		int finite = 1, awake = 0, n = 0;
		for (b2Body* b = t->GetWorld()->GetBodyList(); b; b = b->GetNext())
		{
			const b2Vec2 p = b->GetPosition(), v = b->GetLinearVelocity();
			if (!(p.x == p.x) || !(p.y == p.y) || !(v.x == v.x) || !(v.y == v.y)) finite = 0;
			sum += (p.x < 0 ? -p.x : p.x) + (p.y < 0 ? -p.y : p.y);
			const double sp = (double)v.x * v.x + (double)v.y * v.y;
			if (sp > vmax) vmax = sp;
			awake += b->IsAwake() ? 1 : 0;
			++n;
		}
		out6[0] = n; out6[1] = t->GetWorld()->GetContactCount(); out6[2] = sum; out6[3] = vmax; out6[4] = finite; out6[5] = awake;
		}
	}
}
