// Headless runner of the REFERENCE's own Testbed scene headers (included from /root/reference where they lie, unmodified)
// on top of a Box2D API chosen by the include path: the drop-in one (box2d-mt_amd/host, -> libtestbed_amd.so) or the
// reference's (-> libtestbed_ref.so). TEST INFRASTRUCTURE; the built library goes to oracle/_ref/ (git-ignored, travels).
// Follows TestMT.cpp:4-48: construct the scene, step it `steps` times with default Settings, ask TestPassed().
#include "headless_test.h"

DebugDraw g_debugDraw;
Camera g_camera;

#include "Testbed/Tests/AddPair.h"
#include "Testbed/Tests/ApplyForce.h"
#include "Testbed/Tests/BasicSliderCrank.h"
#include "Testbed/Tests/BodyTypes.h"
#include "Testbed/Tests/Breakable.h"
#include "Testbed/Tests/Bridge.h"
#include "Testbed/Tests/BulletTest.h"
#include "Testbed/Tests/Cantilever.h"
#include "Testbed/Tests/Car.h"
#include "Testbed/Tests/ContinuousTest.h"
#include "Testbed/Tests/Chain.h"
#include "Testbed/Tests/CharacterCollision.h"
#include "Testbed/Tests/CollisionFiltering.h"
#include "Testbed/Tests/CollisionProcessing.h"
#include "Testbed/Tests/CompoundShapes.h"
#include "Testbed/Tests/Confined.h"
#include "Testbed/Tests/ConvexHull.h"
#include "Testbed/Tests/ConveyorBelt.h"
#include "Testbed/Tests/DistanceTest.h"
#include "Testbed/Tests/Dominos.h"
#include "Testbed/Tests/DumpShell.h"
#include "Testbed/Tests/DuplicateProxyTest.h"
#include "Testbed/Tests/DynamicTreeTest.h"
#include "Testbed/Tests/EdgeShapes.h"
#include "Testbed/Tests/EdgeTest.h"
#include "Testbed/Tests/Gears.h"
#include "Testbed/Tests/HeavyOnLight.h"
#include "Testbed/Tests/HeavyOnLightTwo.h"
#include "Testbed/Tests/Mobile.h"
#include "Testbed/Tests/MobileBalanced.h"
#include "Testbed/Tests/MotorJoint.h"
#include "Testbed/Tests/ManyBodies.h"
#include "Testbed/Tests/MultithreadDemo.h"
#include "Testbed/Tests/OneSidedPlatform.h"
#include "Testbed/Tests/Pinball.h"
#include "Testbed/Tests/PolyCollision.h"
#include "Testbed/Tests/PolyShapes.h"
#include "Testbed/Tests/Prismatic.h"
#include "Testbed/Tests/Pulleys.h"
#include "Testbed/Tests/Pyramid.h"
#include "Testbed/Tests/RayCast.h"
#include "Testbed/Tests/Revolute.h"
#include "Testbed/Tests/RopeJoint.h"
#include "Testbed/Tests/SensorTest.h"
#include "Testbed/Tests/ShapeCast.h"
#include "Testbed/Tests/ShapeEditing.h"
#include "Testbed/Tests/SleepCollidePerf.h"
#include "Testbed/Tests/SleepCollideTest.h"
#include "Testbed/Tests/SliderCrank.h"
#include "Testbed/Tests/SphereStack.h"
#include "Testbed/Tests/TheoJansen.h"
#include "Testbed/Tests/Tiles.h"
#include "Testbed/Tests/TimeOfImpact.h"
#include "Testbed/Tests/TunnelingTest.h"
#include "Testbed/Tests/Tumbler.h"
#include "Testbed/Tests/VaryingFriction.h"
#include "Testbed/Tests/VaryingRestitution.h"
#include "Testbed/Tests/VerticalStack.h"
#include "Testbed/Tests/Web.h"

struct Entry { const char* name; TestCreateFcn* create; };
// every scene of the reference's Testbed (Testbed/Tests/TestEntries.cpp) but Rope (Box2D/Rope is out of the hot path's scope)
static const Entry kEntries[] = {
	{ "MultithreadDemo", MultithreadDemo::Create },
	{ "ManyBodies1", ManyBodies1::Create },
	{ "ManyBodies2", ManyBodies2::Create },
	{ "ManyBodies3", ManyBodies3::Create },
	{ "ManyBodies4", ManyBodies4::Create },
	{ "ManyBodies5", ManyBodies5::Create },
	{ "ManyBodies6", ManyBodies6::Create },
	{ "SleepCollidePerf", SleepCollidePerf::Create },
	{ "SleepCollideTest", SleepCollideTest::Create },
	{ "TunnelingTest", TunnelingTest::Create },
	{ "QueryTest", QueryTest::Create },
	{ "ShapeCast", ShapeCast::Create },
	{ "TimeOfImpact", TimeOfImpact::Create },
	{ "CharacterCollision", CharacterCollision::Create },
	{ "Tiles", Tiles::Create },
	{ "HeavyOnLight", HeavyOnLight::Create },
	{ "HeavyOnLightTwo", HeavyOnLightTwo::Create },
	{ "VerticalStack", VerticalStack::Create },
	{ "BasicSliderCrank", BasicSliderCrank::Create },
	{ "SliderCrank", SliderCrank::Create },
	{ "SphereStack", SphereStack::Create },
	{ "ConvexHull", ConvexHull::Create },
	{ "Tumbler", Tumbler::Create },
	{ "RayCast", RayCast::Create },
	{ "DumpShell", DumpShell::Create },
	{ "ApplyForce", ApplyForce::Create },
	{ "ContinuousTest", ContinuousTest::Create },
	{ "MotorJoint", MotorJoint::Create },
	{ "OneSidedPlatform", OneSidedPlatform::Create },
	{ "Mobile", Mobile::Create },
	{ "MobileBalanced", MobileBalanced::Create },
	{ "ConveyorBelt", ConveyorBelt::Create },
	{ "Gears", Gears::Create },
	{ "VaryingRestitution", VaryingRestitution::Create },
	{ "Cantilever", Cantilever::Create },
	{ "EdgeTest", EdgeTest::Create },
	{ "BodyTypes", BodyTypes::Create },
	{ "ShapeEditing", ShapeEditing::Create },
	{ "Car", Car::Create },
	{ "Prismatic", Prismatic::Create },
	{ "Revolute", Revolute::Create },
	{ "Pulleys", Pulleys::Create },
	{ "PolyShapes", PolyShapes::Create },
	{ "Web", Web::Create },
	{ "RopeJoint", RopeJoint::Create },
	{ "Pinball", Pinball::Create },
	{ "BulletTest", BulletTest::Create },
	{ "Confined", Confined::Create },
	{ "Pyramid", Pyramid::Create },
	{ "TheoJansen", TheoJansen::Create },
	{ "EdgeShapes", EdgeShapes::Create },
	{ "PolyCollision", PolyCollision::Create },
	{ "Bridge", Bridge::Create },
	{ "Breakable", Breakable::Create },
	{ "Chain", Chain::Create },
	{ "CollisionFiltering", CollisionFiltering::Create },
	{ "CollisionProcessing", CollisionProcessing::Create },
	{ "CompoundShapes", CompoundShapes::Create },
	{ "DistanceTest", DistanceTest::Create },
	{ "Dominos", Dominos::Create },
	{ "DynamicTreeTest", DynamicTreeTest::Create },
	{ "SensorTest", SensorTest::Create },
	{ "VaryingFriction", VaryingFriction::Create },
	{ "AddPair", AddPair::Create },
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

// The FULL state after every step as one number: FNV-1a over the bit patterns of position, angle, velocities of every body in
// world order, with its awake flag and type - what two backends must agree on bit for bit (the six summary figures are sums).
int testbed_trace_hash(const char* name, int steps, unsigned long long* hashes)
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
			unsigned long long h = 1469598103934665603ull;
			for (b2Body* b = t->GetWorld()->GetBodyList(); b; b = b->GetNext())
			{
				const float v[6] = { b->GetPosition().x, b->GetPosition().y, b->GetAngle(), b->GetLinearVelocity().x, b->GetLinearVelocity().y, b->GetAngularVelocity() };
				unsigned int w[8];
				memcpy(w, v, sizeof(v));
				w[6] = b->IsAwake() ? 1u : 0u;
				w[7] = (unsigned int)b->GetType();
				for (int k = 0; k < 8; ++k) for (int q = 0; q < 4; ++q) h = (h ^ ((w[k] >> (8 * q)) & 0xffu)) * 1099511628211ull;
			}
			hashes[i] = h;
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

// (for drivers that want every scene: the registry's size and names)
extern "C" int testbed_entry_count() { return (int)(sizeof(kEntries) / sizeof(kEntries[0])); }
extern "C" const char* testbed_entry_name(int i) { return i >= 0 && i < testbed_entry_count() ? kEntries[i].name : ""; }
