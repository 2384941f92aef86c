// Headless step-and-dump harness with a plain C ABI (TEST INFRASTRUCTURE, not product).
//
// Compiled twice from this one source (see oracle/Makefile and box2d-mt_amd/Makefile):
//   -DB2H_BACKEND_REF  against /root/reference (sources compiled where they lie) -> oracle/_ref/libb2ref_harness.so
//   -DB2H_BACKEND_AMD  against this repo's drop-in Box2D headers + libb2hip.so   -> box2d-mt_amd/libb2amd_harness.so
// The harness pattern follows the reference's own batch runner (Testbed/Framework/TestMT.cpp:50-132):
// build a scene, step it with a b2ThreadPoolTaskExecutor, read back every body in creation order.
#include "scenes.h"

#include <stdio.h>
#include <string.h>
#include <map>
#include <algorithm>

#if !defined(B2H_BACKEND_REF) && !defined(B2H_BACKEND_AMD)
#error "define B2H_BACKEND_REF or B2H_BACKEND_AMD"
#endif

struct b2h_world;

// Records the listener callbacks (the deferred, deterministic ones) since the last b2h_get_events* call: 10 ints per event:
//   kind 0 BeginContact / 1 EndContact : bodyA, fixtureA, bodyB, fixtureB, 0...
//   kind 2 PreSolve  : ..., old point count, new point count, bits(old normalImpulse[0]), bits(new localNormal.x), enabled after the call
//   kind 3 PostSolve : ..., count, bits(normalImpulses[0]), bits(normalImpulses[1]), bits(tangentImpulses[0]), bits(tangentImpulses[1])
// mode bits: 1 begin / end, 2 PreSolve, 4 PostSolve, 8 PreSolve disables the contacts picked by a fixed rule of the body
// indices (b2Contact::SetEnabled(false): a one-way-platform style use of the callback), 16 PreSolve edits the contact's
// material by a fixed rule of the body indices: every contact of body 0 (the ground) becomes a conveyor belt
// (SetTangentSpeed, as Testbed/Tests/ConveyorBelt.h does), some contacts get another friction / restitution and some are reset.
class b2hEventRecorder : public b2ContactListener
{
public:
	b2hEventRecorder(b2h_world* owner, int mode) : m_owner(owner), m_mode(mode) {}
	void BeginContact(b2Contact* contact) override { Record(0, contact, 0, 0, 0, 0, 0); }
	void EndContact(b2Contact* contact) override { Record(1, contact, 0, 0, 0, 0, 0); }
	void PreSolve(b2Contact* contact, const b2Manifold* oldManifold) override;
	void PostSolve(b2Contact* contact, const b2ContactImpulse* impulse) override;
	// (the *Immediate forms run on the executor's worker threads, each with its own threadId: counted per thread)
	bool BeginContactImmediate(b2Contact*, uint32 t) override { Hit(t); return (m_mode & 1) != 0; }
	bool EndContactImmediate(b2Contact*, uint32 t) override { Hit(t); return (m_mode & 1) != 0; }
	bool PreSolveImmediate(b2Contact*, const b2Manifold*, uint32 t) override { Hit(t); return (m_mode & (2 | 8 | 16)) != 0; }
	bool PostSolveImmediate(b2Contact*, const b2ContactImpulse*, uint32 t) override { Hit(t); return (m_mode & 4) != 0; }
	void Hit(uint32 threadId) { if (threadId < b2_maxThreads) __atomic_fetch_add(&immediateCalls[threadId], 1, __ATOMIC_RELAXED); else __atomic_fetch_add(&badThreadIds, 1, __ATOMIC_RELAXED); }
	int immediateCalls[b2_maxThreads] = {};
	int badThreadIds = 0;
	void Record(int kind, b2Contact* contact, int a, int b, int c, int d, int e);
	std::vector<int> log; // 10 ints per event
	int m_mode;
private:
	b2h_world* m_owner;
};

// A user contact filter: the default rule AND a fixed rule of the body indices (b2ContactFilter::ShouldCollide override).
class b2hFilter : public b2ContactFilter
{
public:
	explicit b2hFilter(b2h_world* owner) : m_owner(owner) {}
	bool ShouldCollide(b2Fixture* fixtureA, b2Fixture* fixtureB, uint32 threadId) override;
private:
	b2h_world* m_owner;
};

struct b2h_world
{
	b2hEventRecorder* recorder;
	b2hFilter* filter;
	b2World* world;
	b2ThreadPoolTaskExecutor* executor;
	b2h::Scene scene;
	std::map<const b2Body*, int> bodyIndex;
	float profileSum[13];
	int profileSteps;
};

enum
{
	B2H_FLAG_CONTINUOUS = 1,
	B2H_FLAG_ALLOW_SLEEP = 2,
	B2H_FLAG_WARM_START = 4,
	B2H_FLAG_SUBSTEPPING = 8
};

extern "C"
{

const char* b2h_backend()
{
#if defined(B2H_BACKEND_NAME)
	return B2H_BACKEND_NAME;
#elif defined(B2H_BACKEND_REF)
	return "reference";
#else
	return "amd";
#endif
}

b2h_world* b2h_create(int scene, int p0, int p1, float f0, float f1, unsigned seed, int flags, int threads)
{
	b2h_world* h = new b2h_world;
	b2ThreadPoolOptions opt;
	opt.totalThreadCount = threads;
	h->executor = new b2ThreadPoolTaskExecutor(opt);
	h->world = new b2World(b2Vec2(0.0f, -10.0f));
	h->world->SetContinuousPhysics((flags & B2H_FLAG_CONTINUOUS) != 0);
	h->world->SetAllowSleeping((flags & B2H_FLAG_ALLOW_SLEEP) != 0);
	h->world->SetWarmStarting((flags & B2H_FLAG_WARM_START) != 0);
	h->world->SetSubStepping((flags & B2H_FLAG_SUBSTEPPING) != 0);
	b2h::SceneParams p;
	p.scene = scene;
	p.p0 = p0;
	p.p1 = p1;
	p.f0 = f0;
	p.f1 = f1;
	p.seed = seed;
	b2h::BuildScene(h->scene, h->world, p);
	for (size_t i = 0; i < h->scene.bodies.size(); ++i)
	{
		h->bodyIndex[h->scene.bodies[i]] = (int)i;
	}
	memset(h->profileSum, 0, sizeof(h->profileSum));
	h->profileSteps = 0;
	h->recorder = nullptr;
	h->filter = nullptr;
	return h;
}

void b2h_destroy(b2h_world* h)
{
	if (h == NULL) return;
	delete h->world;
	delete h->recorder;
	delete h->filter;
	delete h->executor;
	delete h;
}

void b2h_default_iters(b2h_world* h, int* velIters, int* posIters)
{
	*velIters = h->scene.velIters;
	*posIters = h->scene.posIters;
}

void b2h_step(b2h_world* h, int steps, float dt, int velIters, int posIters)
{
	for (int i = 0; i < steps; ++i)
	{
		if (h->scene.sliderBounces && h->scene.joint != NULL)
		{
			// between steps, through the public joint API: turn the motor round once the slider has reached the limit it runs at
			b2PrismaticJoint* slider = static_cast<b2PrismaticJoint*>(h->scene.joint);
			const float speed = slider->GetJointSpeed(), at = slider->GetJointTranslation();
			const bool atUpper = speed > 0.0f && at >= slider->GetUpperLimit() - b2_epsilon;
			const bool atLower = speed < 0.0f && at <= slider->GetLowerLimit() + b2_epsilon;
			if (atUpper || atLower) slider->SetMotorSpeed(-slider->GetMotorSpeed());
		}
		if (h->scene.drag != NULL && h->scene.servoStep == 170)
		{
			// mouse up (Test::MouseUp, Testbed/Framework/Test.cpp:203-215): the crate is let go in mid-air
			h->world->DestroyJoint(h->scene.drag);
			h->scene.drag = NULL;
		}
		if (h->scene.drag != NULL)
		{
			const float t = 0.04f * (float)h->scene.servoStep;
			static_cast<b2MouseJoint*>(h->scene.drag)->SetTarget(b2Vec2(10.5f + 4.0f * sinf(t), 5.0f + 2.0f * sinf(2.0f * t + 1.0f)));
		}
		if (h->scene.props) b2h::PropsEdits(h->scene, h->world);
		if (h->scene.lifecycle)
		{
			const size_t before = h->scene.bodies.size();
			b2h::LifecycleEdits(h->scene, h->world);
			for (size_t k = before; k < h->scene.bodies.size(); ++k) h->bodyIndex[h->scene.bodies[k]] = (int)k;
		}
		if (!h->scene.servos.empty())
		{
			// motor-joint targets move on a figure of eight (the way the Testbed's MotorJoint scene drives its own: new offsets each step)
			const float t = 0.05f * (float)h->scene.servoStep++;
			for (size_t k = 0; k < h->scene.servos.size(); ++k)
			{
				b2MotorJoint* servo = static_cast<b2MotorJoint*>(h->scene.servos[k]);
				b2Vec2 offset(40.0f + 6.0f * (float)k + 3.0f * sinf(2.0f * t), 8.0f + 2.0f * sinf(t));
				servo->SetLinearOffset(offset);
				servo->SetAngularOffset(0.3f * (float)(k + 1) * sinf(0.5f * t));
			}
		}
		h->world->Step(dt, velIters, posIters, *h->executor);
		const b2Profile& p = h->world->GetProfile();
		const float v[13] = { p.step, p.collide, p.solve, p.solveTraversal, p.solveInit, p.solveVelocity,
			p.solvePosition, p.solveTOI, p.solveTOIFindMinContact, p.broadphase, p.broadphaseSyncFixtures,
			p.broadphaseFindContacts, p.locking };
		for (int k = 0; k < 13; ++k) h->profileSum[k] += v[k];
		h->profileSteps += 1;
	}
}

// The C-ABI world behind the drop-in b2World (NULL for the reference backend): lets bench.py reach the
// measurement hooks of include/b2hip.h without going around the public API for stepping.
void* b2h_device_world(b2h_world* h)
{
#if defined(B2H_BACKEND_AMD)
	return (void*)h->world->GetDeviceWorld();
#else
	(void)h;
	return NULL;
#endif
}

int b2h_body_count(b2h_world* h)
{
	return (int)h->scene.bodies.size();
}

// 8 floats per body, creation order: x, y, angle, vx, vy, w, awake(0/1), type
void b2h_get_bodies(b2h_world* h, float* out)
{
	for (size_t i = 0; i < h->scene.bodies.size(); ++i)
	{
		const b2Body* b = h->scene.bodies[i];
		float* o = out + 8 * i;
		if (b == NULL)
		{
			// destroyed (life-cycle scene): the row stays, marked by type -1
			for (int k = 0; k < 7; ++k) o[k] = 0.0f;
			o[7] = -1.0f;
			continue;
		}
		o[0] = b->GetPosition().x;
		o[1] = b->GetPosition().y;
		o[2] = b->GetAngle();
		o[3] = b->GetLinearVelocity().x;
		o[4] = b->GetLinearVelocity().y;
		o[5] = b->GetAngularVelocity();
		o[6] = b->IsAwake() ? 1.0f : 0.0f;
		o[7] = (float)b->GetType();
	}
}

// 6 floats per body: mass, inertia (about center), local center x,y, world center x,y
void b2h_get_mass(b2h_world* h, float* out)
{
	for (size_t i = 0; i < h->scene.bodies.size(); ++i)
	{
		const b2Body* b = h->scene.bodies[i];
		float* o = out + 6 * i;
		if (b == NULL)
		{
			for (int k = 0; k < 6; ++k) o[k] = 0.0f;
			continue;
		}
		o[0] = b->GetMass();
		o[1] = b->GetInertia();
		o[2] = b->GetLocalCenter().x;
		o[3] = b->GetLocalCenter().y;
		o[4] = b->GetWorldCenter().x;
		o[5] = b->GetWorldCenter().y;
	}
}

int b2h_contact_count(b2h_world* h)
{
	return h->world->GetContactCount();
}

static int FixtureIndexInBody(const b2Fixture* f)
{
	// The fixture list is newest-first (b2Body.cpp:210-211); index = creation order within the body.
	const b2Body* b = f->GetBody();
	int total = 0, pos = -1;
	for (const b2Fixture* g = b->GetFixtureList(); g; g = g->GetNext())
	{
		if (g == f) pos = total;
		++total;
	}
	return total - 1 - pos;
}

} // extern "C" (reopened below): the recorder needs FixtureIndexInBody

static int FloatBits(float f)
{
	int i;
	memcpy(&i, &f, 4);
	return i;
}

void b2hEventRecorder::Record(int kind, b2Contact* contact, int a, int b, int c, int d, int e)
{
	const b2Fixture* fA = contact->GetFixtureA();
	const b2Fixture* fB = contact->GetFixtureB();
	const int row[10] = { kind, m_owner->bodyIndex[fA->GetBody()], FixtureIndexInBody(fA), m_owner->bodyIndex[fB->GetBody()],
		FixtureIndexInBody(fB), a, b, c, d, e };
	log.insert(log.end(), row, row + 10);
}

void b2hEventRecorder::PreSolve(b2Contact* contact, const b2Manifold* oldManifold)
{
	const int bA = m_owner->bodyIndex[contact->GetFixtureA()->GetBody()], bB = m_owner->bodyIndex[contact->GetFixtureB()->GetBody()];
	if ((m_mode & 8) != 0 && (bA + 3 * bB) % 7 == 0) contact->SetEnabled(false);
	if ((m_mode & 16) != 0)
	{
		const int lo = bA < bB ? bA : bB, hi = bA < bB ? bB : bA;
		if (lo == 0) contact->SetTangentSpeed((hi & 1) ? 1.5f : -0.75f);             // the ground carries everything sideways
		else if ((lo + hi) % 5 == 0) contact->SetFriction(0.05f + 0.01f * (float)(hi % 7));
		else if ((lo + hi) % 5 == 1) contact->SetRestitution(0.6f);
		else if ((lo + hi) % 5 == 2 && contact->GetFriction() != 0.0f) contact->SetFriction(0.5f * contact->GetFriction()); // (persists: halves every step)
		else if ((lo + 2 * hi) % 9 == 0) { contact->ResetFriction(); contact->ResetRestitution(); }
	}
	if ((m_mode & 2) != 0)
	{
		const b2Manifold* m = contact->GetManifold();
		Record(2, contact, oldManifold->pointCount, m->pointCount, oldManifold->pointCount > 0 ? FloatBits(oldManifold->points[0].normalImpulse) : 0,
			FloatBits(m->localNormal.x), contact->IsEnabled() ? 1 : 0);
	}
}

void b2hEventRecorder::PostSolve(b2Contact* contact, const b2ContactImpulse* impulse)
{
	Record(3, contact, impulse->count, FloatBits(impulse->normalImpulses[0]), impulse->count > 1 ? FloatBits(impulse->normalImpulses[1]) : 0,
		FloatBits(impulse->tangentImpulses[0]), impulse->count > 1 ? FloatBits(impulse->tangentImpulses[1]) : 0);
}

bool b2hFilter::ShouldCollide(b2Fixture* fixtureA, b2Fixture* fixtureB, uint32 threadId)
{
	if (!b2ContactFilter::ShouldCollide(fixtureA, fixtureB, threadId)) return false;
	const int bA = m_owner->bodyIndex[fixtureA->GetBody()], bB = m_owner->bodyIndex[fixtureB->GetBody()];
	const int lo = bA < bB ? bA : bB, hi = bA < bB ? bB : bA;
	return lo == 0 || (31 * lo + 17 * hi) % 11 != 0; // (everything still collides with the ground)
}

extern "C"
{

// Listener: install (mode != 0, see b2hEventRecorder) or remove the recording listener / fetch what it has logged since the
// last call. b2h_get_events: begin / end only, 5 ints per event (kind, bodyA, fixtureA, bodyB, fixtureB);
// b2h_get_events_ex: every recorded callback, 10 ints per event. Both return the number of events and clear the log.
void b2h_record_events(b2h_world* h, int mode)
{
	if (h->recorder)
	{
		h->world->SetContactListener(nullptr);
		delete h->recorder;
		h->recorder = nullptr;
	}
	if (mode)
	{
		h->recorder = new b2hEventRecorder(h, mode);
		h->world->SetContactListener(h->recorder);
	}
}

// How often each thread id called an *Immediate callback since the listener was installed: out[b2_maxThreads], out[b2_maxThreads] =
// calls with an id outside [0, b2_maxThreads). Returns b2_maxThreads.
int b2h_immediate_calls_by_thread(b2h_world* h, int* out)
{
	for (int t = 0; t <= (int)b2_maxThreads; ++t) out[t] = 0;
	if (h->recorder)
	{
		for (int t = 0; t < (int)b2_maxThreads; ++t) out[t] = h->recorder->immediateCalls[t];
		out[b2_maxThreads] = h->recorder->badThreadIds;
	}
	return (int)b2_maxThreads;
}

int b2h_get_events(b2h_world* h, int cap, int* out)
{
	if (!h->recorder) return 0;
	const int n = (int)h->recorder->log.size() / 10;
	int m = 0;
	for (int i = 0; i < n; ++i)
	{
		const int* row = h->recorder->log.data() + 10 * i;
		if (row[0] > 1) continue;
		if (m < cap) memcpy(out + 5 * m, row, 5 * sizeof(int));
		++m;
	}
	h->recorder->log.clear();
	return m;
}

int b2h_get_events_ex(b2h_world* h, int cap, int* out)
{
	if (!h->recorder) return 0;
	const int n = (int)h->recorder->log.size() / 10;
	for (int i = 0; i < n && i < cap; ++i) memcpy(out + 10 * i, h->recorder->log.data() + 10 * i, 10 * sizeof(int));
	h->recorder->log.clear();
	return n;
}

// User contact filter (b2World::SetContactFilter): the harness's rule of the body indices on top of the default rule.
void b2h_set_filter(b2h_world* h, int enable)
{
	if (enable && !h->filter)
	{
		h->filter = new b2hFilter(h);
		h->world->SetContactFilter(h->filter);
	}
	else if (!enable && h->filter)
	{
		h->world->SetContactFilter(nullptr);
		delete h->filter;
		h->filter = nullptr;
	}
}

} // extern "C": the query callbacks below are C++ classes

namespace
{
struct AllFixtures : b2QueryCallback
{
	std::vector<const b2Fixture*> hits;
	bool ReportFixture(b2Fixture* fixture) override { hits.push_back(fixture); return true; }
};
struct ClosestHit : b2RayCastCallback
{
	const b2Fixture* fixture = nullptr;
	b2Vec2 point, normal;
	float32 fraction = 1.0f;
	float32 ReportFixture(b2Fixture* f, const b2Vec2& p, const b2Vec2& n, float32 fr) override
	{
		fixture = f; point = p; normal = n; fraction = fr;
		return fr; // clip the ray to this hit: the last report is the closest one, whatever the visiting order
	}
};
}

extern "C"
{

// b2World::QueryAABB: (body, fixture index in body) of every reported fixture, sorted; returns the count.
int b2h_query_aabb(b2h_world* h, float lx, float ly, float ux, float uy, int cap, int* out)
{
	AllFixtures cb;
	b2AABB aabb;
	aabb.lowerBound.Set(lx, ly);
	aabb.upperBound.Set(ux, uy);
	h->world->QueryAABB(&cb, aabb);
	std::vector<std::pair<int, int> > ids;
	for (size_t i = 0; i < cb.hits.size(); ++i) ids.push_back(std::make_pair(h->bodyIndex[cb.hits[i]->GetBody()], FixtureIndexInBody(cb.hits[i])));
	std::sort(ids.begin(), ids.end());
	for (size_t i = 0; i < ids.size() && (int)i < cap; ++i)
	{
		out[2 * i] = ids[i].first;
		out[2 * i + 1] = ids[i].second;
	}
	return (int)ids.size();
}

// b2World::RayCast with a closest-hit callback: out7 = body, fixture index, point.xy, normal.xy, fraction; returns 1 on a hit.
int b2h_raycast_closest(b2h_world* h, float x1, float y1, float x2, float y2, float* out7)
{
	ClosestHit cb;
	h->world->RayCast(&cb, b2Vec2(x1, y1), b2Vec2(x2, y2));
	if (!cb.fixture) return 0;
	out7[0] = (float)h->bodyIndex[cb.fixture->GetBody()];
	out7[1] = (float)FixtureIndexInBody(cb.fixture);
	out7[2] = cb.point.x; out7[3] = cb.point.y;
	out7[4] = cb.normal.x; out7[5] = cb.normal.y;
	out7[6] = cb.fraction;
	return 1;
}

// Dumps every contact of the world's contact list.
//   ids:      4 ints  per contact: bodyA, fixtureA (index within body), bodyB, fixtureB
//   flags:    1 int   per contact: bit0 touching, bit1 enabled
//   manifold: 16 floats per contact: type, pointCount, localNormal.xy, localPoint.xy,
//             then per point (2x): localPoint.xy, normalImpulse, tangentImpulse, id.key (bit pattern as float)
// Returns the number of contacts written (<= cap). Order is the list order of the backend and is
// NOT comparable between backends: sort by ids on the caller's side.
// b2World::DrawDebugData through a b2Draw that only counts: per primitive kind the number of calls, and a checksum of every
// coordinate / radius / colour it was handed (order-independent: a sum of bit patterns), so that two builds can be compared.
namespace
{
struct CountingDraw : public b2Draw
{
	long long calls[7];
	unsigned long long sum;
	CountingDraw() : sum(0) { for (int k = 0; k < 7; ++k) calls[k] = 0; }
	void Add(float v) { unsigned int u; memcpy(&u, &v, 4); sum += (unsigned long long)u * 2654435761ull + 1ull; }
	void Add(const b2Vec2& v) { Add(v.x); Add(v.y); }
	void Add(const b2Color& c) { Add(c.r); Add(c.g); Add(c.b); }
	void DrawPolygon(const b2Vec2* v, int32 n, const b2Color& c) override { ++calls[0]; for (int32 k = 0; k < n; ++k) Add(v[k]); Add(c); }
	void DrawSolidPolygon(const b2Vec2* v, int32 n, const b2Color& c) override { ++calls[1]; for (int32 k = 0; k < n; ++k) Add(v[k]); Add(c); }
	void DrawCircle(const b2Vec2& p, float32 r, const b2Color& c) override { ++calls[2]; Add(p); Add(r); Add(c); }
	void DrawSolidCircle(const b2Vec2& p, float32 r, const b2Vec2& axis, const b2Color& c) override { ++calls[3]; Add(p); Add(r); Add(axis); Add(c); }
	void DrawSegment(const b2Vec2& a, const b2Vec2& b, const b2Color& c) override { ++calls[4]; Add(a); Add(b); Add(c); }
	void DrawTransform(const b2Transform& xf) override { ++calls[5]; Add(xf.p); Add(xf.q.s); Add(xf.q.c); }
	void DrawPoint(const b2Vec2& p, float32 size, const b2Color& c) override { ++calls[6]; Add(p); Add(size); Add(c); }
};
}

// out[0..6] calls per primitive (polygon, solid polygon, circle, solid circle, segment, transform, point), out[7] checksum
void b2h_debug_draw(b2h_world* h, unsigned int flags, long long* out)
{
	CountingDraw draw;
	draw.SetFlags(flags);
	h->world->SetDebugDraw(&draw);
	h->world->DrawDebugData();
	h->world->SetDebugDraw(nullptr);
	for (int k = 0; k < 7; ++k) out[k] = draw.calls[k];
	out[7] = (long long)draw.sum;
}

// The mixed material of every contact, in b2h_get_contacts' order: friction, restitution, tangent speed (b2Contact.h:40-50,157).
int b2h_get_contact_materials(b2h_world* h, int cap, float* out)
{
	int n = 0;
	for (const b2Contact* c = h->world->GetContactList(); c && n < cap; c = c->GetNext(), ++n)
	{
		out[3 * n + 0] = c->GetFriction();
		out[3 * n + 1] = c->GetRestitution();
		out[3 * n + 2] = c->GetTangentSpeed();
	}
	return n;
}

int b2h_get_contacts(b2h_world* h, int cap, int* ids, int* flags, float* manifold)
{
	int n = 0;
	for (const b2Contact* c = h->world->GetContactList(); c && n < cap; c = c->GetNext())
	{
		const b2Fixture* fA = c->GetFixtureA();
		const b2Fixture* fB = c->GetFixtureB();
		ids[4 * n + 0] = h->bodyIndex[fA->GetBody()];
		ids[4 * n + 1] = FixtureIndexInBody(fA) | (c->GetChildIndexA() << 16); // (child index of a chain shape in the high half)
		ids[4 * n + 2] = h->bodyIndex[fB->GetBody()];
		ids[4 * n + 3] = FixtureIndexInBody(fB) | (c->GetChildIndexB() << 16);
		flags[n] = (c->IsTouching() ? 1 : 0) | (c->IsEnabled() ? 2 : 0);
		const b2Manifold* m = c->GetManifold();
		float* o = manifold + 16 * n;
		memset(o, 0, 16 * sizeof(float));
		o[1] = (float)m->pointCount;
		if (m->pointCount > 0)
		{
			// type / normal / points are only defined while the manifold has points (the reference never
			// initialises m_manifold.type of a contact that has not touched yet, b2Contact.cpp:125-159)
			o[0] = (float)m->type;
			o[2] = m->localNormal.x;
			o[3] = m->localNormal.y;
			o[4] = m->localPoint.x;
			o[5] = m->localPoint.y;
			for (int k = 0; k < m->pointCount; ++k)
			{
				float* q = o + 6 + 5 * k;
				q[0] = m->points[k].localPoint.x;
				q[1] = m->points[k].localPoint.y;
				q[2] = m->points[k].normalImpulse;
				q[3] = m->points[k].tangentImpulse;
				uint32 key = m->points[k].id.key;
				memcpy(q + 4, &key, 4);
			}
		}
		++n;
	}
	return n;
}

// Mean of the 13 b2Profile fields (b2TimeStep.h:25-40, in declaration order) over the steps taken so far.
// Returns the number of steps averaged.
int b2h_get_profile(b2h_world* h, float* out)
{
	float inv = h->profileSteps > 0 ? 1.0f / (float)h->profileSteps : 0.0f;
	for (int k = 0; k < 13; ++k) out[k] = h->profileSum[k] * inv;
	return h->profileSteps;
}

void b2h_reset_profile(b2h_world* h)
{
	memset(h->profileSum, 0, sizeof(h->profileSum));
	h->profileSteps = 0;
}

static void FillPolygon(b2PolygonShape& poly, int count, const float* verts, int asBox)
{
	if (asBox)
	{
		poly.SetAsBox(verts[0], verts[1]);
	}
	else
	{
		b2Vec2 v[b2_maxPolygonVertices];
		for (int i = 0; i < count; ++i) v[i].Set(verts[2 * i], verts[2 * i + 1]);
		poly.Set(v, count);
	}
}

// (both backends: the drop-in API offers b2CollidePolygons & co. as well, on the CPU build of the device's manifold code)
// --- per-function probes (shape-level narrow phase and trig), used to pin the C restatement and
// --- the device math bit-for-bit. Polygons are passed as count + 8 vertices (hull is rebuilt by Set()).

static void DumpManifold(const b2Manifold& m, float* o)
{
	memset(o, 0, 16 * sizeof(float));
	o[0] = (float)m.type;
	o[1] = (float)m.pointCount;
	if (m.pointCount == 0) return;
	o[2] = m.localNormal.x;
	o[3] = m.localNormal.y;
	o[4] = m.localPoint.x;
	o[5] = m.localPoint.y;
	for (int k = 0; k < m.pointCount; ++k)
	{
		float* q = o + 6 + 5 * k;
		q[0] = m.points[k].localPoint.x;
		q[1] = m.points[k].localPoint.y;
		uint32 key = m.points[k].id.key;
		memcpy(q + 4, &key, 4);
	}
}

static b2Transform MakeXf(const float* xf)
{
	b2Transform t;
	t.p.Set(xf[0], xf[1]);
	t.q.Set(xf[2]);
	return t;
}

// xf = {px, py, angle}
void b2h_probe_collide_polygons(int countA, const float* vertsA, int boxA, const float* xfA,
	int countB, const float* vertsB, int boxB, const float* xfB, float* manifold16)
{
	b2PolygonShape a, b;
	FillPolygon(a, countA, vertsA, boxA);
	FillPolygon(b, countB, vertsB, boxB);
	b2Manifold m;
	memset(&m, 0, sizeof(m));
	b2CollidePolygons(&m, &a, MakeXf(xfA), &b, MakeXf(xfB));
	DumpManifold(m, manifold16);
}

void b2h_probe_collide_polygon_circle(int countA, const float* vertsA, int boxA, const float* xfA,
	const float* circleB /* px, py, r */, const float* xfB, float* manifold16)
{
	b2PolygonShape a;
	FillPolygon(a, countA, vertsA, boxA);
	b2CircleShape c;
	c.m_p.Set(circleB[0], circleB[1]);
	c.m_radius = circleB[2];
	b2Manifold m;
	memset(&m, 0, sizeof(m));
	b2CollidePolygonAndCircle(&m, &a, MakeXf(xfA), &c, MakeXf(xfB));
	DumpManifold(m, manifold16);
}

void b2h_probe_collide_circles(const float* circleA, const float* xfA, const float* circleB, const float* xfB,
	float* manifold16)
{
	b2CircleShape a, b;
	a.m_p.Set(circleA[0], circleA[1]);
	a.m_radius = circleA[2];
	b.m_p.Set(circleB[0], circleB[1]);
	b.m_radius = circleB[2];
	b2Manifold m;
	memset(&m, 0, sizeof(m));
	b2CollideCircles(&m, &a, MakeXf(xfA), &b, MakeXf(xfB));
	DumpManifold(m, manifold16);
}

// edge = {v1x, v1y, v2x, v2y, hasV0, v0x, v0y, hasV3, v3x, v3y}
static void FillEdge(b2EdgeShape& e, const float* edge)
{
	e.Set(b2Vec2(edge[0], edge[1]), b2Vec2(edge[2], edge[3]));
	if (edge[4] != 0.0f)
	{
		e.m_hasVertex0 = true;
		e.m_vertex0.Set(edge[5], edge[6]);
	}
	if (edge[7] != 0.0f)
	{
		e.m_hasVertex3 = true;
		e.m_vertex3.Set(edge[8], edge[9]);
	}
}

void b2h_probe_collide_edge_polygon(const float* edgeA, const float* xfA,
	int countB, const float* vertsB, int boxB, const float* xfB, float* manifold16)
{
	b2EdgeShape e;
	FillEdge(e, edgeA);
	b2PolygonShape b;
	FillPolygon(b, countB, vertsB, boxB);
	b2Manifold m;
	memset(&m, 0, sizeof(m));
	b2CollideEdgeAndPolygon(&m, &e, MakeXf(xfA), &b, MakeXf(xfB));
	DumpManifold(m, manifold16);
}

void b2h_probe_collide_edge_circle(const float* edgeA, const float* xfA, const float* circleB, const float* xfB,
	float* manifold16)
{
	b2EdgeShape e;
	FillEdge(e, edgeA);
	b2CircleShape c;
	c.m_p.Set(circleB[0], circleB[1]);
	c.m_radius = circleB[2];
	b2Manifold m;
	memset(&m, 0, sizeof(m));
	b2CollideEdgeAndCircle(&m, &e, MakeXf(xfA), &c, MakeXf(xfB));
	DumpManifold(m, manifold16);
}


// Polygon build probe: returns count, then vertices[8], normals[8], centroid, and mass data for `density`.
// out = 1 + 16 + 16 + 2 + 4 floats = 39
void b2h_probe_polygon(int count, const float* verts, float density, float* out39)
{
	b2PolygonShape p;
	FillPolygon(p, count, verts, 0);
	memset(out39, 0, 39 * sizeof(float));
	out39[0] = (float)p.m_count;
	for (int i = 0; i < p.m_count; ++i)
	{
		out39[1 + 2 * i] = p.m_vertices[i].x;
		out39[2 + 2 * i] = p.m_vertices[i].y;
		out39[17 + 2 * i] = p.m_normals[i].x;
		out39[18 + 2 * i] = p.m_normals[i].y;
	}
	out39[33] = p.m_centroid.x;
	out39[34] = p.m_centroid.y;
	b2MassData md;
	p.ComputeMass(&md, density);
	out39[35] = md.mass;
	out39[36] = md.center.x;
	out39[37] = md.center.y;
	out39[38] = md.I;
}

// sin/cos exactly as b2Rot::Set evaluates them (b2Math.h:294-299).
void b2h_probe_sincos(int n, const float* angles, float* sinOut, float* cosOut)
{
	for (int i = 0; i < n; ++i)
	{
		b2Rot q(angles[i]);
		sinOut[i] = q.s;
		cosOut[i] = q.c;
	}
}

// --- continuous-collision probes (both backends: the drop-in API offers b2Distance / b2TimeOfImpact / b2ShapeCast as well)
// ---: b2Distance and b2TimeOfImpact on raw vertex proxies.
// sweep9 = {localCenter.x, .y, c0.x, c0.y, c.x, c.y, a0, a, alpha0}
static void FillProxy(b2DistanceProxy& p, b2Vec2* store, int count, const float* verts, float radius)
{
	for (int i = 0; i < count; ++i) store[i].Set(verts[2 * i], verts[2 * i + 1]);
	p.Set(store, count, radius);
}

static b2Sweep MakeSweep(const float* s9)
{
	b2Sweep s;
	s.localCenter.Set(s9[0], s9[1]);
	s.c0.Set(s9[2], s9[3]);
	s.c.Set(s9[4], s9[5]);
	s.a0 = s9[6];
	s.a = s9[7];
	s.alpha0 = s9[8];
	return s;
}

// out6 = {pointA.x, pointA.y, pointB.x, pointB.y, distance, iterations}
void b2h_probe_distance(int countA, const float* vertsA, float radiusA, const float* xfA,
	int countB, const float* vertsB, float radiusB, const float* xfB, int useRadii, float* out6)
{
	b2Vec2 storeA[b2_maxPolygonVertices], storeB[b2_maxPolygonVertices];
	b2DistanceInput in;
	FillProxy(in.proxyA, storeA, countA, vertsA, radiusA);
	FillProxy(in.proxyB, storeB, countB, vertsB, radiusB);
	in.transformA = MakeXf(xfA);
	in.transformB = MakeXf(xfB);
	in.useRadii = useRadii != 0;
	b2SimplexCache cache;
	cache.count = 0;
	b2DistanceOutput out;
	b2Distance(&out, &cache, &in);
	out6[0] = out.pointA.x; out6[1] = out.pointA.y;
	out6[2] = out.pointB.x; out6[3] = out.pointB.y;
	out6[4] = out.distance;
	out6[5] = (float)out.iterations;
}

// out2 = {state (b2TOIOutput::State), t}
void b2h_probe_toi(int countA, const float* vertsA, float radiusA, const float* sweepA9,
	int countB, const float* vertsB, float radiusB, const float* sweepB9, float tMax, float* out2)
{
	b2Vec2 storeA[b2_maxPolygonVertices], storeB[b2_maxPolygonVertices];
	b2TOIInput in;
	FillProxy(in.proxyA, storeA, countA, vertsA, radiusA);
	FillProxy(in.proxyB, storeB, countB, vertsB, radiusB);
	in.sweepA = MakeSweep(sweepA9);
	in.sweepB = MakeSweep(sweepB9);
	in.tMax = tMax;
	b2TOIOutput out;
	b2TimeOfImpact(&out, &in);
	out2[0] = (float)out.state;
	out2[1] = out.t;
}
// out6 = {hit, point.x, point.y, normal.x, normal.y, lambda}; iterations in out6[6]
void b2h_probe_shape_cast(int countA, const float* vertsA, float radiusA, const float* xfA,
	int countB, const float* vertsB, float radiusB, const float* xfB, float tx, float ty, float* out7)
{
	b2Vec2 storeA[b2_maxPolygonVertices], storeB[b2_maxPolygonVertices];
	b2ShapeCastInput in;
	FillProxy(in.proxyA, storeA, countA, vertsA, radiusA);
	FillProxy(in.proxyB, storeB, countB, vertsB, radiusB);
	in.transformA = MakeXf(xfA);
	in.transformB = MakeXf(xfB);
	in.translationB.Set(tx, ty);
	b2ShapeCastOutput out;
	const bool hit = b2ShapeCast(&out, &in);
	out7[0] = hit ? 1.0f : 0.0f;
	out7[1] = out.point.x; out7[2] = out.point.y;
	out7[3] = out.normal.x; out7[4] = out.normal.y;
	out7[5] = out.lambda;
	out7[6] = (float)out.iterations;
}

// b2AABB::RayCast: out3 = {fraction, normal.x, normal.y}; returns 1 on a hit
int b2h_probe_aabb_raycast(const float* box4, const float* ray5, float* out3)
{
	b2AABB box;
	box.lowerBound.Set(box4[0], box4[1]);
	box.upperBound.Set(box4[2], box4[3]);
	b2RayCastInput in;
	in.p1.Set(ray5[0], ray5[1]);
	in.p2.Set(ray5[2], ray5[3]);
	in.maxFraction = ray5[4];
	b2RayCastOutput out;
	if (!box.RayCast(&out, in)) return 0;
	out3[0] = out.fraction; out3[1] = out.normal.x; out3[2] = out.normal.y;
	return 1;
}

// b2DynamicTree through its public interface: `ops` random create / move / destroy operations on `count` boxes, after each of
// which an AABB query and a ray cast are compared with brute force over the stored fat boxes. Returns the number of
// disagreements (0 expected); out3 = final height, max balance, area ratio.
namespace
{
struct TreeProbe
{
	bool QueryCallback(int32 id) { hits.push_back(id); return true; }
	float32 RayCastCallback(const b2RayCastInput& in, int32 id) { rayHits.push_back(id); return in.maxFraction; }
	std::vector<int32> hits, rayHits;
};
}
int b2h_probe_dynamic_tree(unsigned seed, int count, int ops, float* out3)
{
	b2h::Pcg32 rng(seed);
	b2DynamicTree tree;
	std::vector<int32> ids(count, b2_nullNode);
	std::vector<b2AABB> tight(count);
	int bad = 0;
	for (int op = 0; op < ops; ++op)
	{
		const int k = (int)(rng.Unit() * (float)count) % count;
		const float roll = rng.Unit();
		if (ids[k] == b2_nullNode)
		{
			const b2Vec2 c(rng.Range(-40.0f, 40.0f), rng.Range(-40.0f, 40.0f));
			const b2Vec2 h(rng.Range(0.1f, 3.0f), rng.Range(0.1f, 3.0f));
			tight[k].lowerBound = c - h;
			tight[k].upperBound = c + h;
			ids[k] = tree.CreateProxy(tight[k], (void*)(intptr_t)k);
		}
		else if (roll < 0.2f)
		{
			tree.DestroyProxy(ids[k]);
			ids[k] = b2_nullNode;
		}
		else
		{
			const b2Vec2 d(rng.Range(-2.0f, 2.0f), rng.Range(-2.0f, 2.0f));
			tight[k].lowerBound += d;
			tight[k].upperBound += d;
			tree.MoveProxy(ids[k], tight[k], d);
		}
		if (op % 8 != 0) continue;
		tree.Validate();
		b2AABB q;
		const b2Vec2 c(rng.Range(-40.0f, 40.0f), rng.Range(-40.0f, 40.0f));
		q.lowerBound = c - b2Vec2(5.0f, 4.0f);
		q.upperBound = c + b2Vec2(5.0f, 4.0f);
		TreeProbe probe;
		tree.Query(&probe, q);
		b2RayCastInput ray;
		ray.p1.Set(rng.Range(-45.0f, 45.0f), rng.Range(-45.0f, 45.0f));
		ray.p2.Set(rng.Range(-45.0f, 45.0f), rng.Range(-45.0f, 45.0f));
		ray.maxFraction = 1.0f;
		tree.RayCast(&probe, ray);
		std::sort(probe.hits.begin(), probe.hits.end());
		std::sort(probe.rayHits.begin(), probe.rayHits.end());
		std::vector<int32> want, wantRay;
		for (int i = 0; i < count; ++i)
		{
			if (ids[i] == b2_nullNode) continue;
			const b2AABB& fat = tree.GetFatAABB(ids[i]);
			if (!fat.Contains(tight[i])) ++bad; // the stored box always holds the tight one
			if ((intptr_t)tree.GetUserData(ids[i]) != i) ++bad;
			if (b2TestOverlap(fat, q)) want.push_back(ids[i]);
			b2RayCastOutput o;
			// (a ray that starts inside a box is not a hit of b2AABB::RayCast but the tree visits the leaf: test the segment's box)
			b2AABB seg;
			seg.lowerBound = b2Min(ray.p1, ray.p2);
			seg.upperBound = b2Max(ray.p1, ray.p2);
			b2RayCastInput back = ray;
			back.p1 = ray.p2;
			back.p2 = ray.p1;
			const bool inside = fat.lowerBound.x <= ray.p1.x && ray.p1.x <= fat.upperBound.x && fat.lowerBound.y <= ray.p1.y && ray.p1.y <= fat.upperBound.y;
			if (b2TestOverlap(fat, seg) && (inside || fat.RayCast(&o, ray) || fat.RayCast(&o, back))) wantRay.push_back(ids[i]);
		}
		std::sort(want.begin(), want.end());
		std::sort(wantRay.begin(), wantRay.end());
		if (want != probe.hits) ++bad;
		// every leaf the exact test names must have been visited (the tree's own test is conservative: it may visit more)
		for (size_t i = 0; i < wantRay.size(); ++i)
			if (!std::binary_search(probe.rayHits.begin(), probe.rayHits.end(), wantRay[i])) ++bad;
	}
	out3[0] = (float)tree.GetHeight();
	out3[1] = (float)tree.GetMaxBalance();
	out3[2] = tree.GetAreaRatio();
	return bad;
}

// b2World::GetTreeHeight / GetTreeBalance / GetTreeQuality / GetProxyCount (b2World.h:199-206)
void b2h_tree_stats(b2h_world* h, float* out4)
{
	out4[0] = (float)h->world->GetTreeHeight();
	out4[1] = (float)h->world->GetTreeBalance();
	out4[2] = h->world->GetTreeQuality();
	out4[3] = (float)h->world->GetProxyCount();
}

// b2Joint::GetReactionForce / GetReactionTorque(inv_dt) of every joint, in creation order: rows of 3 floats; returns the joint count
int b2h_joint_reactions(b2h_world* h, float inv_dt, int cap, float* out)
{
	std::vector<b2Joint*> joints;
	for (b2Joint* j = h->world->GetJointList(); j; j = j->GetNext()) joints.push_back(j);
	std::reverse(joints.begin(), joints.end()); // (the list is newest first)
	for (size_t i = 0; i < joints.size() && (int)i < cap; ++i)
	{
		const b2Vec2 f = joints[i]->GetReactionForce(inv_dt);
		out[3 * i] = f.x;
		out[3 * i + 1] = f.y;
		out[3 * i + 2] = joints[i]->GetReactionTorque(inv_dt);
	}
	return (int)joints.size();
}

// b2Joint::GetAnchorA / GetAnchorB of every joint, in creation order: rows of 4 floats; returns the joint count
int b2h_joint_anchors(b2h_world* h, int cap, float* out)
{
	std::vector<b2Joint*> joints;
	for (b2Joint* j = h->world->GetJointList(); j; j = j->GetNext()) joints.push_back(j);
	std::reverse(joints.begin(), joints.end());
	for (size_t i = 0; i < joints.size() && (int)i < cap; ++i)
	{
		const b2Vec2 a = joints[i]->GetAnchorA(), b = joints[i]->GetAnchorB();
		out[4 * i] = a.x;
		out[4 * i + 1] = a.y;
		out[4 * i + 2] = b.x;
		out[4 * i + 3] = b.y;
	}
	return (int)joints.size();
}

// b2Body::GetJointList of every body: per body the number of joints and, in list order, the index of the body on the other
// side of each (up to 4 per body; -1 pads): rows of 5 ints; returns the body count
int b2h_body_joint_lists(b2h_world* h, int cap, int* out)
{
	const int n = (int)h->scene.bodies.size();
	for (int i = 0; i < n && i < cap; ++i)
	{
		int* row = out + 5 * i;
		row[0] = 0;
		for (int k = 1; k < 5; ++k) row[k] = -1;
		b2Body* b = h->scene.bodies[(size_t)i];
		if (b == NULL) continue;
		for (b2JointEdge* e = b->GetJointList(); e; e = e->next)
		{
			if (row[0] < 4)
			{
				std::map<const b2Body*, int>::const_iterator it = h->bodyIndex.find(e->other);
				row[1 + row[0]] = it != h->bodyIndex.end() ? it->second : -2;
			}
			row[0] += 1;
		}
	}
	return n;
}

// The scalar setters of the joint classes, applied to every joint of the world by type (b2DistanceJoint::SetLength /
// SetFrequency / SetDampingRatio, b2FrictionJoint::SetMaxForce / SetMaxTorque, b2GearJoint::SetRatio, b2MotorJoint::SetMaxForce /
// SetMaxTorque / SetCorrectionFactor, b2MouseJoint::SetMaxForce / SetFrequency / SetDampingRatio, b2RopeJoint::SetMaxLength,
// b2WeldJoint::SetFrequency / SetDampingRatio); returns the number of joints retuned
int b2h_retune_joints(b2h_world* h, int round)
{
	int n = 0;
	const float k = 1.0f + 0.05f * (float)round;
	for (b2Joint* j = h->world->GetJointList(); j; j = j->GetNext())
	{
		++n;
		switch (j->GetType())
		{
		case e_distanceJoint:
		{
			b2DistanceJoint* d = static_cast<b2DistanceJoint*>(j);
			d->SetLength(d->GetLength() * k);
			d->SetFrequency(d->GetFrequency() + 1.0f);
			d->SetDampingRatio(b2Min(1.0f, d->GetDampingRatio() + 0.1f));
			break;
		}
		case e_frictionJoint:
		{
			b2FrictionJoint* f = static_cast<b2FrictionJoint*>(j);
			f->SetMaxForce(0.5f * f->GetMaxForce());
			f->SetMaxTorque(0.5f * f->GetMaxTorque());
			break;
		}
		case e_gearJoint: static_cast<b2GearJoint*>(j)->SetRatio(1.1f * static_cast<b2GearJoint*>(j)->GetRatio()); break;
		case e_motorJoint:
		{
			b2MotorJoint* m = static_cast<b2MotorJoint*>(j);
			m->SetMaxForce(0.7f * m->GetMaxForce());
			m->SetMaxTorque(0.7f * m->GetMaxTorque());
			m->SetCorrectionFactor(0.5f * m->GetCorrectionFactor());
			break;
		}
		case e_mouseJoint:
		{
			b2MouseJoint* m = static_cast<b2MouseJoint*>(j);
			m->SetMaxForce(0.8f * m->GetMaxForce());
			m->SetFrequency(m->GetFrequency() + 0.5f);
			m->SetDampingRatio(0.9f * m->GetDampingRatio());
			break;
		}
		case e_ropeJoint: static_cast<b2RopeJoint*>(j)->SetMaxLength(0.9f * static_cast<b2RopeJoint*>(j)->GetMaxLength()); break;
		case e_weldJoint:
		{
			b2WeldJoint* wj = static_cast<b2WeldJoint*>(j);
			wj->SetFrequency(wj->GetFrequency() + 2.0f);
			wj->SetDampingRatio(0.5f);
			break;
		}
		default: --n; break;
		}
	}
	return n;
}

// b2WheelJoint::GetJointTranslation / GetJointLinearSpeed / GetJointAngle / GetJointAngularSpeed of every wheel joint, in
// creation order: rows of 4 floats; returns the number of wheel joints
int b2h_wheel_states(b2h_world* h, int cap, float* out)
{
	std::vector<b2Joint*> joints;
	for (b2Joint* j = h->world->GetJointList(); j; j = j->GetNext()) if (j->GetType() == e_wheelJoint) joints.push_back(j);
	std::reverse(joints.begin(), joints.end());
	for (size_t i = 0; i < joints.size() && (int)i < cap; ++i)
	{
		const b2WheelJoint* wj = static_cast<const b2WheelJoint*>(joints[i]);
		out[4 * i] = wj->GetJointTranslation();
		out[4 * i + 1] = wj->GetJointLinearSpeed();
		out[4 * i + 2] = wj->GetJointAngle();
		out[4 * i + 3] = wj->GetJointAngularSpeed();
	}
	return (int)joints.size();
}

// b2RopeJoint::GetLimitState of every rope joint, in creation order; returns their number
int b2h_rope_states(b2h_world* h, int cap, int* out)
{
	std::vector<b2Joint*> joints;
	for (b2Joint* j = h->world->GetJointList(); j; j = j->GetNext()) if (j->GetType() == e_ropeJoint) joints.push_back(j);
	std::reverse(joints.begin(), joints.end());
	for (size_t i = 0; i < joints.size() && (int)i < cap; ++i) out[i] = (int)static_cast<const b2RopeJoint*>(joints[i])->GetLimitState();
	return (int)joints.size();
}

} // extern "C"
