// Scene recipes used by the parity harness and the bench.
//
// TEST INFRASTRUCTURE. This file is written against the PUBLIC Box2D-MT API only
// (b2World / b2Body / b2Fixture / shapes / b2RevoluteJointDef / b2DistanceJointDef / b2PrismaticJointDef / b2WeldJointDef / ...), so the very same source is
// compiled twice:
//   * against the reference headers + sources under /root/reference  -> oracle/_ref/libb2ref_harness.so
//   * against this repo's drop-in headers (box2d-mt_amd/host)         -> libb2amd_harness.so
// which makes "identical inputs" true by construction and doubles as the drop-in proof.
//
// Recipes restate (not copy) the reference's Testbed scenes:
//   HelloWorld        HelloWorld/HelloWorld.cpp:27-108
//   Pyramid           Testbed/Tests/Pyramid.h:30-69
//   Tumbler           Testbed/Tests/Tumbler.h:31-87     (boxes pre-placed on a grid)
//   Field             Testbed/Tests/ManyBodies.h:203-313 (explicit PCG32 instead of glibc rand)
//   Piles / Rain      small-island and mixed-shape stress scenes of our own
#ifndef B2H_SCENES_H
#define B2H_SCENES_H

#include "Box2D/Box2D.h"
#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace b2h
{

// PCG32 (O'Neill), fixed increment: deterministic on every platform.
struct Pcg32
{
	uint64_t state;
	explicit Pcg32(uint64_t seed) : state(0)
	{
		Next();
		state += seed;
		Next();
	}
	uint32_t Next()
	{
		uint64_t old = state;
		state = old * 6364136223846793005ULL + 1442695040888963407ULL;
		uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
	}
	// 15-bit resolution like the Testbed's RandomFloat (Testbed/Framework/Test.h:45-60).
	float Unit()
	{
		float r = (float)(Next() & 32767u);
		r /= 32767.0f;
		return r;
	}
	float Range(float lo, float hi)
	{
		return (hi - lo) * Unit() + lo;
	}
	float Signed()
	{
		return 2.0f * Unit() - 1.0f;
	}
};

enum SceneId
{
	e_helloWorld = 0,
	e_pyramid = 1,       // p0 = rows, p1 = number of pyramids (side by side); f0 = shard index, f1 = shard count
	                     //   (f1 >= 2: build only the pyramids k with k % f1 == f0, at the positions of the full scene)
	e_tumbler = 2,       // p0 = boxes per side of the pre-placed grid, p1 = containers side by side (0 / 1: one) ; f0 = half size S (0 -> auto)
	e_field = 3,         // p0 = bodies, p1 = bullet count ; f0 = arena half length R (0 -> auto density), f1 = max radius
	e_piles = 4,         // p0 = piles, p1 = boxes per pile
	e_rain = 5,          // p0 = bodies (mixed circles / boxes / polygons dropped on a box ground)
	e_circleStack = 6,   // p0 = columns, p1 = circles per column, on an edge ground
	e_sensors = 8,       // p0 = falling bodies ; static sensor regions (box, circle, polygon) + one dynamic body that carries
	                     //   a proximity sensor: exercises b2Contact::Update's sensor branch (b2TestOverlap = GJK with radii)
	e_ropes = 9,         // p0 = falling bodies, p1 = planks ; distance joints: a plank bridge hung by rigid rods between its
	                     //   planks and the ground, a soft web of four boxes on damped springs, bodies dropped on both
	e_machines = 10,     // p0 = falling bodies, p1 = cantilever segments ; prismatic joints (motor slider between limits, a free
	                     //   vertical slider resting on its lower limit, a locked one) and weld joints (rigid and soft cantilevers,
	                     //   a welded free-falling pair), a gear train, two pulleys, bodies dropped over all of them
	e_lifecycle = 12,    // p0 = falling bodies ; a scripted tour of the life-cycle API and the mutators between steps: DestroyBody,
	                     //   DestroyFixture, CreateBody / CreateFixture afterwards (proxy ids are reused), SetTransform, SetAwake,
	                     //   SetBullet, ApplyLinearImpulse / ToCenter / ApplyAngularImpulse, SetSensor, SetThickShape,
	                     //   SetFilterData, a wheel joint's spring retuned, and a mouse joint dragged more slowly than the sleep
	                     //   tolerance for more than the time to sleep (the body must stay awake); see LifecycleEdits
	e_vehicles = 11,     // p0 = falling bodies, p1 = cars ; wheel joints (cars with sprung, motor-driven wheels over bumps, one
	                     //   with a rigid axle), rope joints (weights on slack and taut tethers), friction joints (pucks braked
	                     //   against the ground), motor joints (platforms servoed to a pose the step loop keeps moving) and a
	                     //   mouse joint dragging a crate round a circle
	e_chains = 13,       // p0 = falling bodies (every fourth a bullet) ; chain shapes: a closed loop (an arena with a bumpy floor), an
	                     //   open chain with ghost vertices set by hand (a ramp), one without, and a short chain on a kinematic
	                     //   body that travels to and fro: b2ChainAndCircleContact / b2ChainAndPolygonContact, one proxy per child
	e_props = 14,        // p0 = falling bodies (the rain scene) ; a scripted tour of the body / fixture PROPERTY setters between steps:
	                     //   SetLinearDamping, SetAngularDamping, SetGravityScale, SetFixedRotation, SetSleepingAllowed, SetMassData,
	                     //   ResetMassData after b2Fixture::SetDensity, b2Fixture::SetFriction / SetRestitution, and
	                     //   GetLinearVelocityFromLocalPoint feeding an impulse, b2World::ShiftOrigin twice; see PropsEdits
	e_bullets = 7        // p0 = projectiles (every other one flagged bullet), p1 = stack height ; continuous-collision stress:
	                     //   thin static walls + edge ground + box stacks hit by fast small bodies
};

struct SceneParams
{
	int scene;
	int p0, p1;
	float f0, f1;
	uint32_t seed;
};

struct Scene
{
	std::vector<b2Body*> bodies; // creation order == body index used by every dump
	b2Joint* joint;
	b2Joint* drag;                // mouse joint whose target the step loop moves on a circle
	std::vector<b2Joint*> servos; // motor joints whose linear / angular offset the step loop moves along a fixed path
	int servoStep;
	bool lifecycle;     // the step loop runs LifecycleEdits before every step
	int lifecycleStep;
	b2Joint* slowDrag;  // lifecycle scene: mouse joint with a slowly moving target
	b2Joint* spring;    // lifecycle scene: wheel joint whose spring is retuned
	uint32_t lifecycleSeed;
	bool props;         // the step loop runs PropsEdits before every step
	int propsStep;
	bool sliderBounces; // joint is a prismatic motor slider whose motor is reversed by the step loop at either limit
	float dtDefault;
	int velIters, posIters;
	Scene() : joint(NULL), drag(NULL), servoStep(0), lifecycle(false), lifecycleStep(0), slowDrag(NULL), spring(NULL), lifecycleSeed(1), props(false), propsStep(0), sliderBounces(false), dtDefault(1.0f / 60.0f), velIters(8), posIters(3) {}
};

inline b2Body* AddBody(Scene& s, b2World* w, const b2BodyDef& bd)
{
	b2Body* b = w->CreateBody(&bd);
	s.bodies.push_back(b);
	return b;
}

inline void BuildHelloWorld(Scene& s, b2World* w)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	b2BodyDef groundBodyDef;
	groundBodyDef.position.Set(0.0f, -10.0f);
	b2Body* groundBody = AddBody(s, w, groundBodyDef);
	b2PolygonShape groundBox;
	groundBox.SetAsBox(50.0f, 10.0f);
	groundBody->CreateFixture(&groundBox, 0.0f);

	b2BodyDef bodyDef;
	bodyDef.type = b2_dynamicBody;
	bodyDef.position.Set(0.0f, 4.0f);
	b2Body* body = AddBody(s, w, bodyDef);
	b2PolygonShape dynamicBox;
	dynamicBox.SetAsBox(1.0f, 1.0f);
	b2FixtureDef fixtureDef;
	fixtureDef.shape = &dynamicBox;
	fixtureDef.density = 1.0f;
	fixtureDef.friction = 0.3f;
	body->CreateFixture(&fixtureDef);
	s.velIters = 6;
	s.posIters = 2;
}

// rows -> rows*(rows+1)/2 boxes per pyramid; `count` pyramids side by side on one ground edge.
inline void BuildPyramid(Scene& s, b2World* w, int rows, int count, int shard = 0, int shardCount = 1)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	if (count < 1) count = 1;
	float width = 1.125f * (float)rows + 10.0f;
	float L = 0.5f * width * (float)count + 40.0f;
	{
		b2BodyDef bd;
		b2Body* ground = AddBody(s, w, bd);
		b2EdgeShape shape;
		shape.Set(b2Vec2(-L, 0.0f), b2Vec2(L, 0.0f));
		ground->CreateFixture(&shape, 0.0f);
	}
	float a = 0.5f;
	b2PolygonShape shape;
	shape.SetAsBox(a, a);
	for (int k = 0; k < count; ++k)
	{
		if (shardCount > 1 && (k % shardCount) != shard) continue;
		float x0 = -0.5f * width * (float)count + width * (float)k + 5.0f;
		b2Vec2 x(x0, 0.75f);
		b2Vec2 y;
		b2Vec2 deltaX(0.5625f, 1.25f);
		b2Vec2 deltaY(1.125f, 0.0f);
		for (int i = 0; i < rows; ++i)
		{
			y = x;
			for (int j = i; j < rows; ++j)
			{
				b2BodyDef bd;
				bd.type = b2_dynamicBody;
				bd.position = y;
				b2Body* body = AddBody(s, w, bd);
				body->CreateFixture(&shape, 5.0f);
				y += deltaY;
			}
			x += deltaX;
		}
	}
}

// Hollow square container driven by a revolute motor, n*n small boxes pre-placed on a grid.
// `count` containers side by side (0 / 1: the Testbed's one), each pinned to the shared ground body by its own motor joint
// and filled with its own n x n boxes; `count` > 1 is the N-GPU form of config 3: one container per rank of a world sharded
// by spatial ownership, bodies created container after container.
inline void BuildTumbler(Scene& s, b2World* w, int n, float S, int count = 1)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	const float pitch = 0.3f;
	if (S <= 0.0f)
	{
		S = 0.5f * pitch * (float)n + 1.0f;
		if (S < 10.0f) S = 10.0f;
	}
	if (count < 1) count = 1;
	b2Body* ground;
	{
		b2BodyDef bd;
		ground = AddBody(s, w, bd);
	}
	for (int t = 0; t < count; ++t)
	{
		// (a container's corners sweep a circle of radius sqrt(2) (S + 0.5): 3 S + 4 between the centres keeps even the fat AABBs
		// of two revolving neighbours apart - no contact, no shared component, one container per rank for ever)
		// (B2H_TUMBLER_CLOSE=1: 2 S + 4, close enough for the fat AABBs of two revolving neighbours to overlap - a test of
		// components that merge over an ownership boundary, tests/test_gpu_spatial.py)
		const float x0 = (float)t * ((getenv("B2H_TUMBLER_CLOSE") ? 2.0f : 3.0f) * S + 4.0f);
		{
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.allowSleep = false;
			bd.position.Set(x0, S);
			b2Body* body = AddBody(s, w, bd);
			b2PolygonShape shape;
			shape.SetAsBox(0.5f, S, b2Vec2(S, 0.0f), 0.0f);
			body->CreateFixture(&shape, 5.0f);
			shape.SetAsBox(0.5f, S, b2Vec2(-S, 0.0f), 0.0f);
			body->CreateFixture(&shape, 5.0f);
			shape.SetAsBox(S, 0.5f, b2Vec2(0.0f, S), 0.0f);
			body->CreateFixture(&shape, 5.0f);
			shape.SetAsBox(S, 0.5f, b2Vec2(0.0f, -S), 0.0f);
			body->CreateFixture(&shape, 5.0f);

			b2RevoluteJointDef jd;
			jd.bodyA = ground;
			jd.bodyB = body;
			jd.localAnchorA.Set(x0, S);
			jd.localAnchorB.Set(0.0f, 0.0f);
			jd.referenceAngle = 0.0f;
			jd.motorSpeed = 0.05f * b2_pi;
			jd.maxMotorTorque = 1e8f;
			jd.enableMotor = true;
			b2Joint* joint = w->CreateJoint(&jd);
			if (t == 0) s.joint = joint;
		}
		b2PolygonShape box;
		box.SetAsBox(0.125f, 0.125f);
		float start = -0.5f * pitch * (float)(n - 1);
		for (int i = 0; i < n; ++i)
		{
			for (int j = 0; j < n; ++j)
			{
				b2BodyDef bd;
				bd.type = b2_dynamicBody;
				bd.position.Set(x0 + start + pitch * (float)j, S + start + pitch * (float)i);
				b2Body* body = AddBody(s, w, bd);
				body->CreateFixture(&box, 1.0f);
			}
		}
	}
}

// Zero-gravity random field of alternating circles / regular polygons inside 4 thick static walls.
inline void BuildField(Scene& s, b2World* w, int count, int bullets, float R, float maxRadius, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, 0.0f));
	const float kMinRadius = 0.5f;
	if (maxRadius < kMinRadius) maxRadius = 5.0f;
	if (R <= 0.0f)
	{
		// reference density: 20000 bodies in R = 2000 (ManyBodies.h) -> area per body 800
		R = 0.5f * sqrtf(800.0f * (float)count);
		if (R < 50.0f) R = 50.0f;
	}
	const float wallHalf = 5.0f;
	{
		b2BodyDef bd;
		b2Body* ground = AddBody(s, w, bd);
		b2PolygonShape wall;
		b2FixtureDef fd;
		fd.shape = &wall;
		fd.thickShape = true;
		wall.SetAsBox(R, wallHalf, b2Vec2(0.0f, -R), 0.0f);
		ground->CreateFixture(&fd);
		wall.SetAsBox(R, wallHalf, b2Vec2(0.0f, R), 0.0f);
		ground->CreateFixture(&fd);
		wall.SetAsBox(wallHalf, R, b2Vec2(-R, 0.0f), 0.0f);
		ground->CreateFixture(&fd);
		wall.SetAsBox(wallHalf, R, b2Vec2(R, 0.0f), 0.0f);
		ground->CreateFixture(&fd);
	}
	Pcg32 rng(seed);
	const float range = R - wallHalf - maxRadius;
	b2PolygonShape polygon;
	b2CircleShape circle;
	for (int i = 0; i < count; ++i)
	{
		float radius = rng.Range(kMinRadius, maxRadius);
		float speed = 8.0f * radius;
		float x = rng.Range(-range, range);
		float y = rng.Range(-range, range);
		float a = rng.Range(0.0f, 2.0f * b2_pi);
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		b2FixtureDef fd;
		fd.density = 1.0f;
		if (i < bullets)
		{
			speed = 120.0f;
			radius = kMinRadius;
			bd.bullet = true;
			fd.density = 25.0f;
		}
		bd.position = b2Vec2(x, y);
		bd.angle = a;
		bd.angularDamping = 0.25f;
		b2Vec2 n(rng.Signed(), rng.Signed());
		n.Normalize();
		bd.linearVelocity = speed * n;
		b2Body* body = AddBody(s, w, bd);
		if ((i & 1) == 0)
		{
			circle.m_radius = radius;
			fd.shape = &circle;
		}
		else
		{
			b2Vec2 vertices[b2_maxPolygonVertices];
			int vc = i % b2_maxPolygonVertices;
			if (vc < 3) vc = 3;
			float arc = 2.0f * b2_pi / (float)vc;
			for (int v = 0; v < vc; ++v)
			{
				float ang = ((float)v + 1.0f) * arc;
				b2Rot q(ang);
				vertices[v] = b2Mul(q, b2Vec2(radius, 0.0f));
			}
			polygon.Set(vertices, vc);
			fd.shape = &polygon;
		}
		if (radius > 1.0f)
		{
			fd.thickShape = true;
		}
		body->CreateFixture(&fd);
	}
}

// Many independent small stacks on one polygon ground: many small multi-contact islands under gravity.
inline void BuildPiles(Scene& s, b2World* w, int piles, int height, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	float L = 2.0f * (float)piles + 10.0f;
	{
		b2BodyDef bd;
		bd.position.Set(0.0f, -1.0f);
		b2Body* ground = AddBody(s, w, bd);
		b2PolygonShape shape;
		shape.SetAsBox(L, 1.0f);
		ground->CreateFixture(&shape, 0.0f);
	}
	Pcg32 rng(seed);
	b2PolygonShape box;
	b2CircleShape ball;
	for (int p = 0; p < piles; ++p)
	{
		float x = -2.0f * (float)piles + 4.0f * (float)p + 2.0f;
		float y = 0.0f;
		for (int k = 0; k < height; ++k)
		{
			float hw = rng.Range(0.3f, 0.6f);
			float hh = rng.Range(0.2f, 0.4f);
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.position.Set(x + rng.Range(-0.15f, 0.15f), y + hh + 0.02f);
			bd.angle = rng.Range(-0.05f, 0.05f);
			b2Body* body = AddBody(s, w, bd);
			b2FixtureDef fd;
			fd.density = rng.Range(0.5f, 3.0f);
			fd.friction = rng.Range(0.1f, 0.9f);
			fd.restitution = (k % 3 == 0) ? 0.3f : 0.0f;
			if ((p + k) % 4 == 3)
			{
				ball.m_radius = hh;
				fd.shape = &ball;
			}
			else
			{
				box.SetAsBox(hw, hh);
				fd.shape = &box;
			}
			body->CreateFixture(&fd);
			y += 2.0f * hh + 0.04f;
		}
	}
}

// Mixed shapes falling on a box ground with side walls.
inline void BuildRain(Scene& s, b2World* w, int count, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	int cols = 1;
	while (cols * cols < count) ++cols;
	float W = 1.5f * (float)cols * 0.5f + 3.0f;
	{
		b2BodyDef bd;
		b2Body* ground = AddBody(s, w, bd);
		b2PolygonShape shape;
		shape.SetAsBox(W, 1.0f, b2Vec2(0.0f, -1.0f), 0.0f);
		ground->CreateFixture(&shape, 0.0f);
		shape.SetAsBox(1.0f, 4.0f * W, b2Vec2(-W - 1.0f, 4.0f * W - 2.0f), 0.0f);
		ground->CreateFixture(&shape, 0.0f);
		shape.SetAsBox(1.0f, 4.0f * W, b2Vec2(W + 1.0f, 4.0f * W - 2.0f), 0.0f);
		ground->CreateFixture(&shape, 0.0f);
	}
	Pcg32 rng(seed);
	b2PolygonShape poly;
	b2CircleShape ball;
	for (int i = 0; i < count; ++i)
	{
		int cx = i % cols, cy = i / cols;
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(-W + 2.0f + 1.5f * (float)cx + rng.Range(-0.2f, 0.2f), 1.0f + 1.5f * (float)cy);
		bd.angle = rng.Range(0.0f, 2.0f * b2_pi);
		bd.linearVelocity.Set(rng.Range(-1.0f, 1.0f), rng.Range(-2.0f, 0.0f));
		bd.angularVelocity = rng.Range(-1.0f, 1.0f);
		b2Body* body = AddBody(s, w, bd);
		b2FixtureDef fd;
		fd.density = rng.Range(0.5f, 4.0f);
		fd.friction = rng.Range(0.0f, 1.0f);
		fd.restitution = rng.Range(0.0f, 0.5f);
		int kind = i % 3;
		if (kind == 0)
		{
			ball.m_radius = rng.Range(0.2f, 0.5f);
			fd.shape = &ball;
		}
		else if (kind == 1)
		{
			poly.SetAsBox(rng.Range(0.2f, 0.5f), rng.Range(0.2f, 0.5f));
			fd.shape = &poly;
		}
		else
		{
			b2Vec2 vertices[b2_maxPolygonVertices];
			int vc = 3 + (i / 3) % 6;
			float r = rng.Range(0.25f, 0.5f);
			float arc = 2.0f * b2_pi / (float)vc;
			for (int v = 0; v < vc; ++v)
			{
				b2Rot q(((float)v + 0.5f) * arc);
				vertices[v] = b2Mul(q, b2Vec2(r, 0.0f));
			}
			poly.Set(vertices, vc);
			fd.shape = &poly;
		}
		body->CreateFixture(&fd);
	}
}

// Columns of circles resting on an edge: edge-circle + circle-circle manifolds.
inline void BuildCircleStack(Scene& s, b2World* w, int columns, int height)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	float L = 1.5f * (float)columns + 10.0f;
	{
		b2BodyDef bd;
		b2Body* ground = AddBody(s, w, bd);
		b2EdgeShape shape;
		shape.Set(b2Vec2(-L, 0.0f), b2Vec2(L, 0.0f));
		ground->CreateFixture(&shape, 0.0f);
	}
	b2CircleShape ball;
	ball.m_radius = 0.5f;
	for (int c = 0; c < columns; ++c)
	{
		for (int k = 0; k < height; ++k)
		{
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.position.Set(-1.5f * (float)columns + 3.0f * (float)c, 0.55f + 1.05f * (float)k);
			b2Body* body = AddBody(s, w, bd);
			body->CreateFixture(&ball, 1.0f + 0.25f * (float)(k & 3));
		}
	}
}

// Sensors: bodies rain through static sensor regions onto an edge ground; one dynamic "probe" carries a solid box and a
// larger circular proximity sensor. Sensor contacts never enter the solver; their touching flag is b2TestOverlap.
inline void BuildSensors(Scene& s, b2World* w, int count, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	Pcg32 rng(seed ? seed : 11u);
	{
		b2BodyDef bd;
		b2Body* ground = AddBody(s, w, bd);
		b2EdgeShape edge;
		edge.Set(b2Vec2(-30.0f, 0.0f), b2Vec2(30.0f, 0.0f));
		ground->CreateFixture(&edge, 0.0f);
		b2FixtureDef fd;
		fd.isSensor = true;
		b2PolygonShape zone;
		zone.SetAsBox(6.0f, 1.0f, b2Vec2(-8.0f, 6.0f), 0.2f);
		fd.shape = &zone;
		ground->CreateFixture(&fd);
		b2CircleShape disc;
		disc.m_radius = 2.5f;
		disc.m_p.Set(6.0f, 5.0f);
		fd.shape = &disc;
		ground->CreateFixture(&fd);
		b2Vec2 tri[3] = { b2Vec2(-2.0f, 1.0f), b2Vec2(3.0f, 1.5f), b2Vec2(0.5f, 4.0f) };
		b2PolygonShape wedge;
		wedge.Set(tri, 3);
		fd.shape = &wedge;
		ground->CreateFixture(&fd);
	}
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(0.0f, 9.0f);
		bd.angularVelocity = 1.0f;
		b2Body* probe = AddBody(s, w, bd);
		b2PolygonShape core;
		core.SetAsBox(0.4f, 0.4f);
		probe->CreateFixture(&core, 2.0f);
		b2CircleShape halo;
		halo.m_radius = 1.5f;
		b2FixtureDef fd;
		fd.shape = &halo;
		fd.isSensor = true;
		fd.density = 0.0f;
		probe->CreateFixture(&fd);
	}
	for (int i = 0; i < count; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(rng.Range(-14.0f, 10.0f), 8.0f + 1.2f * (float)(i / 6) + rng.Range(0.0f, 0.5f));
		bd.angle = rng.Range(0.0f, 2.0f * b2_pi);
		bd.linearVelocity.Set(rng.Range(-2.0f, 2.0f), rng.Range(-3.0f, 0.0f));
		b2Body* body = AddBody(s, w, bd);
		if (i % 3 == 0)
		{
			b2CircleShape ball;
			ball.m_radius = rng.Range(0.2f, 0.45f);
			body->CreateFixture(&ball, 1.0f);
		}
		else
		{
			b2PolygonShape box;
			box.SetAsBox(rng.Range(0.2f, 0.5f), rng.Range(0.2f, 0.4f));
			body->CreateFixture(&box, 1.0f);
		}
	}
}

// Chain shapes (b2ChainShape.h:32): every child segment is a proxy of its own with its neighbours as ghost vertices, so bodies
// slide over the inner vertices without catching. Circles, boxes and polygons are dropped into a closed loop whose floor is
// a saw-tooth, onto a ramp (open chain, ghost vertices given by SetPrevVertex / SetNextVertex) and onto a plain open chain;
// a kinematic body carries a three-segment chain back and forth through the pile (its children move every step).
inline void BuildChains(Scene& s, b2World* w, int count, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	Pcg32 rng(seed ? seed : 17u);
	{
		b2BodyDef bd;
		b2Body* ground = AddBody(s, w, bd);
		b2Vec2 loop[14];
		int n = 0;
		loop[n++].Set(-20.0f, 0.0f);
		for (int k = 0; k < 9; ++k) loop[n++].Set(-16.0f + 4.0f * (float)k, (k & 1) ? 0.9f : 0.0f);
		loop[n++].Set(20.0f, 0.0f);
		loop[n++].Set(21.0f, 18.0f);
		loop[n++].Set(0.0f, 22.0f);
		loop[n++].Set(-21.0f, 18.0f);
		b2ChainShape arena;
		arena.CreateLoop(loop, n);
		b2FixtureDef fd;
		fd.shape = &arena;
		fd.friction = 0.4f;
		ground->CreateFixture(&fd);

		b2Vec2 ramp[4] = { b2Vec2(-15.0f, 9.0f), b2Vec2(-9.0f, 7.5f), b2Vec2(-4.0f, 7.0f), b2Vec2(-1.0f, 7.4f) };
		b2ChainShape slide;
		slide.CreateChain(ramp, 4);
		slide.SetPrevVertex(b2Vec2(-18.0f, 11.0f));
		slide.SetNextVertex(b2Vec2(1.0f, 8.5f));
		ground->CreateFixture(&slide, 0.0f);

		b2Vec2 shelf[3] = { b2Vec2(4.0f, 6.0f), b2Vec2(9.0f, 6.2f), b2Vec2(14.0f, 8.0f) };
		b2ChainShape plain;
		plain.CreateChain(shelf, 3);
		ground->CreateFixture(&plain, 0.0f);
	}
	{
		b2BodyDef bd;
		bd.type = b2_kinematicBody;
		bd.position.Set(-10.0f, 3.0f);
		bd.linearVelocity.Set(2.5f, 0.0f);
		bd.angularVelocity = 0.15f;
		b2Body* cart = AddBody(s, w, bd);
		b2Vec2 scoop[4] = { b2Vec2(-2.0f, 0.8f), b2Vec2(-1.0f, 0.0f), b2Vec2(1.0f, 0.0f), b2Vec2(2.0f, 0.8f) };
		b2ChainShape bucket;
		bucket.CreateChain(scoop, 4);
		cart->CreateFixture(&bucket, 0.0f);
	}
	for (int i = 0; i < count; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(rng.Range(-16.0f, 16.0f), 9.0f + 1.1f * (float)(i / 8) + rng.Range(0.0f, 0.6f));
		bd.angle = rng.Range(0.0f, 2.0f * b2_pi);
		bd.linearVelocity.Set(rng.Range(-3.0f, 3.0f), rng.Range(-6.0f, 0.0f));
		bd.bullet = (i % 4) == 3;
		if (bd.bullet) bd.linearVelocity.Set(rng.Range(-30.0f, 30.0f), rng.Range(-60.0f, -30.0f));
		b2Body* body = AddBody(s, w, bd);
		if (i % 3 == 0)
		{
			b2CircleShape ball;
			ball.m_radius = rng.Range(0.15f, 0.5f);
			body->CreateFixture(&ball, 1.0f);
		}
		else if (i % 3 == 1)
		{
			b2PolygonShape box;
			box.SetAsBox(rng.Range(0.2f, 0.55f), rng.Range(0.15f, 0.4f));
			body->CreateFixture(&box, 1.0f);
		}
		else
		{
			b2Vec2 pts[5];
			const float r = rng.Range(0.25f, 0.5f);
			for (int k = 0; k < 5; ++k)
			{
				const float a = 2.0f * b2_pi * (float)k / 5.0f + rng.Range(-0.3f, 0.3f);
				pts[k].Set(r * cosf(a), r * sinf(a));
			}
			b2PolygonShape poly;
			poly.Set(pts, 5);
			body->CreateFixture(&poly, 1.5f);
		}
	}
}

// Continuous-collision stress: a 40 x 30 room made of an edge floor and thin (0.1 wide) polygon walls, a few
// box stacks inside, and fast small projectiles (polygons and circles; every other one is a bullet body)
// fired through it. Fast non-bullet bodies still get TOI against the static walls; bullets also against the stacks.
// Distance joints in both regimes (rigid rod: frequencyHz = 0, spring: > 0) mixed with contacts in the same islands.
inline void BuildRopes(Scene& s, b2World* w, int count, int planks, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	Pcg32 rng(seed ? seed : 23u);
	if (planks < 2) planks = 2;
	b2Body* ground;
	{
		b2BodyDef bd;
		ground = AddBody(s, w, bd);
		b2EdgeShape edge;
		edge.Set(b2Vec2(-40.0f, 0.0f), b2Vec2(40.0f, 0.0f));
		ground->CreateFixture(&edge, 0.0f);
	}
	// bridge: planks of half length 0.35, 1.1 apart along a parabola that sags 2.5 below its two ground anchors; neighbours
	// are tied end to end by rods whose rest length is the initial gap
	{
		b2PolygonShape plank;
		const float hx = 0.35f;
		plank.SetAsBox(hx, 0.125f);
		b2FixtureDef fd;
		fd.shape = &plank;
		fd.density = 4.0f;
		fd.friction = 0.4f;
		const float pitch = 1.1f, top = 8.0f, sag = 2.5f;
		const float x0 = -20.0f;
		b2Body* prev = ground;
		b2Vec2 prevAnchor(x0 - pitch + hx, top);
		for (int i = 0; i < planks; ++i)
		{
			const float t = (float)(2 * i - (planks - 1)) / (float)(planks + 1);
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.position.Set(x0 + pitch * (float)i, top - sag * (1.0f - t * t));
			b2Body* body = AddBody(s, w, bd);
			body->CreateFixture(&fd);
			b2DistanceJointDef jd;
			jd.Initialize(prev, body, prevAnchor, b2Vec2(bd.position.x - hx, bd.position.y));
			w->CreateJoint(&jd);
			prev = body;
			prevAnchor.Set(bd.position.x + hx, bd.position.y);
		}
		b2DistanceJointDef jd;
		jd.Initialize(prev, ground, prevAnchor, b2Vec2(prevAnchor.x + pitch - hx, top));
		w->CreateJoint(&jd);
	}
	// web: four boxes on springs to each other and to four ground anchors (frequencies 2..4 Hz)
	{
		b2PolygonShape box;
		box.SetAsBox(0.5f, 0.5f);
		b2Body* web[4];
		const float cx = 12.0f, cy = 10.0f;
		const float ox[4] = { -3.0f, 3.0f, 3.0f, -3.0f }, oy[4] = { -3.0f, -3.0f, 3.0f, 3.0f };
		for (int i = 0; i < 4; ++i)
		{
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.position.Set(cx + ox[i], cy + oy[i]);
			web[i] = AddBody(s, w, bd);
			web[i]->CreateFixture(&box, 5.0f);
		}
		for (int i = 0; i < 4; ++i)
		{
			b2DistanceJointDef jd;
			jd.frequencyHz = 2.0f + 0.5f * (float)i;
			jd.dampingRatio = 0.1f * (float)i;
			jd.Initialize(ground, web[i], b2Vec2(cx + 2.5f * ox[i], cy + 2.5f * oy[i]), web[i]->GetWorldPoint(b2Vec2(0.17f * ox[i], 0.17f * oy[i])));
			w->CreateJoint(&jd);
			const int k = (i + 1) & 3;
			jd.frequencyHz = 4.0f;
			jd.dampingRatio = 0.5f;
			jd.collideConnected = (i & 1) != 0;
			jd.Initialize(web[i], web[k], web[i]->GetWorldPoint(b2Vec2(0.5f * (ox[k] - ox[i]) / 6.0f, 0.5f * (oy[k] - oy[i]) / 6.0f)),
				web[k]->GetWorldPoint(b2Vec2(0.5f * (ox[i] - ox[k]) / 6.0f, 0.5f * (oy[i] - oy[k]) / 6.0f)));
			w->CreateJoint(&jd);
		}
	}
	// bodies dropped over the bridge and the web
	for (int i = 0; i < count; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(rng.Range(-20.0f, 16.0f), rng.Range(8.0f, 30.0f));
		bd.angle = rng.Range(-1.0f, 1.0f);
		b2Body* body = AddBody(s, w, bd);
		if (i & 1)
		{
			b2CircleShape c;
			c.m_radius = rng.Range(0.15f, 0.4f);
			body->CreateFixture(&c, 1.0f);
		}
		else
		{
			b2PolygonShape b;
			b.SetAsBox(rng.Range(0.15f, 0.45f), rng.Range(0.15f, 0.45f));
			body->CreateFixture(&b, 1.0f);
		}
	}
}

// Prismatic and weld joints, every limit / motor / softness branch, sharing islands with contacts.
inline void BuildMachines(Scene& s, b2World* w, int count, int segments, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	Pcg32 rng(seed ? seed : 31u);
	if (segments < 1) segments = 1;
	b2Body* ground;
	{
		b2BodyDef bd;
		ground = AddBody(s, w, bd);
		b2PolygonShape slab;
		slab.SetAsBox(40.0f, 0.5f, b2Vec2(0.0f, -0.5f), 0.0f);
		ground->CreateFixture(&slab, 0.0f);
	}
	b2PolygonShape block;
	block.SetAsBox(1.0f, 1.0f);
	// motor slider along x between -12 and 12, reversed at either limit by the step loop (as MultithreadDemo.h:153-160
	// drives its slider), pushing whatever lies in the way
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(-6.0f, 1.0f);
		bd.allowSleep = false;
		b2Body* body = AddBody(s, w, bd);
		body->CreateFixture(&block, 5.0f);
		b2PrismaticJointDef jd;
		jd.Initialize(ground, body, b2Vec2(0.0f, 1.0f), b2Vec2(1.0f, 0.0f));
		jd.motorSpeed = 6.0f;
		jd.maxMotorForce = 5000.0f;
		jd.enableMotor = true;
		jd.lowerTranslation = -12.0f;
		jd.upperTranslation = 12.0f;
		jd.enableLimit = true;
		s.joint = w->CreateJoint(&jd);
		s.sliderBounces = true;
	}
	// free slider on a tilted axis: falls to its lower limit and stays there under the bodies that land on it
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(-20.0f, 8.0f);
		bd.angle = 0.25f;
		b2Body* body = AddBody(s, w, bd);
		body->CreateFixture(&block, 2.0f);
		b2PrismaticJointDef jd;
		jd.Initialize(ground, body, bd.position, b2Vec2(0.3f, 2.0f)); // not normalised on purpose
		jd.lowerTranslation = -3.0f;
		jd.upperTranslation = 2.0f;
		jd.enableLimit = true;
		w->CreateJoint(&jd);
	}
	// locked slider (equal limits) holding a shelf, and an unlimited one with a weak motor working against gravity
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(22.0f, 6.0f);
		b2Body* body = AddBody(s, w, bd);
		b2PolygonShape shelf;
		shelf.SetAsBox(3.0f, 0.25f);
		body->CreateFixture(&shelf, 1.0f);
		b2PrismaticJointDef jd;
		jd.Initialize(ground, body, bd.position, b2Vec2(0.0f, 1.0f));
		jd.lowerTranslation = 0.0f;
		jd.upperTranslation = 0.0f;
		jd.enableLimit = true;
		jd.enableMotor = true;
		jd.motorSpeed = 1.0f;
		jd.maxMotorForce = 100.0f;
		w->CreateJoint(&jd);

		bd.position.Set(30.0f, 4.0f);
		b2Body* lift = AddBody(s, w, bd);
		lift->CreateFixture(&shelf, 1.0f);
		jd.Initialize(ground, lift, bd.position, b2Vec2(0.0f, 1.0f));
		jd.enableLimit = false;
		jd.motorSpeed = 0.5f;
		jd.maxMotorForce = 40.0f;
		w->CreateJoint(&jd);
	}
	// cantilevers: rigid welds from a ground post at y = 10, soft welds at y = 14
	for (int soft = 0; soft < 2; ++soft)
	{
		b2PolygonShape seg;
		seg.SetAsBox(0.5f, 0.125f);
		b2Body* prev = ground;
		const float y = soft ? 14.0f : 10.0f;
		for (int i = 0; i < segments; ++i)
		{
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.position.Set(4.5f + (float)i, y);
			b2Body* body = AddBody(s, w, bd);
			body->CreateFixture(&seg, 20.0f);
			b2WeldJointDef jd;
			jd.frequencyHz = soft ? 5.0f : 0.0f;
			jd.dampingRatio = soft ? 0.7f : 0.0f;
			jd.Initialize(prev, body, b2Vec2(4.0f + (float)i, y));
			w->CreateJoint(&jd);
			prev = body;
		}
	}
	// a welded pair in free fall (circle + box), fixed-rotation partner welded to a rotating one
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(14.0f, 20.0f);
		b2Body* a = AddBody(s, w, bd);
		a->CreateFixture(&block, 1.0f);
		bd.position.Set(16.0f, 20.5f);
		bd.fixedRotation = true;
		b2Body* b = AddBody(s, w, bd);
		b2CircleShape c;
		c.m_radius = 0.75f;
		b->CreateFixture(&c, 1.0f);
		b2WeldJointDef jd;
		jd.Initialize(a, b, b2Vec2(15.0f, 20.0f));
		w->CreateJoint(&jd);
	}
	// gear train: a small motorised disc drives a big one (revolute - revolute gear) which lifts a rack on a prismatic
	// joint (revolute - prismatic gear); the three bodies are spaced so that they do not touch
	{
		const float y = 12.0f;
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		b2CircleShape disc;
		bd.position.Set(-44.0f, y);
		b2Body* small = AddBody(s, w, bd);
		disc.m_radius = 1.0f;
		small->CreateFixture(&disc, 5.0f);
		b2RevoluteJointDef r1;
		r1.Initialize(ground, small, bd.position);
		r1.enableMotor = true;
		r1.motorSpeed = 1.0f;
		r1.maxMotorTorque = 400.0f;
		b2Joint* j1 = w->CreateJoint(&r1);
		bd.position.Set(-40.9f, y);
		b2Body* big = AddBody(s, w, bd);
		disc.m_radius = 2.0f;
		big->CreateFixture(&disc, 5.0f);
		b2RevoluteJointDef r2;
		r2.Initialize(ground, big, bd.position);
		b2Joint* j2 = w->CreateJoint(&r2);
		bd.position.Set(-38.3f, y);
		b2Body* rack = AddBody(s, w, bd);
		b2PolygonShape bar;
		bar.SetAsBox(0.5f, 5.0f);
		rack->CreateFixture(&bar, 5.0f);
		b2PrismaticJointDef p3;
		p3.Initialize(ground, rack, bd.position, b2Vec2(0.0f, 1.0f));
		p3.lowerTranslation = -5.0f;
		p3.upperTranslation = 5.0f;
		p3.enableLimit = true;
		b2Joint* j3 = w->CreateJoint(&p3);
		b2GearJointDef g;
		g.bodyA = small;
		g.bodyB = big;
		g.joint1 = j1;
		g.joint2 = j2;
		g.ratio = 2.0f;
		w->CreateJoint(&g);
		g.bodyA = big;
		g.bodyB = rack;
		g.joint1 = j2;
		g.joint2 = j3;
		g.ratio = -1.0f / 2.0f;
		w->CreateJoint(&g);
	}
	// pulleys: two pairs of unequal weights over ground points, ratios 1 and 2.5 (the second pair starts out of balance)
	for (int i = 0; i < 2; ++i)
	{
		const float x = -34.0f + 6.0f * (float)i, y = 8.0f;
		b2PolygonShape wgt;
		wgt.SetAsBox(0.6f, 0.8f);
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(x - 1.5f, y);
		b2Body* left = AddBody(s, w, bd);
		left->CreateFixture(&wgt, 3.0f);
		bd.position.Set(x + 1.5f, y - (float)i);
		b2Body* right = AddBody(s, w, bd);
		right->CreateFixture(&wgt, 3.0f + 2.0f * (float)i);
		b2PulleyJointDef jd;
		jd.Initialize(left, right, b2Vec2(x - 1.5f, y + 6.0f), b2Vec2(x + 1.5f, y + 6.0f), b2Vec2(x - 1.5f, y + 0.8f),
			b2Vec2(x + 1.5f, y - (float)i + 0.8f), i ? 2.5f : 1.0f);
		w->CreateJoint(&jd);
	}
	for (int i = 0; i < count; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(rng.Range(-24.0f, 32.0f), rng.Range(16.0f, 40.0f));
		bd.angle = rng.Range(-1.0f, 1.0f);
		b2Body* body = AddBody(s, w, bd);
		if (i % 3 == 0)
		{
			b2CircleShape c;
			c.m_radius = rng.Range(0.15f, 0.4f);
			body->CreateFixture(&c, 1.0f);
		}
		else
		{
			b2PolygonShape b;
			b.SetAsBox(rng.Range(0.15f, 0.45f), rng.Range(0.15f, 0.45f));
			body->CreateFixture(&b, 1.0f);
		}
	}
}

// Wheel, rope, friction and motor joints sharing islands with contacts.
inline void BuildVehicles(Scene& s, b2World* w, int count, int cars, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	Pcg32 rng(seed ? seed : 41u);
	if (cars < 1) cars = 1;
	b2Body* ground;
	{
		b2BodyDef bd;
		ground = AddBody(s, w, bd);
		b2EdgeShape edge;
		edge.Set(b2Vec2(-60.0f, 0.0f), b2Vec2(60.0f, 0.0f));
		ground->CreateFixture(&edge, 0.0f);
		// bumps the cars drive over
		for (int i = 0; i < 12; ++i)
		{
			b2PolygonShape bump;
			bump.SetAsBox(0.6f, 0.15f, b2Vec2(-30.0f + 5.0f * (float)i, 0.15f), (i & 1) ? 0.2f : -0.15f);
			ground->CreateFixture(&bump, 0.0f);
		}
	}
	for (int c = 0; c < cars; ++c)
	{
		const float x = -40.0f + 9.0f * (float)c, y = 1.0f;
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(x, y);
		b2Body* chassis = AddBody(s, w, bd);
		b2PolygonShape hull;
		b2Vec2 pts[6] = { b2Vec2(-1.5f, -0.5f), b2Vec2(1.5f, -0.5f), b2Vec2(1.5f, 0.0f), b2Vec2(0.0f, 0.9f), b2Vec2(-1.15f, 0.9f), b2Vec2(-1.5f, 0.2f) };
		hull.Set(pts, 6);
		chassis->CreateFixture(&hull, 1.0f);
		b2CircleShape tyre;
		tyre.m_radius = 0.4f;
		b2FixtureDef fd;
		fd.shape = &tyre;
		fd.density = 1.0f;
		fd.friction = 0.9f;
		for (int k = 0; k < 2; ++k)
		{
			bd.position.Set(x + (k ? 1.0f : -1.0f), y - 0.65f);
			b2Body* wheel = AddBody(s, w, bd);
			wheel->CreateFixture(&fd);
			b2WheelJointDef jd;
			jd.Initialize(chassis, wheel, wheel->GetPosition(), b2Vec2(0.1f * (float)(c % 3), 1.0f));
			jd.motorSpeed = k ? 0.0f : -12.0f - 2.0f * (float)c;
			jd.maxMotorTorque = k ? 10.0f : 20.0f;
			jd.enableMotor = (k == 0) || (c & 1);
			jd.frequencyHz = (c % 4 == 3) ? 0.0f : 4.0f; // every fourth car: rigid axle (no spring row)
			jd.dampingRatio = 0.7f;
			w->CreateJoint(&jd);
		}
	}
	// tethered weights: anchors on a static beam 12 up, ropes longer (slack) and shorter (taut from the start) than the drop
	for (int i = 0; i < 6; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(20.0f + 2.0f * (float)i, 9.0f);
		b2Body* weight = AddBody(s, w, bd);
		b2PolygonShape box;
		box.SetAsBox(0.4f, 0.4f);
		weight->CreateFixture(&box, 2.0f + (float)i);
		b2RopeJointDef jd;
		jd.bodyA = ground;
		jd.bodyB = weight;
		jd.localAnchorA.Set(20.0f + 2.0f * (float)i + ((i & 1) ? 1.5f : 0.0f), 12.0f);
		jd.localAnchorB.Set(0.0f, 0.4f);
		jd.maxLength = (i < 2) ? 2.0f : 4.0f + 0.5f * (float)i;
		jd.collideConnected = true;
		w->CreateJoint(&jd);
	}
	// pucks thrown along the ground plane of a top-down table (no gravity on them), braked by friction joints
	for (int i = 0; i < 5; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.gravityScale = 0.0f;
		bd.position.Set(-50.0f + 3.0f * (float)i, 20.0f + 1.5f * (float)i);
		bd.linearVelocity.Set(6.0f + (float)i, 1.0f - 0.5f * (float)i);
		bd.angularVelocity = 3.0f - (float)i;
		b2Body* puck = AddBody(s, w, bd);
		b2PolygonShape box;
		box.SetAsBox(0.5f, 0.3f);
		puck->CreateFixture(&box, 1.0f);
		b2FrictionJointDef jd;
		jd.Initialize(ground, puck, puck->GetPosition());
		jd.maxForce = 1.0f + 2.0f * (float)i;
		jd.maxTorque = 0.2f * (float)(i + 1);
		jd.collideConnected = true;
		w->CreateJoint(&jd);
	}
	// servo platforms: motor joints towards a pose that the step loop moves on a figure of eight
	for (int i = 0; i < 3; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(40.0f + 6.0f * (float)i, 8.0f);
		b2Body* plat = AddBody(s, w, bd);
		b2PolygonShape box;
		box.SetAsBox(2.0f, 0.25f);
		b2FixtureDef fd;
		fd.shape = &box;
		fd.friction = 0.6f;
		fd.density = 2.0f;
		plat->CreateFixture(&fd);
		b2MotorJointDef jd;
		jd.Initialize(ground, plat);
		jd.maxForce = 1000.0f;
		jd.maxTorque = 1000.0f;
		jd.correctionFactor = 0.3f + 0.2f * (float)i;
		s.servos.push_back(w->CreateJoint(&jd));
	}
	// a crate picked up off-centre by a mouse joint (as Test::MouseDown does: ground as bodyA, force cap 1000 x mass)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(10.0f, 3.0f);
		bd.angle = 0.3f;
		b2Body* crate = AddBody(s, w, bd);
		b2PolygonShape box;
		box.SetAsBox(0.8f, 0.6f);
		crate->CreateFixture(&box, 2.0f);
		b2MouseJointDef jd;
		jd.bodyA = ground;
		jd.bodyB = crate;
		jd.target.Set(10.5f, 3.3f);
		jd.maxForce = 1000.0f * crate->GetMass();
		s.drag = w->CreateJoint(&jd);
	}
	for (int i = 0; i < count; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(rng.Range(-45.0f, 56.0f), rng.Range(14.0f, 34.0f));
		bd.angle = rng.Range(-1.0f, 1.0f);
		b2Body* body = AddBody(s, w, bd);
		if (i % 3 == 0)
		{
			b2CircleShape c;
			c.m_radius = rng.Range(0.15f, 0.4f);
			body->CreateFixture(&c, 1.0f);
		}
		else
		{
			b2PolygonShape b;
			b.SetAsBox(rng.Range(0.15f, 0.45f), rng.Range(0.15f, 0.45f));
			body->CreateFixture(&b, 1.0f);
		}
	}
}

inline void BuildBullets(Scene& s, b2World* w, int projectiles, int stackHeight, uint32_t seed)
{
	w->SetGravity(b2Vec2(0.0f, -10.0f));
	{
		b2BodyDef bd;
		b2Body* room = AddBody(s, w, bd);
		b2EdgeShape floor;
		floor.Set(b2Vec2(-20.0f, 0.0f), b2Vec2(20.0f, 0.0f));
		room->CreateFixture(&floor, 0.0f);
		b2PolygonShape wall;
		wall.SetAsBox(0.05f, 15.0f, b2Vec2(-20.0f, 15.0f), 0.0f);
		room->CreateFixture(&wall, 0.0f);
		wall.SetAsBox(0.05f, 15.0f, b2Vec2(20.0f, 15.0f), 0.0f);
		room->CreateFixture(&wall, 0.0f);
		wall.SetAsBox(20.0f, 0.05f, b2Vec2(0.0f, 30.0f), 0.0f);
		room->CreateFixture(&wall, 0.0f);
		// a slanted thin divider
		wall.SetAsBox(0.05f, 4.0f, b2Vec2(6.0f, 4.0f), 0.3f);
		room->CreateFixture(&wall, 0.0f);
	}
	Pcg32 rng(seed);
	b2PolygonShape box;
	box.SetAsBox(0.5f, 0.5f);
	for (int k = 0; k < 3; ++k)
	{
		float x = -10.0f + 5.0f * (float)k;
		for (int i = 0; i < stackHeight; ++i)
		{
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.position.Set(x + 0.02f * rng.Signed(), 0.5f + 1.0f * (float)i);
			b2Body* body = AddBody(s, w, bd);
			body->CreateFixture(&box, 1.0f);
		}
	}
	b2CircleShape ball;
	b2PolygonShape dart;
	for (int i = 0; i < projectiles; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.bullet = (i & 1) == 0;
		bd.position.Set(rng.Range(-18.0f, 18.0f), rng.Range(12.0f, 28.0f));
		bd.angle = rng.Range(0.0f, 2.0f * b2_pi);
		float speed = rng.Range(60.0f, 250.0f);
		b2Vec2 n(rng.Signed(), rng.Signed() - 0.5f);
		n.Normalize();
		bd.linearVelocity = speed * n;
		bd.angularVelocity = rng.Range(-20.0f, 20.0f);
		b2Body* body = AddBody(s, w, bd);
		b2FixtureDef fd;
		fd.density = 5.0f;
		fd.restitution = (i % 3 == 0) ? 0.5f : 0.0f;
		if (i % 4 < 2)
		{
			ball.m_radius = rng.Range(0.1f, 0.3f);
			fd.shape = &ball;
		}
		else
		{
			dart.SetAsBox(rng.Range(0.1f, 0.4f), rng.Range(0.05f, 0.15f));
			fd.shape = &dart;
		}
		body->CreateFixture(&fd);
	}
}

// ---- life cycle scene -----------------------------------------------------------------------------------------------------
// Ground with two walls, a heap of boxes / discs / two-fixture "dumbbells", a cart on sprung wheel joints, a crate on a
// mouse joint. The edits are made by LifecycleEdits, called by the step loop before every step.
inline void BuildLifecycle(Scene& s, b2World* w, int count, uint32_t seed)
{
	Pcg32 rng(seed);
	s.lifecycle = true;
	s.lifecycleSeed = seed;
	b2BodyDef gd;
	b2Body* ground = AddBody(s, w, gd);
	{
		b2EdgeShape edge;
		edge.Set(b2Vec2(-30.0f, 0.0f), b2Vec2(30.0f, 0.0f));
		ground->CreateFixture(&edge, 0.0f);
		b2PolygonShape wall;
		wall.SetAsBox(0.25f, 4.0f, b2Vec2(-12.0f, 4.0f), 0.0f);
		ground->CreateFixture(&wall, 0.0f);
		wall.SetAsBox(0.05f, 4.0f, b2Vec2(12.0f, 4.0f), 0.0f); // thin wall: bullets test it
		ground->CreateFixture(&wall, 0.0f);
	}
	for (int i = 0; i < count; ++i)
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(rng.Range(-9.0f, 9.0f), 1.0f + 0.6f * (float)(i / 6) + rng.Range(0.0f, 0.2f));
		bd.angle = rng.Range(0.0f, 2.0f * b2_pi);
		b2Body* body = AddBody(s, w, bd);
		b2FixtureDef fd;
		fd.density = 1.0f + 0.5f * (float)(i % 3);
		fd.friction = 0.3f;
		if (i % 5 == 0)
		{
			b2CircleShape disc;
			disc.m_radius = rng.Range(0.2f, 0.4f);
			fd.shape = &disc;
			body->CreateFixture(&fd);
		}
		else if (i % 5 == 1)
		{
			// dumbbell: two fixtures on one body
			b2PolygonShape bar;
			bar.SetAsBox(0.5f, 0.1f);
			fd.shape = &bar;
			body->CreateFixture(&fd);
			b2CircleShape knob;
			knob.m_radius = 0.25f;
			knob.m_p.Set(0.5f, 0.0f);
			fd.shape = &knob;
			body->CreateFixture(&fd);
		}
		else
		{
			b2PolygonShape box;
			box.SetAsBox(rng.Range(0.2f, 0.45f), rng.Range(0.2f, 0.35f));
			fd.shape = &box;
			body->CreateFixture(&fd);
		}
	}
	// cart on two sprung wheels
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(-20.0f, 1.2f);
		b2Body* chassis = AddBody(s, w, bd);
		b2PolygonShape box;
		box.SetAsBox(1.4f, 0.3f);
		chassis->CreateFixture(&box, 1.0f);
		for (int k = 0; k < 2; ++k)
		{
			bd.position.Set(-20.0f + (k == 0 ? -1.0f : 1.0f), 0.5f);
			b2Body* wheel = AddBody(s, w, bd);
			b2CircleShape disc;
			disc.m_radius = 0.4f;
			b2FixtureDef fd;
			fd.shape = &disc;
			fd.density = 1.0f;
			fd.friction = 0.9f;
			wheel->CreateFixture(&fd);
			b2WheelJointDef jd;
			jd.Initialize(chassis, wheel, wheel->GetPosition(), b2Vec2(0.0f, 1.0f));
			jd.frequencyHz = 4.0f;
			jd.dampingRatio = 0.7f;
			jd.enableMotor = k == 0;
			jd.motorSpeed = -3.0f;
			jd.maxMotorTorque = 15.0f;
			b2Joint* j = w->CreateJoint(&jd);
			if (k == 0) s.spring = j;
		}
	}
	// crate held by a mouse joint
	{
		b2BodyDef bd;
		bd.type = b2_dynamicBody;
		bd.position.Set(20.0f, 3.0f);
		b2Body* crate = AddBody(s, w, bd);
		b2PolygonShape box;
		box.SetAsBox(0.5f, 0.5f);
		crate->CreateFixture(&box, 1.0f);
		b2MouseJointDef jd;
		jd.bodyA = ground;
		jd.bodyB = crate;
		jd.target.Set(20.0f, 3.0f);
		jd.maxForce = 1000.0f * crate->GetMass();
		s.slowDrag = w->CreateJoint(&jd);
	}
}

// The script. `first` = index of the first heap body in s.bodies (1), `count` = heap bodies. Destroyed bodies leave a NULL
// in s.bodies (indices stay what they were); bodies created later are appended.
// The property setters of b2Body / b2Fixture between steps (b2Body.h:620-688, b2Body.cpp:310-424, 546-565; b2Fixture.h:306-334),
// on the bodies of the rain scene, by body index so that every backend edits the same bodies.
inline void PropsEdits(Scene& s, b2World* w)
{
	const int step = s.propsStep++;
	const int count = (int)s.bodies.size();
	for (int i = 1; i < count; ++i)
	{
		b2Body* b = s.bodies[(size_t)i];
		if (b == NULL || b->GetType() != b2_dynamicBody) continue;
		if (step == 10 && i % 5 == 0) { b->SetLinearDamping(0.8f); b->SetAngularDamping(1.5f); }
		if (step == 20 && i % 7 == 1) b->SetGravityScale(-0.3f);
		if (step == 20 && i % 7 == 2) b->SetGravityScale(2.0f);
		if (step == 30 && i % 4 == 2) b->SetFixedRotation(true);
		if (step == 50 && i % 6 == 3)
		{
			for (b2Fixture* f = b->GetFixtureList(); f; f = f->GetNext()) f->SetDensity(5.0f * f->GetDensity() + 0.5f);
			b->ResetMassData();
		}
		if (step == 60 && i % 9 == 4)
		{
			b2MassData md;
			md.mass = 3.0f;
			md.center.Set(0.1f, 0.05f);
			md.I = 0.8f;
			b->SetMassData(&md);
		}
		if (step == 70 && i % 3 == 0) b->SetSleepingAllowed(false);
		if (step == 70 && i % 2 == 0)
		{
			for (b2Fixture* f = b->GetFixtureList(); f; f = f->GetNext()) { f->SetFriction(0.9f); f->SetRestitution(0.6f); }
		}
		if (step == 90 && i % 8 == 2) b->SetFixedRotation(false);
		if (step == 90 && i % 7 == 1) b->SetGravityScale(1.0f);
		if (step == 110 && i % 6 == 0) b->SetSleepingAllowed(true);
		if (step % 15 == 7 && i % 11 == 5)
		{
			// a drag against the velocity of an off-centre point of the body
			const b2Vec2 v = b->GetLinearVelocityFromLocalPoint(b2Vec2(0.2f, 0.1f));
			b->ApplyLinearImpulseToCenter(-0.1f * b->GetMass() * v, true);
		}
	}
	// b2World::ShiftOrigin (b2World.cpp:1862-1887): bodies and broad-phase boxes move, nothing else changes
	if (step == 100) w->ShiftOrigin(b2Vec2(3.5f, -1.25f));
	if (step == 150) w->ShiftOrigin(b2Vec2(-40.0f, 17.0f));
	if (step == 80)
	{
		// the ground's material, for the contacts made from now on
		for (b2Fixture* f = s.bodies[0]->GetFixtureList(); f; f = f->GetNext()) { f->SetFriction(0.05f); f->SetRestitution(0.3f); }
	}
}

inline void LifecycleEdits(Scene& s, b2World* w)
{
	const int step = s.lifecycleStep++;
	const int count = (int)s.bodies.size();
	auto alive = [&](int i) -> b2Body* { return i >= 0 && i < count ? s.bodies[(size_t)i] : NULL; };
	// impulses every step on a rotating subset (the ManyBodies floater pattern, ManyBodies.h:29-68)
	for (int k = 0; k < 4; ++k)
	{
		b2Body* b = alive(1 + (step * 3 + k * 7) % 40);
		if (b == NULL || b->GetType() != b2_dynamicBody) continue;
		if (k == 0) b->ApplyLinearImpulseToCenter(b2Vec2(0.2f * b->GetMass(), 0.6f * b->GetMass()), step % 2 == 0);
		else if (k == 1) b->ApplyAngularImpulse(0.05f * b->GetInertia(), true);
		else if (k == 2) b->ApplyLinearImpulse(b2Vec2(-0.3f * b->GetMass(), 0.0f), b->GetWorldPoint(b2Vec2(0.1f, 0.2f)), true);
	}
	if (step == 25 || step == 26 || step == 60)
	{
		// destroy bodies out of the heap (touching contacts, TOI candidates against the ground among them)
		const int victims[3] = { 1 + (step % 5), 9 + (step % 3), 17 };
		for (int k = 0; k < 3; ++k)
		{
			b2Body* b = alive(victims[k]);
			if (b == NULL) continue;
			w->DestroyBody(b);
			s.bodies[(size_t)victims[k]] = NULL;
		}
	}
	if (step == 40)
	{
		// the cart loses a wheel: DestroyBody takes the wheel joint with it
		b2Body* wheel = alive(count - 2);
		if (wheel != NULL && s.spring != NULL)
		{
			s.spring = NULL;
			w->DestroyBody(wheel);
			s.bodies[(size_t)(count - 2)] = NULL;
		}
	}
	if (step == 32 || step == 33)
	{
		// a dumbbell loses its knob (the newest fixture), another one its bar (the oldest)
		for (int i = 1; i < count; ++i)
		{
			b2Body* b = alive(i);
			if (b == NULL || b->GetFixtureList() == NULL || b->GetFixtureList()->GetNext() == NULL) continue;
			if ((i / 5) % 2 == (step % 2))
			{
				b2Fixture* f = step == 32 ? b->GetFixtureList() : b->GetFixtureList()->GetNext();
				b->DestroyFixture(f);
			}
		}
	}
	if (step == 45 || step == 70 || step == 71)
	{
		// new bodies after destroys: the freed proxy ids come back in the tree's LIFO order
		for (int k = 0; k < (step == 45 ? 4 : 2); ++k)
		{
			b2BodyDef bd;
			bd.type = b2_dynamicBody;
			bd.position.Set(-6.0f + 3.0f * (float)k, 9.0f + 0.5f * (float)(step % 3));
			b2Body* body = AddBody(s, w, bd);
			b2PolygonShape box;
			box.SetAsBox(0.3f, 0.3f);
			body->CreateFixture(&box, 2.0f);
			if (k == 1)
			{
				b2CircleShape disc;
				disc.m_radius = 0.2f;
				disc.m_p.Set(0.0f, 0.4f);
				body->CreateFixture(&disc, 1.0f);
			}
		}
	}
	if (step == 50 || step == 51 || step == 90)
	{
		// teleports: onto the heap, far away (new fat AABB), and by a hair (stays inside its fat AABB)
		b2Body* b = alive(5 + step % 7);
		if (b != NULL) b->SetTransform(b2Vec2(step == 50 ? 0.0f : 6.0f, step == 90 ? 1.0f : 8.0f), 0.3f * (float)(step % 4));
		b2Body* c = alive(20);
		if (c != NULL) c->SetTransform(c->GetPosition() + b2Vec2(0.01f, 0.0f), c->GetAngle());
	}
	if (step == 55)
	{
		// bullets: three heap bodies are shot at the thin wall
		for (int k = 0; k < 3; ++k)
		{
			b2Body* b = alive(22 + k);
			if (b == NULL) continue;
			b->SetBullet(true);
			b->SetTransform(b2Vec2(4.0f, 2.0f + 1.5f * (float)k), 0.0f);
			b->SetLinearVelocity(b2Vec2(150.0f, 0.0f));
		}
	}
	if (step == 80)
	{
		for (int k = 0; k < 3; ++k)
		{
			b2Body* b = alive(22 + k);
			if (b != NULL) b->SetBullet(false);
		}
	}
	if (step == 65)
	{
		// put part of the heap to sleep by hand; one of them is woken again two steps later
		for (int i = 26; i < 32; ++i)
		{
			b2Body* b = alive(i);
			if (b != NULL) b->SetAwake(false);
		}
	}
	if (step == 67)
	{
		b2Body* b = alive(27);
		if (b != NULL) b->SetAwake(true);
	}
	if (step == 75 || step == 95)
	{
		// sensors on and off: the bodies fall through their neighbours while their fixture is a sensor
		for (int i = 33; i < 37; ++i)
		{
			b2Body* b = alive(i);
			if (b != NULL && b->GetFixtureList() != NULL) b->GetFixtureList()->SetSensor(step == 75);
		}
	}
	if (step == 85)
	{
		// filter data: these bodies stop colliding with each other (their contacts are filtered again and go)
		for (int i = 1; i < count; i += 3)
		{
			b2Body* b = alive(i);
			if (b == NULL || b->GetFixtureList() == NULL || b->GetType() != b2_dynamicBody) continue;
			b2Filter filter;
			filter.categoryBits = 0x0002;
			filter.maskBits = 0xFFFD;
			b->GetFixtureList()->SetFilterData(filter);
		}
	}
	if (step == 104 || step == 118)
	{
		// bodies switched off (their proxies and contacts go, b2Body.cpp:496-544) and on again (new proxy ids, contacts with
		// the next pair update); one of them gets a fixture while it is off
		for (int i = 6; i < 9; ++i)
		{
			b2Body* b = alive(i);
			if (b != NULL) b->SetActive(step == 118);
		}
	}
	if (step == 110)
	{
		b2Body* b = alive(7);
		if (b != NULL && !b->IsActive())
		{
			b2CircleShape knob;
			knob.m_radius = 0.15f;
			knob.m_p.Set(0.0f, 0.35f);
			b->CreateFixture(&knob, 1.5f);
		}
	}
	if (step == 122 || step == 140)
	{
		// body types (b2Body.cpp:118-188): two heap bodies freeze (static), one becomes a kinematic mover; later they thaw
		b2Body* a = alive(12);
		b2Body* c = alive(13);
		b2Body* k = alive(14);
		if (step == 122)
		{
			if (a != NULL) a->SetType(b2_staticBody);
			if (c != NULL) c->SetType(b2_staticBody);
			if (k != NULL)
			{
				k->SetType(b2_kinematicBody);
				k->SetLinearVelocity(b2Vec2(0.5f, 0.2f));
				k->SetAngularVelocity(0.3f);
			}
		}
		else
		{
			if (a != NULL) a->SetType(b2_dynamicBody);
			if (c != NULL) c->SetType(b2_kinematicBody);
			if (k != NULL) k->SetType(b2_dynamicBody);
		}
	}
	if (step == 58 || step == 100)
	{
		// the thin wall becomes a thick shape (its contacts stop being TOI candidates) and thin again
		b2Fixture* f = s.bodies[0]->GetFixtureList(); // newest ground fixture = the thin wall
		f->SetThickShape(step == 58);
	}
	if (step == 62 && s.spring != NULL)
	{
		b2WheelJoint* wj = static_cast<b2WheelJoint*>(s.spring);
		wj->SetSpringFrequencyHz(1.5f);
		wj->SetSpringDampingRatio(0.2f);
	}
	if (s.slowDrag != NULL)
	{
		// steps 110..175: the target creeps at 0.3 mm per step (0.018 m/s: above the 0.01 m/s sleep tolerance it would not
		// prove anything, so it creeps at 0.1 mm per step = 0.006 m/s), far longer than b2_timeToSleep; then it jumps
		b2MouseJoint* mj = static_cast<b2MouseJoint*>(s.slowDrag);
		if (step >= 110 && step < 176) mj->SetTarget(b2Vec2(20.0f + 0.0001f * (float)(step - 109), 3.0f));
		else if (step == 176) mj->SetTarget(b2Vec2(22.0f, 4.0f));
	}
}

inline void BuildScene(Scene& s, b2World* w, const SceneParams& p)
{
	switch (p.scene)
	{
	case e_helloWorld: BuildHelloWorld(s, w); break;
	case e_pyramid: BuildPyramid(s, w, p.p0, p.p1, (int)p.f0, p.f1 >= 2.0f ? (int)p.f1 : 1); break;
	case e_tumbler: BuildTumbler(s, w, p.p0, p.f0, p.p1); break;
	case e_field: BuildField(s, w, p.p0, p.p1, p.f0, p.f1, p.seed); break;
	case e_piles: BuildPiles(s, w, p.p0, p.p1, p.seed); break;
	case e_rain: BuildRain(s, w, p.p0, p.seed); break;
	case e_circleStack: BuildCircleStack(s, w, p.p0, p.p1); break;
	case e_bullets: BuildBullets(s, w, p.p0, p.p1, p.seed); break;
	case e_sensors: BuildSensors(s, w, p.p0, p.seed); break;
	case e_ropes: BuildRopes(s, w, p.p0, p.p1, p.seed); break;
	case e_machines: BuildMachines(s, w, p.p0, p.p1, p.seed); break;
	case e_vehicles: BuildVehicles(s, w, p.p0, p.p1, p.seed); break;
	case e_lifecycle: BuildLifecycle(s, w, p.p0, p.seed); break;
	case e_chains: BuildChains(s, w, p.p0, p.seed); break;
	case e_props: BuildRain(s, w, p.p0, p.seed); s.props = true; break;
	default: break;
	}
}

} // namespace b2h

#endif
