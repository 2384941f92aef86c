// Host-side implementation of the drop-in b2World / b2Body / b2Fixture / b2Contact: thin binding onto
// the C ABI of libb2hip.so (include/b2hip.h). No physics is computed here: bodies and fixtures are
// forwarded to the device world at creation, Step() is one b2hip_step, getters read the host mirror.
#include "Box2D/Box2D.h"

#include "b2hip.h"

#include <stdio.h>
#include <string.h>
#include <new>
#include <stdint.h>

// the 152-byte device record of a shape / of child `child` of a chain (host/src/b2_shapes.cpp)
void b2ShapeToRecord(const b2Shape* shape, int32 child, void* record152);

// ---- b2World ------------------------------------------------------------------------------------
b2World::b2World(const b2Vec2& gravity) : m_stepExecutor(nullptr), m_shadowValid(false), m_contactManager(this)
{
	m_gravity = gravity;
	m_allowSleep = true;
	m_warmStarting = true;
	m_continuousPhysics = true;
	m_subStepping = false;
	m_autoClearForces = true;
	m_locked = false;
	m_bodyList = nullptr;
	m_jointList = nullptr;
	m_bodyCount = 0;
	m_jointCount = 0;
	m_destructionListener = nullptr;
	m_contactFilter = nullptr;
	m_contactListener = nullptr;
	memset(&m_profile, 0, sizeof(m_profile));
	m_statesValid = false;
	m_contactsValid = false;
	m_fatValid = false;
	m_hip = nullptr;

	b2hip_world_def def;
	def.gravity_x = gravity.x;
	def.gravity_y = gravity.y;
	def.allow_sleep = 1;
	def.warm_starting = 1;
	def.continuous = 1;
	def.sub_stepping = 0;
	def.auto_clear_forces = 1;
	def.device = -1;
	int rc = b2hip_world_create(&def, &m_hip);
	if (rc != B2HIP_OK)
	{
		// Loud failure: there is no CPU implementation of Step() behind this API.
		fprintf(stderr, "b2World: cannot create the HIP world (%d): %s\n", rc, b2hip_last_error());
		m_hip = nullptr;
	}
}

b2World::~b2World()
{
	b2Body* b = m_bodyList;
	while (b)
	{
		b2Body* next = b->m_next;
		b2Fixture* f = b->m_fixtureList;
		while (f)
		{
			b2Fixture* fn = f->m_next;
			if (f->m_shape)
			{
				f->m_shape->~b2Shape();
				b2Free(f->m_shape);
			}
			f->~b2Fixture();
			b2Free(f);
			f = fn;
		}
		b->~b2Body();
		b2Free(b);
		b = next;
	}
	b2Joint* j = m_jointList;
	while (j)
	{
		b2Joint* next = j->m_next;
		j->~b2Joint();
		b2Free(j);
		j = next;
	}
	if (m_hip) b2hip_world_destroy(m_hip);
}

void b2World::PushFlags()
{
	if (m_hip) b2hip_set_flags(m_hip, m_allowSleep, m_warmStarting, m_continuousPhysics, m_subStepping);
}

void b2World::SetAllowSleeping(bool flag)
{
	m_allowSleep = flag;
	PushFlags();
}

void b2World::SetWarmStarting(bool flag)
{
	m_warmStarting = flag;
	PushFlags();
}

void b2World::SetContinuousPhysics(bool flag)
{
	m_continuousPhysics = flag;
	PushFlags();
}

void b2World::SetSubStepping(bool flag)
{
	m_subStepping = flag;
	PushFlags();
}

void b2World::SetGravity(const b2Vec2& gravity)
{
	m_gravity = gravity;
	if (m_hip) b2hip_set_gravity(m_hip, gravity.x, gravity.y);
}

b2Body* b2World::CreateBody(const b2BodyDef* def)
{
	if (IsLocked() || !m_hip) return nullptr;
	b2hip_body_def d;
	d.type = (int)def->type;
	d.px = def->position.x;
	d.py = def->position.y;
	d.angle = def->angle;
	d.vx = def->linearVelocity.x;
	d.vy = def->linearVelocity.y;
	d.w = def->angularVelocity;
	d.linear_damping = def->linearDamping;
	d.angular_damping = def->angularDamping;
	d.gravity_scale = def->gravityScale;
	d.allow_sleep = def->allowSleep;
	d.awake = def->awake;
	d.fixed_rotation = def->fixedRotation;
	d.bullet = def->bullet;
	d.active = def->active;
	int id = b2hip_create_body(m_hip, &d);
	if (id < 0)
	{
		fprintf(stderr, "b2World::CreateBody: %s\n", b2hip_last_error());
		return nullptr;
	}
	void* mem = b2Alloc(sizeof(b2Body));
	b2Body* b = new (mem) b2Body(def, this, id);
	b->m_prev = nullptr;
	b->m_next = m_bodyList;
	if (m_bodyList) m_bodyList->m_prev = b;
	m_bodyList = b;
	++m_bodyCount;
	m_bodies.push_back(b);
	m_statesValid = false;
	return b;
}

// The host-side view of a fixture that is gone (its shape clone included); the id stays reserved on the device
void b2World::DestroyFixtureView(b2Fixture* f)
{
	for (int32 child = 0; child < f->m_childCount; ++child)
	{
		const int32 id = f->m_id + child;
		if (id >= 0 && id < (int32)m_fixtures.size()) m_fixtures[id] = nullptr;
	}
	if (f->m_shape)
	{
		f->m_shape->~b2Shape();
		b2Free(f->m_shape);
	}
	f->~b2Fixture();
	b2Free(f);
}

// b2ContactManager::Destroy (b2ContactManager.cpp:104-107) for the contacts that go with a body (fixture == nullptr) or a
// fixture: a touching contact ends, and the listener hears it while the fixtures still exist.
void b2World::EndContactsOf(b2Body* body, b2Fixture* fixture)
{
	if (!m_contactListener) return;
	(void)GetContactList();
	for (size_t i = 0; i < m_contactViews.size(); ++i)
	{
		b2Contact* c = &m_contactViews[i];
		if (!c->m_touching) continue;
		const bool mine = fixture ? (c->m_fixtureA == fixture || c->m_fixtureB == fixture)
		                          : (c->m_fixtureA->GetBody() == body || c->m_fixtureB->GetBody() == body);
		if (!mine) continue;
		// (the step's own event list will name this contact's end as well: remembered, so that it is delivered once)
		m_endedEarly.push_back(std::make_pair(c->m_fixtureA->m_id + c->m_indexA, c->m_fixtureB->m_id + c->m_indexB));
		if (m_contactListener->EndContactImmediate(c, 0)) m_contactListener->EndContact(c);
	}
}

void b2World::DestroyBody(b2Body* b)
{
	if (IsLocked() || !m_hip || b == nullptr) return;
	// nothing is torn down and no listener hears anything for a body the device does not know (any more)
	if (b->m_world != this || b->m_id < 0 || b->m_id >= (int32)m_bodies.size() || m_bodies[b->m_id] != b || b2hip_body_is_destroyed(m_hip, b->m_id)) return;
	// the reference's order (b2World.cpp:594-643): the attached joints go first (SayGoodbye, then destroyed), then the
	// contacts end, then the fixtures say goodbye and go with the body
	for (b2Joint* j = m_jointList; j != nullptr;)
	{
		b2Joint* next = j->m_next;
		if (j->m_bodyA == b || j->m_bodyB == b)
		{
			if (m_destructionListener) m_destructionListener->SayGoodbye(j);
			DestroyJoint(j);
		}
		j = next;
	}
	EndContactsOf(b, nullptr);
	for (b2Fixture* f = b->m_fixtureList; f != nullptr; f = f->m_next)
	{
		if (m_destructionListener) m_destructionListener->SayGoodbye(f);
	}
	if (b2hip_destroy_body(m_hip, b->m_id) != B2HIP_OK)
	{
		fprintf(stderr, "b2World::DestroyBody: %s\n", b2hip_last_error());
		return;
	}
	for (b2Fixture* f = b->m_fixtureList; f != nullptr;)
	{
		b2Fixture* next = f->m_next;
		DestroyFixtureView(f);
		f = next;
	}
	b->m_fixtureList = nullptr;
	if (b->m_prev) b->m_prev->m_next = b->m_next;
	if (b->m_next) b->m_next->m_prev = b->m_prev;
	if (b == m_bodyList) m_bodyList = b->m_next;
	--m_bodyCount;
	m_bodies[b->m_id] = nullptr;
	b->~b2Body();
	b2Free(b);
	m_statesValid = false;
	m_contactsValid = false;
	m_fatValid = false;
}

b2Joint* b2World::CreateJoint(const b2JointDef* def)
{
	if (IsLocked() || !m_hip) return nullptr;
	int id = -1;
	b2Joint* j = nullptr;
	if (def->type == e_revoluteJoint)
	{
		const b2RevoluteJointDef* rd = static_cast<const b2RevoluteJointDef*>(def);
		b2hip_revolute_joint_def d;
		d.body_a = rd->bodyA->GetDeviceId();
		d.body_b = rd->bodyB->GetDeviceId();
		d.local_anchor_a[0] = rd->localAnchorA.x;
		d.local_anchor_a[1] = rd->localAnchorA.y;
		d.local_anchor_b[0] = rd->localAnchorB.x;
		d.local_anchor_b[1] = rd->localAnchorB.y;
		d.reference_angle = rd->referenceAngle;
		d.enable_limit = rd->enableLimit;
		d.lower_angle = rd->lowerAngle;
		d.upper_angle = rd->upperAngle;
		d.enable_motor = rd->enableMotor;
		d.motor_speed = rd->motorSpeed;
		d.max_motor_torque = rd->maxMotorTorque;
		d.collide_connected = rd->collideConnected;
		id = b2hip_create_revolute_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2RevoluteJoint))) b2RevoluteJoint(rd);
	}
	else if (def->type == e_distanceJoint)
	{
		const b2DistanceJointDef* dd = static_cast<const b2DistanceJointDef*>(def);
		b2hip_distance_joint_def d;
		d.body_a = dd->bodyA->GetDeviceId();
		d.body_b = dd->bodyB->GetDeviceId();
		d.local_anchor_a[0] = dd->localAnchorA.x;
		d.local_anchor_a[1] = dd->localAnchorA.y;
		d.local_anchor_b[0] = dd->localAnchorB.x;
		d.local_anchor_b[1] = dd->localAnchorB.y;
		d.length = dd->length;
		d.frequency_hz = dd->frequencyHz;
		d.damping_ratio = dd->dampingRatio;
		d.collide_connected = dd->collideConnected;
		id = b2hip_create_distance_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2DistanceJoint))) b2DistanceJoint(dd);
	}
	else if (def->type == e_prismaticJoint)
	{
		const b2PrismaticJointDef* pd = static_cast<const b2PrismaticJointDef*>(def);
		b2hip_prismatic_joint_def d;
		d.body_a = pd->bodyA->GetDeviceId();
		d.body_b = pd->bodyB->GetDeviceId();
		d.local_anchor_a[0] = pd->localAnchorA.x;
		d.local_anchor_a[1] = pd->localAnchorA.y;
		d.local_anchor_b[0] = pd->localAnchorB.x;
		d.local_anchor_b[1] = pd->localAnchorB.y;
		d.local_axis_a[0] = pd->localAxisA.x;
		d.local_axis_a[1] = pd->localAxisA.y;
		d.reference_angle = pd->referenceAngle;
		d.enable_limit = pd->enableLimit;
		d.lower_translation = pd->lowerTranslation;
		d.upper_translation = pd->upperTranslation;
		d.enable_motor = pd->enableMotor;
		d.motor_speed = pd->motorSpeed;
		d.max_motor_force = pd->maxMotorForce;
		d.collide_connected = pd->collideConnected;
		id = b2hip_create_prismatic_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2PrismaticJoint))) b2PrismaticJoint(pd);
	}
	else if (def->type == e_weldJoint)
	{
		const b2WeldJointDef* wd = static_cast<const b2WeldJointDef*>(def);
		b2hip_weld_joint_def d;
		d.body_a = wd->bodyA->GetDeviceId();
		d.body_b = wd->bodyB->GetDeviceId();
		d.local_anchor_a[0] = wd->localAnchorA.x;
		d.local_anchor_a[1] = wd->localAnchorA.y;
		d.local_anchor_b[0] = wd->localAnchorB.x;
		d.local_anchor_b[1] = wd->localAnchorB.y;
		d.reference_angle = wd->referenceAngle;
		d.frequency_hz = wd->frequencyHz;
		d.damping_ratio = wd->dampingRatio;
		d.collide_connected = wd->collideConnected;
		id = b2hip_create_weld_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2WeldJoint))) b2WeldJoint(wd);
	}
	else if (def->type == e_wheelJoint)
	{
		const b2WheelJointDef* wd = static_cast<const b2WheelJointDef*>(def);
		b2hip_wheel_joint_def d;
		d.body_a = wd->bodyA->GetDeviceId();
		d.body_b = wd->bodyB->GetDeviceId();
		d.local_anchor_a[0] = wd->localAnchorA.x;
		d.local_anchor_a[1] = wd->localAnchorA.y;
		d.local_anchor_b[0] = wd->localAnchorB.x;
		d.local_anchor_b[1] = wd->localAnchorB.y;
		d.local_axis_a[0] = wd->localAxisA.x;
		d.local_axis_a[1] = wd->localAxisA.y;
		d.frequency_hz = wd->frequencyHz;
		d.damping_ratio = wd->dampingRatio;
		d.enable_motor = wd->enableMotor;
		d.motor_speed = wd->motorSpeed;
		d.max_motor_torque = wd->maxMotorTorque;
		d.collide_connected = wd->collideConnected;
		id = b2hip_create_wheel_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2WheelJoint))) b2WheelJoint(wd);
	}
	else if (def->type == e_ropeJoint)
	{
		const b2RopeJointDef* rd = static_cast<const b2RopeJointDef*>(def);
		b2hip_rope_joint_def d;
		d.body_a = rd->bodyA->GetDeviceId();
		d.body_b = rd->bodyB->GetDeviceId();
		d.local_anchor_a[0] = rd->localAnchorA.x;
		d.local_anchor_a[1] = rd->localAnchorA.y;
		d.local_anchor_b[0] = rd->localAnchorB.x;
		d.local_anchor_b[1] = rd->localAnchorB.y;
		d.max_length = rd->maxLength;
		d.collide_connected = rd->collideConnected;
		id = b2hip_create_rope_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2RopeJoint))) b2RopeJoint(rd);
	}
	else if (def->type == e_frictionJoint)
	{
		const b2FrictionJointDef* fd = static_cast<const b2FrictionJointDef*>(def);
		b2hip_friction_joint_def d;
		d.body_a = fd->bodyA->GetDeviceId();
		d.body_b = fd->bodyB->GetDeviceId();
		d.local_anchor_a[0] = fd->localAnchorA.x;
		d.local_anchor_a[1] = fd->localAnchorA.y;
		d.local_anchor_b[0] = fd->localAnchorB.x;
		d.local_anchor_b[1] = fd->localAnchorB.y;
		d.max_force = fd->maxForce;
		d.max_torque = fd->maxTorque;
		d.collide_connected = fd->collideConnected;
		id = b2hip_create_friction_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2FrictionJoint))) b2FrictionJoint(fd);
	}
	else if (def->type == e_motorJoint)
	{
		const b2MotorJointDef* md = static_cast<const b2MotorJointDef*>(def);
		b2hip_motor_joint_def d;
		d.body_a = md->bodyA->GetDeviceId();
		d.body_b = md->bodyB->GetDeviceId();
		d.linear_offset[0] = md->linearOffset.x;
		d.linear_offset[1] = md->linearOffset.y;
		d.angular_offset = md->angularOffset;
		d.max_force = md->maxForce;
		d.max_torque = md->maxTorque;
		d.correction_factor = md->correctionFactor;
		d.collide_connected = md->collideConnected;
		id = b2hip_create_motor_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2MotorJoint))) b2MotorJoint(md);
	}
	else if (def->type == e_pulleyJoint)
	{
		const b2PulleyJointDef* pd = static_cast<const b2PulleyJointDef*>(def);
		b2hip_pulley_joint_def d;
		d.body_a = pd->bodyA->GetDeviceId();
		d.body_b = pd->bodyB->GetDeviceId();
		d.ground_anchor_a[0] = pd->groundAnchorA.x;
		d.ground_anchor_a[1] = pd->groundAnchorA.y;
		d.ground_anchor_b[0] = pd->groundAnchorB.x;
		d.ground_anchor_b[1] = pd->groundAnchorB.y;
		d.local_anchor_a[0] = pd->localAnchorA.x;
		d.local_anchor_a[1] = pd->localAnchorA.y;
		d.local_anchor_b[0] = pd->localAnchorB.x;
		d.local_anchor_b[1] = pd->localAnchorB.y;
		d.length_a = pd->lengthA;
		d.length_b = pd->lengthB;
		d.ratio = pd->ratio;
		d.collide_connected = pd->collideConnected;
		id = b2hip_create_pulley_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2PulleyJoint))) b2PulleyJoint(pd);
	}
	else if (def->type == e_mouseJoint)
	{
		const b2MouseJointDef* md = static_cast<const b2MouseJointDef*>(def);
		b2hip_mouse_joint_def d;
		d.body_a = md->bodyA->GetDeviceId();
		d.body_b = md->bodyB->GetDeviceId();
		d.target[0] = md->target.x;
		d.target[1] = md->target.y;
		d.max_force = md->maxForce;
		d.frequency_hz = md->frequencyHz;
		d.damping_ratio = md->dampingRatio;
		d.collide_connected = md->collideConnected;
		id = b2hip_create_mouse_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2MouseJoint))) b2MouseJoint(md);
	}
	else if (def->type == e_gearJoint)
	{
		const b2GearJointDef* gd = static_cast<const b2GearJointDef*>(def);
		b2hip_gear_joint_def d;
		d.joint1 = gd->joint1->GetDeviceId();
		d.joint2 = gd->joint2->GetDeviceId();
		d.ratio = gd->ratio;
		d.collide_connected = gd->collideConnected;
		id = b2hip_create_gear_joint(m_hip, &d);
		if (id >= 0) j = new (b2Alloc(sizeof(b2GearJoint))) b2GearJoint(gd);
	}
	else
	{
		fprintf(stderr, "b2World::CreateJoint: unknown joint type %d\n", (int)def->type);
		return nullptr;
	}
	if (id < 0)
	{
		fprintf(stderr, "b2World::CreateJoint: %s\n", b2hip_last_error());
		return nullptr;
	}
	j->m_id = id;
	switch (def->type)
	{
	case e_revoluteJoint: j->m_anchorA = static_cast<const b2RevoluteJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2RevoluteJointDef*>(def)->localAnchorB; break;
	case e_prismaticJoint: j->m_anchorA = static_cast<const b2PrismaticJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2PrismaticJointDef*>(def)->localAnchorB; break;
	case e_distanceJoint: j->m_anchorA = static_cast<const b2DistanceJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2DistanceJointDef*>(def)->localAnchorB; break;
	case e_pulleyJoint: j->m_anchorA = static_cast<const b2PulleyJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2PulleyJointDef*>(def)->localAnchorB; break;
	case e_wheelJoint: j->m_anchorA = static_cast<const b2WheelJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2WheelJointDef*>(def)->localAnchorB; break;
	case e_weldJoint: j->m_anchorA = static_cast<const b2WeldJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2WeldJointDef*>(def)->localAnchorB; break;
	case e_frictionJoint: j->m_anchorA = static_cast<const b2FrictionJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2FrictionJointDef*>(def)->localAnchorB; break;
	case e_ropeJoint: j->m_anchorA = static_cast<const b2RopeJointDef*>(def)->localAnchorA; j->m_anchorB = static_cast<const b2RopeJointDef*>(def)->localAnchorB; break;
	case e_mouseJoint:
		j->m_anchorKind = 1;
		j->m_anchorB = b2MulT(def->bodyB->GetTransform(), static_cast<const b2MouseJointDef*>(def)->target); // (b2MouseJoint.cpp:40)
		break;
	case e_motorJoint: j->m_anchorKind = 2; break;
	case e_gearJoint:
		j->m_anchorA = static_cast<const b2GearJointDef*>(def)->joint1->m_anchorB; // (b2GearJoint.cpp:67,76 and :99,108)
		j->m_anchorB = static_cast<const b2GearJointDef*>(def)->joint2->m_anchorB;
		break;
	default: break;
	}
	// the bodies' joint lists (b2World.cpp:258-270); a gear's bodies are the second bodies of its two joints
	j->m_edgeA.joint = j;
	j->m_edgeA.other = j->m_bodyB;
	j->m_edgeA.prev = nullptr;
	j->m_edgeA.next = j->m_bodyA->m_jointList;
	if (j->m_bodyA->m_jointList) j->m_bodyA->m_jointList->prev = &j->m_edgeA;
	j->m_bodyA->m_jointList = &j->m_edgeA;
	j->m_edgeB.joint = j;
	j->m_edgeB.other = j->m_bodyA;
	j->m_edgeB.prev = nullptr;
	j->m_edgeB.next = j->m_bodyB->m_jointList;
	if (j->m_bodyB->m_jointList) j->m_bodyB->m_jointList->prev = &j->m_edgeB;
	j->m_bodyB->m_jointList = &j->m_edgeB;
	j->m_prev = nullptr;
	j->m_next = m_jointList;
	if (m_jointList) m_jointList->m_prev = j;
	m_jointList = j;
	++m_jointCount;
	return j;
}

void b2World::ShiftOrigin(const b2Vec2& newOrigin)
{
	if (IsLocked() || !m_hip) return;
	if (b2hip_shift_origin(m_hip, newOrigin.x, newOrigin.y) != B2HIP_OK)
	{
		fprintf(stderr, "b2World::ShiftOrigin: %s\n", b2hip_last_error());
		return;
	}
	for (b2Joint* j = m_jointList; j; j = j->m_next)
	{
		if (j->m_type == e_mouseJoint) static_cast<b2MouseJoint*>(j)->m_targetA -= newOrigin;
		else if (j->m_type == e_pulleyJoint)
		{
			static_cast<b2PulleyJoint*>(j)->m_groundAnchorA -= newOrigin;
			static_cast<b2PulleyJoint*>(j)->m_groundAnchorB -= newOrigin;
		}
	}
	m_statesValid = false;
	m_contactsValid = false;
	m_fatValid = false;
}

b2Vec2 b2Joint::GetAnchorA() const
{
	if (m_anchorKind == 1) return static_cast<const b2MouseJoint*>(this)->GetTarget();
	if (m_anchorKind == 2) return m_bodyA->GetPosition();
	return m_bodyA->GetWorldPoint(m_anchorA);
}

b2Vec2 b2Joint::GetAnchorB() const
{
	if (m_anchorKind == 2) return m_bodyB->GetPosition();
	return m_bodyB->GetWorldPoint(m_anchorB);
}

void b2World::DestroyJoint(b2Joint* j)
{
	if (IsLocked() || !m_hip || j == nullptr) return;
	if (b2hip_destroy_joint(m_hip, j->m_id) != B2HIP_OK)
	{
		fprintf(stderr, "b2World::DestroyJoint: %s\n", b2hip_last_error());
		return;
	}
	if (j->m_prev) j->m_prev->m_next = j->m_next;
	if (j->m_next) j->m_next->m_prev = j->m_prev;
	if (j == m_jointList) m_jointList = j->m_next;
	// (b2World.cpp:304-333)
	if (j->m_edgeA.prev) j->m_edgeA.prev->next = j->m_edgeA.next;
	if (j->m_edgeA.next) j->m_edgeA.next->prev = j->m_edgeA.prev;
	if (&j->m_edgeA == j->m_bodyA->m_jointList) j->m_bodyA->m_jointList = j->m_edgeA.next;
	if (j->m_edgeB.prev) j->m_edgeB.prev->next = j->m_edgeB.next;
	if (j->m_edgeB.next) j->m_edgeB.next->prev = j->m_edgeB.prev;
	if (&j->m_edgeB == j->m_bodyB->m_jointList) j->m_bodyB->m_jointList = j->m_edgeB.next;
	--m_jointCount;
	j->~b2Joint();
	b2Free(j);
}

void b2World::Step(float32 dt, int32 velocityIterations, int32 positionIterations, b2TaskExecutor& executor)
{
	// The physics phases are kernels; what is left on the host - the listener's *Immediate callbacks and the user's contact
	// filter - runs on this executor's worker threads, as the reference runs them from its collide / solve tasks.
	m_stepExecutor = &executor;
	if (!m_hip)
	{
		fprintf(stderr, "b2World::Step: no HIP world (no device): nothing can be stepped\n");
		return;
	}
	m_locked = true;
	int rc = b2hip_step(m_hip, dt, velocityIterations, positionIterations);
	m_locked = false;
	if (rc != B2HIP_OK)
	{
		fprintf(stderr, "b2World::Step: b2hip_step failed (%d): %s\n", rc, b2hip_last_error());
	}
	m_statesValid = false;
	m_contactsValid = false;
	m_fatValid = false;
	// the reference's order within a step: begin / end (Collide), [PreSolve: called by the step itself], PostSolve (Solve)
	DeliverContactEvents();
	DeliverPostSolve();
	DeliverToiCallbacks();
	m_stepExecutor = nullptr;
	float ms[13];
	if (b2hip_get_profile(m_hip, ms) == B2HIP_OK)
	{
		m_profile.step = ms[0];
		m_profile.collide = ms[1];
		m_profile.solve = ms[2];
		m_profile.solveTraversal = ms[3];
		m_profile.solveInit = ms[4];
		m_profile.solveVelocity = ms[5];
		m_profile.solvePosition = ms[6];
		m_profile.solveTOI = ms[7];
		m_profile.solveTOIFindMinContact = ms[8];
		m_profile.broadphase = ms[9];
		m_profile.broadphaseSyncFixtures = ms[10];
		m_profile.broadphaseFindContacts = ms[11];
		m_profile.locking = ms[12];
	}
}

void b2World::SetContactListener(b2ContactListener* listener)
{
	m_contactListener = listener;
	if (!m_hip) return;
	(void)b2hip_enable_contact_events(m_hip, listener != nullptr ? 1 : 0);
	(void)b2hip_set_pre_solve_batch(m_hip, listener != nullptr ? &b2World::PreSolveBatchTrampoline : nullptr, this);
	(void)b2hip_enable_post_solve(m_hip, listener != nullptr ? 1 : 0);
}

void b2World::SetContactFilter(b2ContactFilter* filter)
{
	m_contactFilter = filter;
	if (m_hip) (void)b2hip_set_contact_filter_batch(m_hip, filter != nullptr ? &b2World::FilterBatchTrampoline : nullptr, this);
}

// b2ContactManager::AddPair / Collide -> m_contactFilter->ShouldCollide (b2ContactManager.cpp:283-287, 195-203)
int b2World::FilterTrampoline(void* user, int fixtureA, int fixtureB)
{
	b2World* self = static_cast<b2World*>(user);
	if (!self->m_contactFilter) return 1;
	return self->m_contactFilter->ShouldCollide(self->m_fixtures[fixtureA], self->m_fixtures[fixtureB], 0) ? 1 : 0;
}

// ---- host-side callbacks on the executor ------------------------------------------------------------------------------------
namespace
{
struct CallbackRange : public b2RangeTask
{
	CallbackRange() : fn(nullptr), ctx(nullptr) {}
	void Execute(const b2ThreadContext& tc, const b2RangeTaskRange& range) override
	{
		for (uint32 i = range.begin; i < range.end; ++i) fn(ctx, i, tc.threadId);
	}
	void (*fn)(void*, uint32, uint32);
	void* ctx;
};
}

void b2World::RunOnExecutor(uint32 count, void (*fn)(void* ctx, uint32 index, uint32 threadId), void* ctx)
{
	if (count == 0) return;
	m_mtLocked = true; // (b2World::SetMtLock around the reference's collide / solve tasks)
	if (m_stepExecutor == nullptr || m_stepExecutor->GetThreadCount() < 2 || count < 2)
	{
		for (uint32 i = 0; i < count; ++i) fn(ctx, i, 0);
		m_mtLocked = false;
		return;
	}
	CallbackRange task;
	task.fn = fn;
	task.ctx = ctx;
	task.SetRange(b2RangeTaskRange(0, count));
	b2ExecuteRangeTask(*m_stepExecutor, task);
	m_mtLocked = false;
}

// b2ContactFilter::ShouldCollide(fixtureA, fixtureB, threadId) for every pair of one decision point, from the worker threads
// (b2WorldCallbacks.h:57-62: the reference asks from its find-contacts and collide tasks)
void b2World::FilterBatchTrampoline(void* user, int count, const int32_t* pairs, int32_t* verdict)
{
	b2World* self = static_cast<b2World*>(user);
	if (!self->m_contactFilter) return;
	struct Ctx { b2World* w; const int32_t* pairs; int32_t* verdict; } c = { self, pairs, verdict };
	self->RunOnExecutor((uint32)count, [](void* p, uint32 i, uint32 threadId)
	{
		Ctx* k = static_cast<Ctx*>(p);
		k->verdict[i] = k->w->m_contactFilter->ShouldCollide(k->w->m_fixtures[k->pairs[2 * i]], k->w->m_fixtures[k->pairs[2 * i + 1]], threadId) ? 1 : 0;
	}, &c);
}

// A device fixture id is (b2Fixture, child): a chain's children are consecutive device fixtures (b2Body::CreateFixture)
void b2World::BindFixtures(b2Contact& c, int deviceFixtureA, int deviceFixtureB) const
{
	c.m_fixtureA = m_fixtures[deviceFixtureA];
	c.m_fixtureB = m_fixtures[deviceFixtureB];
	c.m_indexA = c.m_fixtureA ? deviceFixtureA - c.m_fixtureA->m_id : 0;
	c.m_indexB = c.m_fixtureB ? deviceFixtureB - c.m_fixtureB->m_id : 0;
}

static void FillManifold(b2Manifold& out, const b2hip_manifold& m)
{
	memset(&out, 0, sizeof(out));
	out.type = (b2Manifold::Type)m.type;
	out.pointCount = m.point_count;
	out.localNormal.Set(m.local_normal[0], m.local_normal[1]);
	out.localPoint.Set(m.local_point[0], m.local_point[1]);
	for (int k = 0; k < 2; ++k)
	{
		out.points[k].localPoint.Set(m.point_local[k][0], m.point_local[k][1]);
		out.points[k].normalImpulse = m.normal_impulse[k];
		out.points[k].tangentImpulse = m.tangent_impulse[k];
		out.points[k].id.key = m.id_key[k];
	}
}

// b2Contact::Update -> PreSolveImmediate / PreSolve (b2Contact.cpp:283-297): called by the step between Collide and the
// island build; the contact view lives for the duration of the call.
int b2World::PreSolveTrampoline(void* user, int contactIndex, int fixtureA, int fixtureB, const b2hip_manifold* oldManifold,
	const b2hip_manifold* manifold, b2hip_contact_material* material)
{
	B2_NOT_USED(contactIndex);
	b2World* self = static_cast<b2World*>(user);
	if (!self->m_contactListener) return 1;
	b2Contact c;
	FillManifold(c.m_manifold, *manifold);
	self->BindFixtures(c, fixtureA, fixtureB);
	c.m_next = nullptr;
	// the contact's own values (b2Contact::m_friction ...: a listener may have overridden the mixture in an earlier step)
	c.m_friction = material->friction;
	c.m_restitution = material->restitution;
	c.m_tangentSpeed = material->tangent_speed;
	c.m_touching = true;
	c.m_enabled = true;
	b2Manifold old;
	FillManifold(old, *oldManifold);
	if (self->m_contactListener->PreSolveImmediate(&c, &old, 0)) self->m_contactListener->PreSolve(&c, &old);
	// SetFriction / SetRestitution / SetTangentSpeed / SetEnabled made on the view go back to the contact
	material->friction = c.m_friction;
	material->restitution = c.m_restitution;
	material->tangent_speed = c.m_tangentSpeed;
	return c.m_enabled ? 1 : 0;
}

// All PreSolve records of a step: PreSolveImmediate on the executor's threads (each contact view is a lane's own), then the
// deferred PreSolve of the contacts that asked for it, on this thread, in the reference's deferred order (the records' order).
void b2World::PreSolveBatchTrampoline(void* user, int count, b2hip_pre_solve_record* records)
{
	b2World* self = static_cast<b2World*>(user);
	if (!self->m_contactListener || count <= 0) return;
	struct Ctx
	{
		b2World* w;
		b2hip_pre_solve_record* records;
		std::vector<b2Contact> views;
		std::vector<b2Manifold> olds;
		std::vector<char> deferred;
	} c;
	c.w = self;
	c.records = records;
	c.views.resize((size_t)count);
	c.olds.resize((size_t)count);
	c.deferred.assign((size_t)count, 0);
	for (int i = 0; i < count; ++i)
	{
		b2Contact& v = c.views[i];
		FillManifold(v.m_manifold, records[i].manifold);
		FillManifold(c.olds[i], records[i].old_manifold);
		self->BindFixtures(v, records[i].fixture_a, records[i].fixture_b);
		v.m_next = nullptr;
		v.m_friction = records[i].material.friction;
		v.m_restitution = records[i].material.restitution;
		v.m_tangentSpeed = records[i].material.tangent_speed;
		v.m_touching = true;
		v.m_enabled = true;
	}
	self->RunOnExecutor((uint32)count, [](void* p, uint32 i, uint32 threadId)
	{
		Ctx* k = static_cast<Ctx*>(p);
		k->deferred[i] = k->w->m_contactListener->PreSolveImmediate(&k->views[i], &k->olds[i], threadId) ? 1 : 0;
	}, &c);
	for (int i = 0; i < count; ++i)
	{
		if (c.deferred[i]) self->m_contactListener->PreSolve(&c.views[i], &c.olds[i]);
		records[i].enabled = c.views[i].m_enabled ? 1 : 0;
		records[i].material.friction = c.views[i].m_friction;
		records[i].material.restitution = c.views[i].m_restitution;
		records[i].material.tangent_speed = c.views[i].m_tangentSpeed;
	}
}

// b2Island::Report -> PostSolveImmediate / PostSolve (b2Island.cpp:532-570, b2ContactManager.cpp:454-470)
void b2World::DeliverPostSolve()
{
	if (!m_hip || !m_contactListener) return;
	int count = b2hip_get_post_solve(m_hip, 0, nullptr);
	if (count <= 0) return;
	std::vector<b2hip_contact_impulse> rec(count);
	count = b2hip_get_post_solve(m_hip, count, rec.data());
	(void)GetContactList();
	const int n = (int)m_contactViews.size();
	// PostSolveImmediate on the executor's threads (the reference calls it from its solve tasks, b2Island.cpp:532-570), then the
	// deferred PostSolve of the contacts that asked for it, in the records' (proxy-key) order (b2ContactManager.cpp:454-470)
	struct Ctx
	{
		b2World* w;
		std::vector<b2Contact*> contacts;
		std::vector<b2ContactImpulse> impulses;
		std::vector<char> deferred;
	} c;
	c.w = this;
	for (int i = 0; i < count; ++i)
	{
		const b2hip_contact_impulse& r = rec[i];
		if (r.contact_index < 0 || r.contact_index >= n) continue;
		b2ContactImpulse impulse;
		impulse.count = r.count;
		for (int k = 0; k < b2_maxManifoldPoints; ++k)
		{
			impulse.normalImpulses[k] = k < r.count ? r.normal_impulses[k] : 0.0f;
			impulse.tangentImpulses[k] = k < r.count ? r.tangent_impulses[k] : 0.0f;
		}
		c.contacts.push_back(&m_contactViews[n - 1 - r.contact_index]);
		c.impulses.push_back(impulse);
	}
	c.deferred.assign(c.contacts.size(), 0);
	RunOnExecutor((uint32)c.contacts.size(), [](void* p, uint32 i, uint32 threadId)
	{
		Ctx* k = static_cast<Ctx*>(p);
		k->deferred[i] = k->w->m_contactListener->PostSolveImmediate(k->contacts[i], &k->impulses[i], threadId) ? 1 : 0;
	}, &c);
	for (size_t i = 0; i < c.contacts.size(); ++i)
	{
		if (c.deferred[i]) m_contactListener->PostSolve(c.contacts[i], &c.impulses[i]);
	}
}

// The listener calls the reference makes from inside its TOI sub-steps (b2World.cpp:866,946: contact->Update(listener);
// b2Island.cpp:527: Report), replayed from the step's log in the reference's call order, after the Collide and Solve callbacks
// like there. The sub-steps run on one thread in the reference: immediate and deferred form are called back to back, threadId 0.
// (PreSolve is not among them: the step itself calls it - through PreSolveBatchTrampoline, one record at a time - where the
// reference's sub-step does, because its answer changes that sub-step; include/b2hip.h, b2hip_toi_callback.)
void b2World::DeliverToiCallbacks()
{
	if (!m_hip || !m_contactListener) return;
	int count = b2hip_get_toi_callbacks(m_hip, 0, nullptr);
	if (count <= 0) return;
	std::vector<b2hip_toi_callback> rec((size_t)count);
	count = b2hip_get_toi_callbacks(m_hip, count, rec.data());
	for (int i = 0; i < count; ++i)
	{
		const b2hip_toi_callback& r = rec[i];
		if (r.fixture_a < 0 || r.fixture_b < 0 || r.fixture_a >= (int)m_fixtures.size() || r.fixture_b >= (int)m_fixtures.size()) continue;
		if (!m_fixtures[r.fixture_a] || !m_fixtures[r.fixture_b]) continue;
		b2Contact c;
		FillManifold(c.m_manifold, r.manifold);
		BindFixtures(c, r.fixture_a, r.fixture_b);
		c.m_next = nullptr;
		c.m_friction = r.material.friction;
		c.m_restitution = r.material.restitution;
		c.m_tangentSpeed = r.material.tangent_speed;
		c.m_enabled = true;
		if (r.kind & 8)
		{
			// (the record's impulses are the sub-step solver's, its point count the solver's; the view shows the contact's manifold)
			b2ContactImpulse impulse;
			impulse.count = r.manifold.point_count;
			for (int k = 0; k < b2_maxManifoldPoints; ++k)
			{
				impulse.normalImpulses[k] = k < impulse.count ? r.manifold.normal_impulse[k] : 0.0f;
				impulse.tangentImpulses[k] = k < impulse.count ? r.manifold.tangent_impulse[k] : 0.0f;
			}
			c.m_touching = true;
			if (m_contactListener->PostSolveImmediate(&c, &impulse, 0)) m_contactListener->PostSolve(&c, &impulse);
			continue;
		}
		c.m_touching = (r.kind & 2) == 0;
		if ((r.kind & 1) && m_contactListener->BeginContactImmediate(&c, 0)) m_contactListener->BeginContact(&c);
		if ((r.kind & 2) && m_contactListener->EndContactImmediate(&c, 0)) m_contactListener->EndContact(&c);
		if (r.kind & 4)
		{
			b2Manifold old;
			FillManifold(old, r.old_manifold);
			if (m_contactListener->PreSolveImmediate(&c, &old, 0)) m_contactListener->PreSolve(&c, &old);
		}
	}
}

// b2ContactManager::Collide's deferred callbacks (b2ContactManager.cpp:420-438), fed from the device's event list.
void b2World::DeliverContactEvents()
{
	if (!m_hip || !m_contactListener) return;
	int count = b2hip_get_contact_events(m_hip, 0, nullptr);
	if (count <= 0)
	{
		m_endedEarly.clear();
		return;
	}
	std::vector<b2hip_contact_event> ev(count);
	count = b2hip_get_contact_events(m_hip, count, ev.data());
	(void)GetContactList(); // views of this step's contacts, newest first
	const int n = (int)m_contactViews.size();
	// Begin / EndContactImmediate on the executor's threads (the reference calls them from its collide tasks,
	// b2Contact.cpp:253-281), then the deferred calls of the contacts that asked for them, in the list's order
	struct Ctx
	{
		b2World* w;
		std::vector<b2Contact*> contacts;
		std::vector<char> ended, deferred;
		std::vector<b2Contact> gone; // views of contacts that no longer exist (their end events)
	} c;
	c.w = this;
	c.gone.reserve((size_t)count);
	for (int i = 0; i < count; ++i)
	{
		const b2hip_contact_event& e = ev[i];
		// (the end of a contact that went with its body or fixture was delivered by DestroyBody / DestroyFixture itself)
		if (e.fixture_a >= (int)m_fixtures.size() || e.fixture_b >= (int)m_fixtures.size() || !m_fixtures[e.fixture_a] || !m_fixtures[e.fixture_b]) continue;
		if (e.kind != 0)
		{
			// ... and so was the end of a contact that went when its body was switched off or changed type
			bool early = false;
			for (size_t k = 0; k < m_endedEarly.size() && !early; ++k)
			{
				if ((m_endedEarly[k].first == e.fixture_a && m_endedEarly[k].second == e.fixture_b) ||
					(m_endedEarly[k].first == e.fixture_b && m_endedEarly[k].second == e.fixture_a))
				{
					m_endedEarly.erase(m_endedEarly.begin() + (long)k);
					early = true;
				}
			}
			if (early) continue;
		}
		b2Contact* view = nullptr;
		if (e.contact_index >= 0 && e.contact_index < n) view = &m_contactViews[n - 1 - e.contact_index];
		else
		{
			c.gone.push_back(b2Contact());
			b2Contact& g = c.gone.back();
			memset(&g.m_manifold, 0, sizeof(g.m_manifold));
			BindFixtures(g, e.fixture_a, e.fixture_b);
			g.m_next = nullptr;
			g.m_friction = 0.0f;
			g.m_restitution = 0.0f;
			g.m_touching = false;
			g.m_enabled = true;
			view = &g;
		}
		c.contacts.push_back(view);
		c.ended.push_back(e.kind != 0 ? 1 : 0);
	}
	c.deferred.assign(c.contacts.size(), 0);
	RunOnExecutor((uint32)c.contacts.size(), [](void* p, uint32 i, uint32 threadId)
	{
		Ctx* k = static_cast<Ctx*>(p);
		b2ContactListener* listener = k->w->m_contactListener;
		k->deferred[i] = (k->ended[i] ? listener->EndContactImmediate(k->contacts[i], threadId) : listener->BeginContactImmediate(k->contacts[i], threadId)) ? 1 : 0;
	}, &c);
	for (size_t i = 0; i < c.contacts.size(); ++i)
	{
		if (!c.deferred[i]) continue;
		if (c.ended[i]) m_contactListener->EndContact(c.contacts[i]); else m_contactListener->BeginContact(c.contacts[i]);
	}
	m_endedEarly.clear();
}

const std::vector<b2AABB>& b2World::FatAABBs()
{
	if (!m_fatValid)
	{
		const int n = (int)m_fixtures.size();
		m_fatAABBs.resize(n);
		std::vector<float> raw(4 * (size_t)(n > 0 ? n : 1));
		if (m_hip && n > 0 && b2hip_get_fat_aabbs(m_hip, 0, n, raw.data()) == B2HIP_OK)
		{
			for (int i = 0; i < n; ++i)
			{
				m_fatAABBs[i].lowerBound.Set(raw[4 * i + 0], raw[4 * i + 1]);
				m_fatAABBs[i].upperBound.Set(raw[4 * i + 2], raw[4 * i + 3]);
			}
		}
		m_fatValid = true;
		m_shadowValid = false;
	}
	return m_fatAABBs;
}

// The shadow tree: one leaf per live proxy (device fixture id of an active body), its box = the device's fat AABB. After a
// step the boxes are read back once (FatAABBs) and only the leaves whose box changed are re-inserted - the proxies that left
// their fat AABB, which is what the reference's tree re-inserts as well (b2DynamicTree::MoveProxy).
void b2World::SyncShadowTree()
{
	const std::vector<b2AABB>& fat = FatAABBs();
	if (m_shadowValid) return;
	const size_t n = fat.size();
	if (m_shadowLeaf.size() < n) m_shadowLeaf.resize(n, b2_nullNode);
	for (size_t i = 0; i < m_shadowLeaf.size(); ++i)
	{
		const bool live = i < n && m_fixtures[i] != nullptr && m_fixtures[i]->GetBody()->IsActive();
		int32& leaf = m_shadowLeaf[i];
		if (!live)
		{
			if (leaf != b2_nullNode)
			{
				m_shadowTree.RemoveFat(leaf);
				leaf = b2_nullNode;
			}
			continue;
		}
		if (leaf == b2_nullNode) leaf = m_shadowTree.InsertFat(fat[i], (void*)(intptr_t)i);
		else if (memcmp(&m_shadowTree.GetFatAABB(leaf), &fat[i], sizeof(b2AABB)) != 0) m_shadowTree.ReplaceFat(leaf, fat[i]);
	}
	m_shadowValid = true;
}

namespace
{
// b2WorldQueryWrapper / b2WorldRayCastWrapper of the reference (b2World.cpp:1740-1795): leaf -> fixture (+ child)
struct ShadowQuery
{
	bool QueryCallback(int32 leaf)
	{
		const size_t id = (size_t)(intptr_t)tree->GetUserData(leaf);
		return callback->ReportFixture((*fixtures)[id]);
	}
	const b2DynamicTree* tree;
	const std::vector<b2Fixture*>* fixtures;
	b2QueryCallback* callback;
};

struct ShadowRayCast
{
	float32 RayCastCallback(const b2RayCastInput& input, int32 leaf)
	{
		const size_t id = (size_t)(intptr_t)tree->GetUserData(leaf);
		b2Fixture* fixture = (*fixtures)[id];
		b2RayCastOutput output;
		if (!fixture->GetShape()->RayCast(&output, input, fixture->GetBody()->GetTransform(), (int32)id - fixture->GetDeviceId())) return input.maxFraction;
		const float32 fraction = output.fraction;
		const b2Vec2 point = (1.0f - fraction) * input.p1 + fraction * input.p2;
		return callback->ReportFixture(fixture, point, output.normal, fraction);
	}
	const b2DynamicTree* tree;
	const std::vector<b2Fixture*>* fixtures;
	b2RayCastCallback* callback;
};
}

void b2World::QueryAABB(b2QueryCallback* callback, const b2AABB& aabb)
{
	SyncShadowTree();
	ShadowQuery q = { &m_shadowTree, &m_fixtures, callback };
	m_shadowTree.Query(&q, aabb);
}

void b2World::RayCast(b2RayCastCallback* callback, const b2Vec2& point1, const b2Vec2& point2)
{
	SyncShadowTree();
	ShadowRayCast q = { &m_shadowTree, &m_fixtures, callback };
	b2RayCastInput input;
	input.p1 = point1;
	input.p2 = point2;
	input.maxFraction = 1.0f;
	m_shadowTree.RayCast(&q, input);
}

int32 b2World::GetTreeHeight() const
{
	const_cast<b2World*>(this)->SyncShadowTree();
	return m_shadowTree.GetHeight();
}

int32 b2World::GetTreeBalance() const
{
	const_cast<b2World*>(this)->SyncShadowTree();
	return m_shadowTree.GetMaxBalance();
}

float32 b2World::GetTreeQuality() const
{
	const_cast<b2World*>(this)->SyncShadowTree();
	return m_shadowTree.GetAreaRatio();
}

int32 b2BroadPhase::GetProxyCount() const { return m_world->GetProxyCount(); }
int32 b2BroadPhase::GetTreeHeight() const { return m_world->GetTreeHeight(); }
int32 b2BroadPhase::GetTreeBalance() const { return m_world->GetTreeBalance(); }
float32 b2BroadPhase::GetTreeQuality() const { return m_world->GetTreeQuality(); }
const b2AABB& b2BroadPhase::GetFatAABB(int32 proxyId) const
{
	const std::vector<b2AABB>& fat = m_world->FatAABBs();
	b2AABB& out = m_scratch[m_turn ^= 1];
	if (proxyId >= 0 && proxyId < (int32)fat.size()) out = fat[proxyId];
	else { out.lowerBound.SetZero(); out.upperBound.SetZero(); }
	return out;
}

void b2World::ClearForces()
{
	// forces are cleared on the device at the end of Step when auto-clear is on (default)
}

int32 b2World::GetProxyCount() const
{
	// proxies in the broad-phase = live fixtures (children counted) of active bodies (b2BroadPhase::GetProxyCount)
	int32 count = 0;
	for (size_t i = 0; i < m_fixtures.size(); ++i)
	{
		if (m_fixtures[i] != nullptr && m_fixtures[i]->GetBody()->IsActive()) ++count;
	}
	return count;
}

int32 b2World::GetContactCount() const
{
	return m_hip ? b2hip_contact_count(m_hip) : 0;
}

// The per-body getters and setters may be called from user range tasks on several threads at once, on different bodies
// (Testbed/Tests/ManyBodies.h:39-64): the first reader after a step fills the cache under a lock, a setter refreshes its
// own body's row only.
void b2World::RefreshStates() const
{
	if (m_statesValid.load(std::memory_order_acquire)) return;
	std::lock_guard<std::mutex> lock(m_statesMutex);
	if (m_statesValid.load(std::memory_order_relaxed)) return;
	m_states.resize(m_bodies.size());
	if (m_hip && !m_bodies.empty())
	{
		b2hip_get_body_states(m_hip, 0, (int)m_bodies.size(), m_states.data());
	}
	m_statesValid.store(true, std::memory_order_release);
}

void b2World::TouchState(int32 id) const
{
	if (!m_hip || !m_statesValid.load(std::memory_order_acquire) || (size_t)id >= m_states.size()) return;
	b2hip_get_body_states(m_hip, id, 1, &m_states[id]);
}

const b2hip_body_state& b2World::State(int32 id) const
{
	RefreshStates();
	return m_states[id];
}

b2Contact* b2World::GetContactList()
{
	if (!m_hip) return nullptr;
	if (!m_contactsValid)
	{
		int n = b2hip_contact_count(m_hip);
		std::vector<b2hip_contact> raw(n > 0 ? n : 1);
		n = n > 0 ? b2hip_get_contacts(m_hip, n, raw.data()) : 0;
		if (n < 0) n = 0;
		m_contactViews.assign(n, b2Contact());
		// the reference's list is newest first (b2ContactManager.cpp:715-724)
		for (int i = 0; i < n; ++i)
		{
			const b2hip_contact& r = raw[n - 1 - i];
			b2Contact& c = m_contactViews[i];
			memset(&c.m_manifold, 0, sizeof(c.m_manifold));
			c.m_manifold.type = (b2Manifold::Type)r.manifold_type;
			c.m_manifold.pointCount = r.point_count;
			c.m_manifold.localNormal.Set(r.local_normal[0], r.local_normal[1]);
			c.m_manifold.localPoint.Set(r.local_point[0], r.local_point[1]);
			for (int k = 0; k < 2; ++k)
			{
				c.m_manifold.points[k].localPoint.Set(r.point_local[k][0], r.point_local[k][1]);
				c.m_manifold.points[k].normalImpulse = r.normal_impulse[k];
				c.m_manifold.points[k].tangentImpulse = r.tangent_impulse[k];
				c.m_manifold.points[k].id.key = r.id_key[k];
			}
			BindFixtures(c, r.fixture_a, r.fixture_b);
			c.m_friction = r.friction;
			c.m_restitution = r.restitution;
			c.m_tangentSpeed = r.tangent_speed;
			c.m_touching = (r.flags & 1u) != 0;
			c.m_enabled = (r.flags & 2u) != 0;
			c.m_next = nullptr;
		}
		for (int i = 0; i + 1 < n; ++i) m_contactViews[i].m_next = &m_contactViews[i + 1];
		m_contactsValid = true;
	}
	return m_contactViews.empty() ? nullptr : &m_contactViews[0];
}

// ---- b2Body -------------------------------------------------------------------------------------
b2Body::b2Body(const b2BodyDef* bd, b2World* world, int32 id)
{
	m_world = world;
	m_id = id;
	m_type = bd->type;
	m_bullet = bd->bullet;
	m_allowSleep = bd->allowSleep;
	m_active = bd->active;
	m_fixedRotation = bd->fixedRotation;
	m_linearDamping = bd->linearDamping;
	m_angularDamping = bd->angularDamping;
	m_gravityScale = bd->gravityScale;
	m_prev = nullptr;
	m_next = nullptr;
	m_fixtureList = nullptr;
	m_jointList = nullptr;
	m_fixtureCount = 0;
	m_userData = bd->userData;
}

b2Body::~b2Body()
{
}

b2Fixture* b2Body::CreateFixture(const b2FixtureDef* def)
{
	if (m_world->IsLocked() || !m_world->m_hip) return nullptr;
	const b2Shape* shape = def->shape;
	b2hip_fixture_def fd;
	memset(&fd, 0, sizeof(fd));
	fd.density = def->density;
	fd.friction = def->friction;
	fd.restitution = def->restitution;
	fd.category_bits = def->filter.categoryBits;
	fd.mask_bits = def->filter.maskBits;
	fd.group_index = def->filter.groupIndex;
	fd.is_sensor = def->isSensor;
	fd.thick_shape = def->thickShape;
	// one device fixture (= one broad-phase proxy) per child, in child order like b2Fixture::CreateProxies
	// (b2Fixture.cpp:126-141): a chain of n segments is n consecutive records of type "chain child"
	const int32 children = shape->GetChildCount();
	if (children < 1) return nullptr;
	int id = -1;
	for (int32 child = 0; child < children; ++child)
	{
		b2hip_shape hs;
		b2ShapeToRecord(shape, child, &hs);
		const int got = b2hip_create_fixture(m_world->m_hip, m_id, &fd, &hs);
		if (got < 0 || (child > 0 && got != id + child))
		{
			fprintf(stderr, "b2Body::CreateFixture: %s\n", got < 0 ? b2hip_last_error() : "fixture ids of a chain are not consecutive");
			return nullptr;
		}
		if (child == 0) id = got;
	}
	void* mem = b2Alloc(sizeof(b2Fixture));
	b2Fixture* f = new (mem) b2Fixture;
	f->m_id = id;
	f->m_childCount = children;
	f->m_body = this;
	f->m_density = def->density;
	f->m_friction = def->friction;
	f->m_restitution = def->restitution;
	f->m_filter = def->filter;
	f->m_isSensor = def->isSensor;
	f->m_isThickShape = def->thickShape;
	f->m_userData = def->userData;
	f->m_shape = shape->Clone(&m_world->m_blockAllocator);
	f->m_next = m_fixtureList;
	m_fixtureList = f;
	++m_fixtureCount;
	if ((int)m_world->m_fixtures.size() < id + children) m_world->m_fixtures.resize(id + children, nullptr);
	for (int32 child = 0; child < children; ++child) m_world->m_fixtures[id + child] = f;
	m_world->m_fatValid = false;
	m_world->m_statesValid = false;
	return f;
}

b2Fixture* b2Body::CreateFixture(const b2Shape* shape, float32 density)
{
	b2FixtureDef def;
	def.shape = shape;
	def.density = density;
	return CreateFixture(&def);
}

const b2Transform& b2Body::GetTransform() const
{
	const b2hip_body_state& s = m_world->State(m_id);
	m_xfCache.p.Set(s.px, s.py);
	m_xfCache.q.Set(s.angle);
	return m_xfCache;
}

const b2Vec2& b2Body::GetPosition() const
{
	const b2hip_body_state& s = m_world->State(m_id);
	m_vecCache[0].Set(s.px, s.py);
	return m_vecCache[0];
}

float32 b2Body::GetAngle() const
{
	return m_world->State(m_id).angle;
}

const b2Vec2& b2Body::GetWorldCenter() const
{
	const b2hip_body_state& s = m_world->State(m_id);
	m_vecCache[1].Set(s.cx, s.cy);
	return m_vecCache[1];
}

const b2Vec2& b2Body::GetLocalCenter() const
{
	b2hip_mass_data md;
	b2hip_get_mass_data(m_world->m_hip, m_id, &md);
	m_vecCache[2].Set(md.local_center[0], md.local_center[1]);
	return m_vecCache[2];
}

const b2Vec2& b2Body::GetLinearVelocity() const
{
	const b2hip_body_state& s = m_world->State(m_id);
	m_vecCache[2].Set(s.vx, s.vy);
	return m_vecCache[2];
}

float32 b2Body::GetAngularVelocity() const
{
	return m_world->State(m_id).w;
}

bool b2Body::IsAwake() const
{
	return (m_world->State(m_id).flags & B2HIP_BODY_AWAKE) != 0;
}

void b2Body::SetLinearVelocity(const b2Vec2& v)
{
	if (m_type == b2_staticBody) return;
	const b2hip_body_state& s = m_world->State(m_id);
	b2hip_set_velocity(m_world->m_hip, m_id, v.x, v.y, s.w);
	m_world->TouchState(m_id);
}

void b2Body::SetAngularVelocity(float32 omega)
{
	if (m_type == b2_staticBody) return;
	const b2hip_body_state& s = m_world->State(m_id);
	b2hip_set_velocity(m_world->m_hip, m_id, s.vx, s.vy, omega);
	m_world->TouchState(m_id);
}

void b2Body::ApplyForce(const b2Vec2& force, const b2Vec2& point, bool wake)
{
	const b2hip_body_state& s = m_world->State(m_id);
	float32 torque = b2Cross(point - b2Vec2(s.cx, s.cy), force);
	b2hip_apply_force(m_world->m_hip, m_id, force.x, force.y, torque, wake);
	m_world->TouchState(m_id);
}

void b2Body::ApplyForceToCenter(const b2Vec2& force, bool wake)
{
	b2hip_apply_force(m_world->m_hip, m_id, force.x, force.y, 0.0f, wake);
	m_world->TouchState(m_id);
}

void b2Body::ApplyTorque(float32 torque, bool wake)
{
	b2hip_apply_force(m_world->m_hip, m_id, 0.0f, 0.0f, torque, wake);
	m_world->TouchState(m_id);
}

void b2Body::ApplyLinearImpulse(const b2Vec2& impulse, const b2Vec2& point, bool wake)
{
	b2hip_apply_linear_impulse(m_world->m_hip, m_id, impulse.x, impulse.y, point.x, point.y, wake);
	m_world->TouchState(m_id);
}

void b2Body::ApplyLinearImpulseToCenter(const b2Vec2& impulse, bool wake)
{
	b2hip_apply_linear_impulse_to_center(m_world->m_hip, m_id, impulse.x, impulse.y, wake);
	m_world->TouchState(m_id);
}

void b2Body::ApplyAngularImpulse(float32 impulse, bool wake)
{
	b2hip_apply_angular_impulse(m_world->m_hip, m_id, impulse, wake);
	m_world->TouchState(m_id);
}

void b2Body::SetTransform(const b2Vec2& position, float32 angle)
{
	if (m_world->IsLocked()) return;
	if (b2hip_set_transform(m_world->m_hip, m_id, position.x, position.y, angle) != B2HIP_OK)
		fprintf(stderr, "b2Body::SetTransform: %s\n", b2hip_last_error());
	m_world->m_statesValid = false;
	m_world->m_fatValid = false;
}

void b2Body::SetAwake(bool flag)
{
	b2hip_set_awake(m_world->m_hip, m_id, flag ? 1 : 0);
	m_world->TouchState(m_id);
}

void b2Body::SetBullet(bool flag)
{
	m_bullet = flag;
	b2hip_set_bullet(m_world->m_hip, m_id, flag ? 1 : 0);
	m_world->m_contactsValid = false;
}

// b2Body::SetActive (b2Body.cpp:496-544): the contacts of a body that is switched off end first (the listener hears it),
// as when the body is destroyed
void b2Body::SetActive(bool flag)
{
	if (m_world->IsLocked() || flag == m_active) return;
	if (!flag) m_world->EndContactsOf(this, nullptr);
	m_active = flag;
	if (b2hip_set_active(m_world->m_hip, m_id, flag ? 1 : 0) != B2HIP_OK) fprintf(stderr, "b2Body::SetActive: %s\n", b2hip_last_error());
	m_world->m_statesValid = false;
	m_world->m_contactsValid = false;
	m_world->m_fatValid = false;
}

// b2Body::SetType (b2Body.cpp:118-188)
void b2Body::SetType(b2BodyType type)
{
	if (m_world->IsLocked() || type == m_type) return;
	m_world->EndContactsOf(this, nullptr);
	m_type = type;
	if (b2hip_set_type(m_world->m_hip, m_id, (int)type) != B2HIP_OK) fprintf(stderr, "b2Body::SetType: %s\n", b2hip_last_error());
	m_world->m_statesValid = false;
	m_world->m_contactsValid = false;
	m_world->m_fatValid = false;
}

void b2Body::DestroyFixture(b2Fixture* fixture)
{
	if (fixture == nullptr || m_world->IsLocked() || fixture->m_body != this) return;
	// the reference's order (b2Body.cpp:254-290): out of the body's list first, then its contacts end, then it is destroyed
	b2Fixture** node = &m_fixtureList;
	while (*node != nullptr && *node != fixture) node = &(*node)->m_next;
	if (*node == fixture) *node = fixture->m_next;
	--m_fixtureCount;
	m_world->EndContactsOf(this, fixture);
	// (the proxies of a chain go in child order, b2Fixture::DestroyProxies b2Fixture.cpp:143-155)
	for (int32 child = 0; child < fixture->m_childCount; ++child)
	{
		if (b2hip_destroy_fixture(m_world->m_hip, fixture->m_id + child) != B2HIP_OK) fprintf(stderr, "b2Body::DestroyFixture: %s\n", b2hip_last_error());
	}
	m_world->DestroyFixtureView(fixture);
	m_world->m_statesValid = false;
	m_world->m_contactsValid = false;
	m_world->m_fatValid = false;
}

b2ContactEdge* b2Body::GetContactList()
{
	// edges of this body over the world's contact views (newest contact first, like the reference's per-body list)
	b2World* w = m_world;
	(void)w->GetContactList();
	std::vector<b2ContactEdge>& edges = w->m_edgeViews;
	edges.clear();
	for (size_t i = 0; i < w->m_contactViews.size(); ++i)
	{
		b2Contact& c = w->m_contactViews[i];
		b2Body* bA = c.GetFixtureA()->GetBody();
		b2Body* bB = c.GetFixtureB()->GetBody();
		if (bA != this && bB != this) continue;
		b2ContactEdge e;
		e.other = bA == this ? bB : bA;
		e.contact = &c;
		e.prev = nullptr;
		e.next = nullptr;
		edges.push_back(e);
	}
	for (size_t i = 0; i < edges.size(); ++i)
	{
		edges[i].prev = i > 0 ? &edges[i - 1] : nullptr;
		edges[i].next = i + 1 < edges.size() ? &edges[i + 1] : nullptr;
	}
	return edges.empty() ? nullptr : &edges[0];
}

void b2Fixture::SetFilterData(const b2Filter& filter)
{
	m_filter = filter;
	for (int32 child = 0; child < m_childCount; ++child) b2hip_fixture_set_filter(m_body->GetWorld()->GetDeviceWorld(), m_id + child, filter.categoryBits, filter.maskBits, filter.groupIndex);
}

void b2Fixture::Refilter()
{
	if (m_body == nullptr) return;
	for (int32 child = 0; child < m_childCount; ++child) b2hip_fixture_refilter(m_body->GetWorld()->GetDeviceWorld(), m_id + child);
}

void b2Fixture::SetSensor(bool sensor)
{
	if (sensor == m_isSensor) return;
	m_isSensor = sensor;
	for (int32 child = 0; child < m_childCount; ++child) b2hip_fixture_set_sensor(m_body->GetWorld()->GetDeviceWorld(), m_id + child, sensor ? 1 : 0);
	m_body->m_world->TouchState(m_body->m_id); // (SetSensor wakes its body)
}

void b2Fixture::SetDensity(float32 density)
{
	m_density = density;
	for (int32 child = 0; child < m_childCount; ++child) b2hip_fixture_set_material(m_body->GetWorld()->GetDeviceWorld(), m_id + child, m_density, m_friction, m_restitution);
}

void b2Fixture::SetFriction(float32 friction)
{
	m_friction = friction;
	for (int32 child = 0; child < m_childCount; ++child) b2hip_fixture_set_material(m_body->GetWorld()->GetDeviceWorld(), m_id + child, m_density, m_friction, m_restitution);
}

void b2Fixture::SetRestitution(float32 restitution)
{
	m_restitution = restitution;
	for (int32 child = 0; child < m_childCount; ++child) b2hip_fixture_set_material(m_body->GetWorld()->GetDeviceWorld(), m_id + child, m_density, m_friction, m_restitution);
}

bool b2Fixture::RayCast(b2RayCastOutput* output, const b2RayCastInput& input, int32 childIndex) const
{
	return m_shape->RayCast(output, input, m_body->GetTransform(), childIndex);
}

void b2Fixture::SetThickShape(bool flag)
{
	if (flag == m_isThickShape) return;
	m_isThickShape = flag;
	for (int32 child = 0; child < m_childCount; ++child) b2hip_fixture_set_thick(m_body->GetWorld()->GetDeviceWorld(), m_id + child, flag ? 1 : 0);
}

float32 b2Body::GetMass() const
{
	b2hip_mass_data md;
	b2hip_get_mass_data(m_world->m_hip, m_id, &md);
	return md.mass;
}

float32 b2Body::GetInertia() const
{
	b2hip_mass_data md;
	b2hip_get_mass_data(m_world->m_hip, m_id, &md);
	return md.inertia;
}

void b2Body::GetMassData(b2MassData* data) const
{
	b2hip_mass_data md;
	b2hip_get_mass_data(m_world->m_hip, m_id, &md);
	data->mass = md.mass;
	data->I = md.inertia;
	data->center.Set(md.local_center[0], md.local_center[1]);
}

void b2Body::SetMassData(const b2MassData* data)
{
	if (m_world->IsLocked() || data == nullptr) return;
	b2hip_mass_data md;
	memset(&md, 0, sizeof(md));
	md.mass = data->mass;
	md.inertia = data->I;
	md.local_center[0] = data->center.x;
	md.local_center[1] = data->center.y;
	b2hip_set_mass_data(m_world->m_hip, m_id, &md);
	m_world->TouchState(m_id);
}

void b2Body::ResetMassData()
{
	b2hip_set_mass_data(m_world->m_hip, m_id, nullptr);
	m_world->TouchState(m_id);
}

void b2Body::SetLinearDamping(float32 linearDamping)
{
	m_linearDamping = linearDamping;
	b2hip_set_body_damping(m_world->m_hip, m_id, m_linearDamping, m_angularDamping, m_gravityScale);
}

void b2Body::SetAngularDamping(float32 angularDamping)
{
	m_angularDamping = angularDamping;
	b2hip_set_body_damping(m_world->m_hip, m_id, m_linearDamping, m_angularDamping, m_gravityScale);
}

void b2Body::SetGravityScale(float32 scale)
{
	m_gravityScale = scale;
	b2hip_set_body_damping(m_world->m_hip, m_id, m_linearDamping, m_angularDamping, m_gravityScale);
}

void b2Body::SetFixedRotation(bool flag)
{
	if (m_fixedRotation == flag) return;
	m_fixedRotation = flag;
	b2hip_set_fixed_rotation(m_world->m_hip, m_id, flag ? 1 : 0);
	m_world->TouchState(m_id);
}

void b2Body::SetSleepingAllowed(bool flag)
{
	m_allowSleep = flag;
	b2hip_set_sleeping_allowed(m_world->m_hip, m_id, flag ? 1 : 0);
	m_world->TouchState(m_id);
}

// ---- b2Fixture ------------------------------------------------------------------------------------
bool b2Fixture::TestPoint(const b2Vec2& p) const
{
	return m_shape->TestPoint(m_body->GetTransform(), p);
}

const b2AABB& b2Fixture::GetAABB(int32 childIndex) const
{
	float a[4] = { 0, 0, 0, 0 };
	b2hip_get_fat_aabb(m_body->m_world->m_hip, m_id + (childIndex >= 0 && childIndex < m_childCount ? childIndex : 0), a);
	m_aabbCache.lowerBound.Set(a[0], a[1]);
	m_aabbCache.upperBound.Set(a[2], a[3]);
	return m_aabbCache;
}

// ---- b2Contact ------------------------------------------------------------------------------------
void b2Contact::GetWorldManifold(b2WorldManifold* worldManifold) const
{
	const b2Body* bodyA = m_fixtureA->GetBody();
	const b2Body* bodyB = m_fixtureB->GetBody();
	b2Transform xfA = bodyA->GetTransform();
	b2Transform xfB = bodyB->GetTransform();
	worldManifold->Initialize(&m_manifold, xfA, m_fixtureA->GetShape()->m_radius, xfB, m_fixtureB->GetShape()->m_radius);
}

// ---- joints ---------------------------------------------------------------------------------------
void b2RevoluteJointDef::Initialize(b2Body* bA, b2Body* bB, const b2Vec2& anchor)
{
	bodyA = bA;
	bodyB = bB;
	localAnchorA = bodyA->GetLocalPoint(anchor);
	localAnchorB = bodyB->GetLocalPoint(anchor);
	referenceAngle = bodyB->GetAngle() - bodyA->GetAngle();
}

void b2DistanceJointDef::Initialize(b2Body* bA, b2Body* bB, const b2Vec2& anchorA, const b2Vec2& anchorB)
{
	bodyA = bA;
	bodyB = bB;
	localAnchorA = bodyA->GetLocalPoint(anchorA);
	localAnchorB = bodyB->GetLocalPoint(anchorB);
	length = (anchorB - anchorA).Length();
}

void b2PrismaticJointDef::Initialize(b2Body* bA, b2Body* bB, const b2Vec2& anchor, const b2Vec2& axis)
{
	bodyA = bA;
	bodyB = bB;
	localAnchorA = bodyA->GetLocalPoint(anchor);
	localAnchorB = bodyB->GetLocalPoint(anchor);
	localAxisA = bodyA->GetLocalVector(axis);
	referenceAngle = bodyB->GetAngle() - bodyA->GetAngle();
}

void b2WeldJointDef::Initialize(b2Body* bA, b2Body* bB, const b2Vec2& anchor)
{
	bodyA = bA;
	bodyB = bB;
	localAnchorA = bodyA->GetLocalPoint(anchor);
	localAnchorB = bodyB->GetLocalPoint(anchor);
	referenceAngle = bodyB->GetAngle() - bodyA->GetAngle();
}

// Joint setters between steps: the definition lives in the device record (b2hip_joint_set_motor / _set_limits).
b2Vec2 b2Joint::GetReactionForce(float32 inv_dt) const
{
	float r[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
	b2hip_get_joint_reaction(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, inv_dt, r);
	return b2Vec2(r[0], r[1]);
}

float32 b2Joint::GetReactionTorque(float32 inv_dt) const
{
	float r[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
	b2hip_get_joint_reaction(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, inv_dt, r);
	return r[2];
}

float32 b2Joint::MotorReaction(float32 inv_dt) const
{
	float r[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
	b2hip_get_joint_reaction(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, inv_dt, r);
	return r[3];
}

bool b2Joint::IsActive() const { return m_bodyA->IsActive() && m_bodyB->IsActive(); }

float32 b2PulleyJoint::GetCurrentLengthA() const { return (m_bodyA->GetWorldPoint(m_localAnchorA) - m_groundAnchorA).Length(); }
float32 b2PulleyJoint::GetCurrentLengthB() const { return (m_bodyB->GetWorldPoint(m_localAnchorB) - m_groundAnchorB).Length(); }

void b2RevoluteJoint::PushMotor()
{
	b2hip_joint_set_motor(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_enableMotor, m_motorSpeed, m_maxMotorTorque);
}
void b2RevoluteJoint::PushLimits()
{
	b2hip_joint_set_limits(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_enableLimit, m_lowerAngle, m_upperAngle);
}
void b2RevoluteJoint::EnableMotor(bool flag) { m_enableMotor = flag; PushMotor(); }
void b2RevoluteJoint::SetMotorSpeed(float32 speed) { m_motorSpeed = speed; PushMotor(); }
void b2RevoluteJoint::SetMaxMotorTorque(float32 torque) { m_maxMotorTorque = torque; PushMotor(); }
void b2RevoluteJoint::EnableLimit(bool flag) { m_enableLimit = flag; PushLimits(); }
void b2RevoluteJoint::SetLimits(float32 lower, float32 upper) { m_lowerAngle = lower; m_upperAngle = upper; PushLimits(); }
float32 b2RevoluteJoint::GetJointAngle() const { return m_bodyB->GetAngle() - m_bodyA->GetAngle() - m_referenceAngle; }
float32 b2RevoluteJoint::GetJointSpeed() const { return m_bodyB->GetAngularVelocity() - m_bodyA->GetAngularVelocity(); }

void b2PrismaticJoint::PushMotor()
{
	b2hip_joint_set_motor(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_enableMotor, m_motorSpeed, m_maxMotorForce);
}
void b2PrismaticJoint::PushLimits()
{
	b2hip_joint_set_limits(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_enableLimit, m_lowerTranslation, m_upperTranslation);
}
void b2PrismaticJoint::EnableMotor(bool flag) { m_enableMotor = flag; PushMotor(); }
void b2PrismaticJoint::SetMotorSpeed(float32 speed) { m_motorSpeed = speed; PushMotor(); }
void b2PrismaticJoint::SetMaxMotorForce(float32 force) { m_maxMotorForce = force; PushMotor(); }
void b2PrismaticJoint::EnableLimit(bool flag) { m_enableLimit = flag; PushLimits(); }
void b2PrismaticJoint::SetLimits(float32 lower, float32 upper) { m_lowerTranslation = lower; m_upperTranslation = upper; PushLimits(); }

float32 b2PrismaticJoint::GetJointTranslation() const
{
	b2Vec2 d = m_bodyB->GetWorldPoint(m_localAnchorB) - m_bodyA->GetWorldPoint(m_localAnchorA);
	return b2Dot(d, m_bodyA->GetWorldVector(m_localAxisA));
}

float32 b2PrismaticJoint::GetJointSpeed() const
{
	const b2Rot qA = m_bodyA->GetTransform().q, qB = m_bodyB->GetTransform().q;
	b2Vec2 rA = b2Mul(qA, m_localAnchorA - m_bodyA->GetLocalCenter());
	b2Vec2 rB = b2Mul(qB, m_localAnchorB - m_bodyB->GetLocalCenter());
	b2Vec2 d = (m_bodyB->GetWorldCenter() + rB) - (m_bodyA->GetWorldCenter() + rA);
	b2Vec2 axis = b2Mul(qA, m_localAxisA);
	b2Vec2 vA = m_bodyA->GetLinearVelocity(), vB = m_bodyB->GetLinearVelocity();
	float32 wA = m_bodyA->GetAngularVelocity(), wB = m_bodyB->GetAngularVelocity();
	return b2Dot(d, b2Cross(wA, axis)) + b2Dot(axis, vB + b2Cross(wB, rB) - vA - b2Cross(wA, rA));
}

float32 b2WheelJoint::GetJointTranslation() const
{
	b2Vec2 d = m_bodyB->GetWorldPoint(m_localAnchorB) - m_bodyA->GetWorldPoint(m_localAnchorA);
	return b2Dot(d, m_bodyA->GetWorldVector(m_localAxisA));
}

float32 b2WheelJoint::GetJointLinearSpeed() const
{
	const b2Rot qA = m_bodyA->GetTransform().q, qB = m_bodyB->GetTransform().q;
	b2Vec2 rA = b2Mul(qA, m_localAnchorA - m_bodyA->GetLocalCenter());
	b2Vec2 rB = b2Mul(qB, m_localAnchorB - m_bodyB->GetLocalCenter());
	b2Vec2 d = (m_bodyB->GetWorldCenter() + rB) - (m_bodyA->GetWorldCenter() + rA);
	b2Vec2 axis = b2Mul(qA, m_localAxisA);
	b2Vec2 vA = m_bodyA->GetLinearVelocity(), vB = m_bodyB->GetLinearVelocity();
	float32 wA = m_bodyA->GetAngularVelocity(), wB = m_bodyB->GetAngularVelocity();
	return b2Dot(d, b2Cross(wA, axis)) + b2Dot(axis, vB + b2Cross(wB, rB) - vA - b2Cross(wA, rA));
}

float32 b2WheelJoint::GetJointAngle() const { return m_bodyB->GetAngle() - m_bodyA->GetAngle(); }
float32 b2WheelJoint::GetJointAngularSpeed() const { return m_bodyB->GetAngularVelocity() - m_bodyA->GetAngularVelocity(); }

void b2WheelJointDef::Initialize(b2Body* bA, b2Body* bB, const b2Vec2& anchor, const b2Vec2& axis)
{
	bodyA = bA;
	bodyB = bB;
	localAnchorA = bodyA->GetLocalPoint(anchor);
	localAnchorB = bodyB->GetLocalPoint(anchor);
	localAxisA = bodyA->GetLocalVector(axis);
}

void b2WheelJoint::PushMotor()
{
	b2hip_joint_set_motor(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_enableMotor, m_motorSpeed, m_maxMotorTorque);
}
void b2WheelJoint::EnableMotor(bool flag) { m_enableMotor = flag; PushMotor(); }
void b2WheelJoint::SetMotorSpeed(float32 speed) { m_motorSpeed = speed; PushMotor(); }
void b2WheelJoint::SetMaxMotorTorque(float32 torque) { m_maxMotorTorque = torque; PushMotor(); }
// ---- the scalar setters of the other joint classes: plain assignments in the reference, one C-ABI call each ------------------
#define B2_JOINT_DEVICE (m_bodyA->GetWorld()->GetDeviceWorld())
void b2DistanceJoint::SetLength(float32 length) { m_length = length; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_LENGTH, length); }
void b2DistanceJoint::SetFrequency(float32 hz) { m_frequencyHz = hz; b2hip_joint_set_spring(B2_JOINT_DEVICE, m_id, m_frequencyHz, m_dampingRatio); }
void b2DistanceJoint::SetDampingRatio(float32 ratio) { m_dampingRatio = ratio; b2hip_joint_set_spring(B2_JOINT_DEVICE, m_id, m_frequencyHz, m_dampingRatio); }
void b2FrictionJoint::SetMaxForce(float32 force) { m_maxForce = force; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_MAX_FORCE, force); }
void b2FrictionJoint::SetMaxTorque(float32 torque) { m_maxTorque = torque; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_MAX_TORQUE, torque); }
void b2GearJoint::SetRatio(float32 ratio) { m_ratio = ratio; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_RATIO, ratio); }
void b2MotorJoint::SetMaxForce(float32 force) { m_maxForce = force; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_MAX_FORCE, force); }
void b2MotorJoint::SetMaxTorque(float32 torque) { m_maxTorque = torque; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_MAX_TORQUE, torque); }
void b2MotorJoint::SetCorrectionFactor(float32 factor) { m_correctionFactor = factor; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_CORRECTION_FACTOR, factor); }
void b2MouseJoint::SetMaxForce(float32 force) { m_maxForce = force; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_MAX_FORCE, force); }
void b2MouseJoint::SetFrequency(float32 hz) { m_frequencyHz = hz; b2hip_joint_set_spring(B2_JOINT_DEVICE, m_id, m_frequencyHz, m_dampingRatio); }
void b2MouseJoint::SetDampingRatio(float32 ratio) { m_dampingRatio = ratio; b2hip_joint_set_spring(B2_JOINT_DEVICE, m_id, m_frequencyHz, m_dampingRatio); }
void b2RopeJoint::SetMaxLength(float32 length) { m_maxLength = length; b2hip_joint_set_param(B2_JOINT_DEVICE, m_id, B2HIP_JOINT_LENGTH, length); }
b2LimitState b2RopeJoint::GetLimitState() const
{
	const int state = b2hip_get_joint_limit_state(B2_JOINT_DEVICE, m_id);
	return state >= 0 ? (b2LimitState)state : e_inactiveLimit;
}
void b2WeldJoint::SetFrequency(float32 hz) { m_frequencyHz = hz; b2hip_joint_set_spring(B2_JOINT_DEVICE, m_id, m_frequencyHz, m_dampingRatio); }
void b2WeldJoint::SetDampingRatio(float32 ratio) { m_dampingRatio = ratio; b2hip_joint_set_spring(B2_JOINT_DEVICE, m_id, m_frequencyHz, m_dampingRatio); }
#undef B2_JOINT_DEVICE

void b2WheelJoint::SetSpringFrequencyHz(float32 hz)
{
	m_frequencyHz = hz;
	b2hip_joint_set_spring(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_frequencyHz, m_dampingRatio);
}
void b2WheelJoint::SetSpringDampingRatio(float32 ratio)
{
	m_dampingRatio = ratio;
	b2hip_joint_set_spring(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_frequencyHz, m_dampingRatio);
}

void b2FrictionJointDef::Initialize(b2Body* bA, b2Body* bB, const b2Vec2& anchor)
{
	bodyA = bA;
	bodyB = bB;
	localAnchorA = bodyA->GetLocalPoint(anchor);
	localAnchorB = bodyB->GetLocalPoint(anchor);
}

void b2MotorJointDef::Initialize(b2Body* bA, b2Body* bB)
{
	bodyA = bA;
	bodyB = bB;
	linearOffset = bodyA->GetLocalPoint(bodyB->GetPosition());
	angularOffset = bodyB->GetAngle() - bodyA->GetAngle();
}

void b2MotorJoint::SetLinearOffset(const b2Vec2& linearOffset)
{
	m_linearOffset = linearOffset;
	b2hip_joint_set_offsets(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_linearOffset.x, m_linearOffset.y, m_angularOffset);
}

void b2MotorJoint::SetAngularOffset(float32 angularOffset)
{
	m_angularOffset = angularOffset;
	b2hip_joint_set_offsets(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, m_linearOffset.x, m_linearOffset.y, m_angularOffset);
}

void b2PulleyJointDef::Initialize(b2Body* bA, b2Body* bB, const b2Vec2& groundA, const b2Vec2& groundB, const b2Vec2& anchorA,
	const b2Vec2& anchorB, float32 r)
{
	bodyA = bA;
	bodyB = bB;
	groundAnchorA = groundA;
	groundAnchorB = groundB;
	localAnchorA = bodyA->GetLocalPoint(anchorA);
	localAnchorB = bodyB->GetLocalPoint(anchorB);
	lengthA = (anchorA - groundA).Length();
	lengthB = (anchorB - groundB).Length();
	ratio = r;
}

void b2MouseJoint::SetTarget(const b2Vec2& target)
{
	m_targetA = target;
	b2hip_joint_set_target(m_bodyA->GetWorld()->GetDeviceWorld(), m_id, target.x, target.y);
}

// ---- callbacks / collision helpers ----------------------------------------------------------------
bool b2ContactFilter::ShouldCollide(b2Fixture* fixtureA, b2Fixture* fixtureB, uint32 threadId)
{
	B2_NOT_USED(threadId);
	const b2Filter& filterA = fixtureA->GetFilterData();
	const b2Filter& filterB = fixtureB->GetFilterData();
	if (filterA.groupIndex == filterB.groupIndex && filterA.groupIndex != 0)
	{
		return filterA.groupIndex > 0;
	}
	return (filterA.maskBits & filterB.categoryBits) != 0 && (filterA.categoryBits & filterB.maskBits) != 0;
}

void b2WorldManifold::Initialize(const b2Manifold* manifold, const b2Transform& xfA, float32 radiusA,
	const b2Transform& xfB, float32 radiusB)
{
	if (manifold->pointCount == 0) return;
	if (manifold->type == b2Manifold::e_circles)
	{
		normal.Set(1.0f, 0.0f);
		b2Vec2 pointA = b2Mul(xfA, manifold->localPoint);
		b2Vec2 pointB = b2Mul(xfB, manifold->points[0].localPoint);
		if (b2DistanceSquared(pointA, pointB) > b2_epsilon * b2_epsilon)
		{
			normal = pointB - pointA;
			normal.Normalize();
		}
		b2Vec2 cA = pointA + radiusA * normal;
		b2Vec2 cB = pointB - radiusB * normal;
		points[0] = 0.5f * (cA + cB);
		separations[0] = b2Dot(cB - cA, normal);
		return;
	}
	const bool faceA = manifold->type == b2Manifold::e_faceA;
	const b2Transform& xfRef = faceA ? xfA : xfB;
	const b2Transform& xfInc = faceA ? xfB : xfA;
	const float32 radiusRef = faceA ? radiusA : radiusB;
	const float32 radiusInc = faceA ? radiusB : radiusA;
	normal = b2Mul(xfRef.q, manifold->localNormal);
	b2Vec2 planePoint = b2Mul(xfRef, manifold->localPoint);
	for (int32 i = 0; i < manifold->pointCount; ++i)
	{
		b2Vec2 clipPoint = b2Mul(xfInc, manifold->points[i].localPoint);
		b2Vec2 cRef = clipPoint + (radiusRef - b2Dot(clipPoint - planePoint, normal)) * normal;
		b2Vec2 cInc = clipPoint - radiusInc * normal;
		points[i] = 0.5f * (cRef + cInc);
		separations[i] = b2Dot(cInc - cRef, normal);
	}
	if (!faceA) normal = -normal;
}

void b2GetPointStates(b2PointState state1[b2_maxManifoldPoints], b2PointState state2[b2_maxManifoldPoints],
	const b2Manifold* manifold1, const b2Manifold* manifold2)
{
	for (int32 i = 0; i < b2_maxManifoldPoints; ++i)
	{
		state1[i] = b2_nullState;
		state2[i] = b2_nullState;
	}
	for (int32 i = 0; i < manifold1->pointCount; ++i)
	{
		state1[i] = b2_removeState;
		for (int32 j = 0; j < manifold2->pointCount; ++j)
		{
			if (manifold2->points[j].id.key == manifold1->points[i].id.key) state1[i] = b2_persistState;
		}
	}
	for (int32 i = 0; i < manifold2->pointCount; ++i)
	{
		state2[i] = b2_addState;
		for (int32 j = 0; j < manifold1->pointCount; ++j)
		{
			if (manifold1->points[j].id.key == manifold2->points[i].id.key) state2[i] = b2_persistState;
		}
	}
}

// ---- debug drawing (reference: b2World.cpp:1797-2045: DrawShape, DrawJoint, DrawDebugData) ----------------------------------
// Behaviour restated: which primitive a shape type becomes, the colour of a body's state, what each b2Draw flag shows. Nothing
// here is part of the step: the host mirrors and the device's fat AABBs (the table QueryAABB reads) are all it looks at.
void b2World::DrawShape(const b2Fixture* fixture, const b2Transform& xf, const b2Color& color)
{
	const b2Shape* shape = fixture->GetShape();
	switch (shape->GetType())
	{
	case b2Shape::e_circle:
	{
		const b2CircleShape* circle = static_cast<const b2CircleShape*>(shape);
		m_debugDraw->DrawSolidCircle(b2Mul(xf, circle->m_p), circle->m_radius, b2Mul(xf.q, b2Vec2(1.0f, 0.0f)), color);
		break;
	}
	case b2Shape::e_edge:
	{
		const b2EdgeShape* edge = static_cast<const b2EdgeShape*>(shape);
		m_debugDraw->DrawSegment(b2Mul(xf, edge->m_vertex1), b2Mul(xf, edge->m_vertex2), color);
		break;
	}
	case b2Shape::e_chain:
	{
		// the links in the body's colour, a point on every vertex; ghost vertices (if any) dimmed, with a small circle
		const b2ChainShape* chain = static_cast<const b2ChainShape*>(shape);
		const b2Color ghost(0.75f * color.r, 0.75f * color.g, 0.75f * color.b, color.a);
		b2Vec2 from = b2Mul(xf, chain->m_vertices[0]);
		m_debugDraw->DrawPoint(from, 4.0f, color);
		if (chain->m_hasPrevVertex)
		{
			const b2Vec2 before = b2Mul(xf, chain->m_prevVertex);
			m_debugDraw->DrawSegment(before, from, ghost);
			m_debugDraw->DrawCircle(before, 0.1f, ghost);
		}
		for (int32 k = 1; k < chain->m_count; ++k)
		{
			const b2Vec2 to = b2Mul(xf, chain->m_vertices[k]);
			m_debugDraw->DrawSegment(from, to, color);
			m_debugDraw->DrawPoint(to, 4.0f, color);
			from = to;
		}
		if (chain->m_hasNextVertex)
		{
			const b2Vec2 after = b2Mul(xf, chain->m_nextVertex);
			m_debugDraw->DrawSegment(from, after, ghost);
			m_debugDraw->DrawCircle(after, 0.1f, ghost);
		}
		break;
	}
	case b2Shape::e_polygon:
	{
		const b2PolygonShape* poly = static_cast<const b2PolygonShape*>(shape);
		b2Vec2 world[b2_maxPolygonVertices];
		for (int32 k = 0; k < poly->m_count; ++k) world[k] = b2Mul(xf, poly->m_vertices[k]);
		m_debugDraw->DrawSolidPolygon(world, poly->m_count, color);
		break;
	}
	default:
		break;
	}
}

void b2World::DrawJoint(b2Joint* joint)
{
	const b2Vec2 originA = joint->GetBodyA()->GetTransform().p, originB = joint->GetBodyB()->GetTransform().p;
	const b2Vec2 anchorA = joint->GetAnchorA(), anchorB = joint->GetAnchorB();
	const b2Color line(0.5f, 0.8f, 0.8f);
	switch (joint->GetType())
	{
	case e_distanceJoint:
		m_debugDraw->DrawSegment(anchorA, anchorB, line);
		break;
	case e_pulleyJoint:
	{
		const b2PulleyJoint* pulley = static_cast<const b2PulleyJoint*>(joint);
		const b2Vec2 groundA = pulley->GetGroundAnchorA(), groundB = pulley->GetGroundAnchorB();
		m_debugDraw->DrawSegment(groundA, anchorA, line);
		m_debugDraw->DrawSegment(groundB, anchorB, line);
		m_debugDraw->DrawSegment(groundA, groundB, line);
		break;
	}
	case e_mouseJoint:
		m_debugDraw->DrawPoint(anchorA, 4.0f, b2Color(0.0f, 1.0f, 0.0f));
		m_debugDraw->DrawPoint(anchorB, 4.0f, b2Color(0.0f, 1.0f, 0.0f));
		m_debugDraw->DrawSegment(anchorA, anchorB, b2Color(0.8f, 0.8f, 0.8f));
		break;
	default:
		m_debugDraw->DrawSegment(originA, anchorA, line);
		m_debugDraw->DrawSegment(anchorA, anchorB, line);
		m_debugDraw->DrawSegment(originB, anchorB, line);
	}
}

void b2World::DrawDebugData()
{
	if (m_debugDraw == nullptr) return;
	const uint32 flags = m_debugDraw->GetFlags();
	if (flags & b2Draw::e_shapeBit)
	{
		// inactive, static, kinematic, asleep, awake: the reference's five colours
		static const float32 palette[5][3] = { { 0.5f, 0.5f, 0.3f }, { 0.5f, 0.9f, 0.5f }, { 0.5f, 0.5f, 0.9f }, { 0.6f, 0.6f, 0.6f }, { 0.9f, 0.7f, 0.7f } };
		for (b2Body* b = m_bodyList; b; b = b->GetNext())
		{
			const int state = !b->IsActive() ? 0 : (b->GetType() == b2_staticBody ? 1 : (b->GetType() == b2_kinematicBody ? 2 : (!b->IsAwake() ? 3 : 4)));
			const b2Color color(palette[state][0], palette[state][1], palette[state][2]);
			const b2Transform& xf = b->GetTransform();
			for (const b2Fixture* f = b->GetFixtureList(); f; f = f->GetNext()) DrawShape(f, xf, color);
		}
	}
	if (flags & b2Draw::e_jointBit)
	{
		for (b2Joint* j = m_jointList; j; j = j->GetNext()) DrawJoint(j);
	}
	// (e_pairBit: the reference's loop over the contacts draws nothing - its body is commented out, b2World.cpp:1985-1997)
	if (flags & b2Draw::e_aabbBit)
	{
		const b2Color color(0.9f, 0.3f, 0.9f);
		for (b2Body* b = m_bodyList; b; b = b->GetNext())
		{
			if (!b->IsActive()) continue;
			for (const b2Fixture* f = b->GetFixtureList(); f; f = f->GetNext())
			{
				for (int32 child = 0; child < f->GetShape()->GetChildCount(); ++child)
				{
					const b2AABB& box = f->GetAABB(child); // the fat AABB the device's broad-phase holds for this proxy
					const b2Vec2 corners[4] = { b2Vec2(box.lowerBound.x, box.lowerBound.y), b2Vec2(box.upperBound.x, box.lowerBound.y),
						b2Vec2(box.upperBound.x, box.upperBound.y), b2Vec2(box.lowerBound.x, box.upperBound.y) };
					m_debugDraw->DrawPolygon(corners, 4, color);
				}
			}
		}
	}
	if (flags & b2Draw::e_centerOfMassBit)
	{
		for (b2Body* b = m_bodyList; b; b = b->GetNext())
		{
			b2Transform xf = b->GetTransform();
			xf.p = b->GetWorldCenter();
			m_debugDraw->DrawTransform(xf);
		}
	}
}

// ---- b2World::Dump (reference: b2World.cpp:2107-2164, b2Body.cpp:632-661, b2Fixture.cpp:203-280: the world as C++ statements
// through b2Log). Bodies, fixtures and shapes are written in full; a joint is written as its generic definition (type, the two
// bodies by index, world anchors, collideConnected) - the per-type parameters are the device record's, not kept on the host.
void b2World::Dump()
{
	if (m_locked) return;
	b2Log("b2Vec2 g(%.15lef, %.15lef);\n", m_gravity.x, m_gravity.y);
	b2Log("m_world->SetGravity(g);\n");
	b2Log("b2Body** bodies = (b2Body**)b2Alloc(%d * sizeof(b2Body*));\n", m_bodyCount);
	b2Log("b2Joint** joints = (b2Joint**)b2Alloc(%d * sizeof(b2Joint*));\n", m_jointCount);
	std::vector<const b2Body*> order;
	for (b2Body* b = m_bodyList; b; b = b->GetNext()) order.push_back(b);
	for (size_t i = 0; i < order.size(); ++i)
	{
		const b2Body* b = order[i];
		b2Log("{\n  b2BodyDef bd;\n");
		b2Log("  bd.type = b2BodyType(%d);\n", (int)b->GetType());
		b2Log("  bd.position.Set(%.15lef, %.15lef);\n", b->GetPosition().x, b->GetPosition().y);
		b2Log("  bd.angle = %.15lef;\n", b->GetAngle());
		b2Log("  bd.linearVelocity.Set(%.15lef, %.15lef);\n", b->GetLinearVelocity().x, b->GetLinearVelocity().y);
		b2Log("  bd.angularVelocity = %.15lef;\n", b->GetAngularVelocity());
		b2Log("  bd.linearDamping = %.15lef;\n", b->GetLinearDamping());
		b2Log("  bd.angularDamping = %.15lef;\n", b->GetAngularDamping());
		b2Log("  bd.allowSleep = bool(%d);\n", b->IsSleepingAllowed() ? 1 : 0);
		b2Log("  bd.awake = bool(%d);\n", b->IsAwake() ? 1 : 0);
		b2Log("  bd.fixedRotation = bool(%d);\n", b->IsFixedRotation() ? 1 : 0);
		b2Log("  bd.bullet = bool(%d);\n", b->IsBullet() ? 1 : 0);
		b2Log("  bd.active = bool(%d);\n", b->IsActive() ? 1 : 0);
		b2Log("  bd.gravityScale = %.15lef;\n", b->GetGravityScale());
		b2Log("  bodies[%d] = m_world->CreateBody(&bd);\n", (int)i);
		for (const b2Fixture* f = b->GetFixtureList(); f; f = f->GetNext())
		{
			b2Log("\n  {\n    b2FixtureDef fd;\n");
			b2Log("    fd.friction = %.15lef;\n    fd.restitution = %.15lef;\n    fd.density = %.15lef;\n", f->GetFriction(), f->GetRestitution(), f->GetDensity());
			b2Log("    fd.isSensor = bool(%d);\n    fd.thickShape = bool(%d);\n", f->IsSensor() ? 1 : 0, f->IsThickShape() ? 1 : 0);
			b2Log("    fd.filter.categoryBits = uint16(%d);\n    fd.filter.maskBits = uint16(%d);\n    fd.filter.groupIndex = int16(%d);\n",
				(int)f->GetFilterData().categoryBits, (int)f->GetFilterData().maskBits, (int)f->GetFilterData().groupIndex);
			const b2Shape* shape = f->GetShape();
			if (shape->GetType() == b2Shape::e_circle)
			{
				const b2CircleShape* s = static_cast<const b2CircleShape*>(shape);
				b2Log("    b2CircleShape shape;\n    shape.m_radius = %.15lef;\n    shape.m_p.Set(%.15lef, %.15lef);\n", s->m_radius, s->m_p.x, s->m_p.y);
			}
			else if (shape->GetType() == b2Shape::e_edge)
			{
				const b2EdgeShape* s = static_cast<const b2EdgeShape*>(shape);
				b2Log("    b2EdgeShape shape;\n    shape.m_radius = %.15lef;\n", s->m_radius);
				b2Log("    shape.m_vertex0.Set(%.15lef, %.15lef);\n    shape.m_vertex1.Set(%.15lef, %.15lef);\n", s->m_vertex0.x, s->m_vertex0.y, s->m_vertex1.x, s->m_vertex1.y);
				b2Log("    shape.m_vertex2.Set(%.15lef, %.15lef);\n    shape.m_vertex3.Set(%.15lef, %.15lef);\n", s->m_vertex2.x, s->m_vertex2.y, s->m_vertex3.x, s->m_vertex3.y);
				b2Log("    shape.m_hasVertex0 = bool(%d);\n    shape.m_hasVertex3 = bool(%d);\n", s->m_hasVertex0 ? 1 : 0, s->m_hasVertex3 ? 1 : 0);
			}
			else if (shape->GetType() == b2Shape::e_polygon)
			{
				const b2PolygonShape* s = static_cast<const b2PolygonShape*>(shape);
				b2Log("    b2PolygonShape shape;\n    b2Vec2 vs[%d];\n", (int)b2_maxPolygonVertices);
				for (int32 k = 0; k < s->m_count; ++k) b2Log("    vs[%d].Set(%.15lef, %.15lef);\n", (int)k, s->m_vertices[k].x, s->m_vertices[k].y);
				b2Log("    shape.Set(vs, %d);\n", (int)s->m_count);
			}
			else if (shape->GetType() == b2Shape::e_chain)
			{
				const b2ChainShape* s = static_cast<const b2ChainShape*>(shape);
				b2Log("    b2ChainShape shape;\n    b2Vec2 vs[%d];\n", (int)s->m_count);
				for (int32 k = 0; k < s->m_count; ++k) b2Log("    vs[%d].Set(%.15lef, %.15lef);\n", (int)k, s->m_vertices[k].x, s->m_vertices[k].y);
				b2Log("    shape.CreateChain(vs, %d);\n", (int)s->m_count);
				b2Log("    shape.m_prevVertex.Set(%.15lef, %.15lef);\n    shape.m_nextVertex.Set(%.15lef, %.15lef);\n", s->m_prevVertex.x, s->m_prevVertex.y, s->m_nextVertex.x, s->m_nextVertex.y);
				b2Log("    shape.m_hasPrevVertex = bool(%d);\n    shape.m_hasNextVertex = bool(%d);\n", s->m_hasPrevVertex ? 1 : 0, s->m_hasNextVertex ? 1 : 0);
			}
			b2Log("\n    fd.shape = &shape;\n\n    bodies[%d]->CreateFixture(&fd);\n  }\n", (int)i);
		}
		b2Log("}\n");
	}
	auto indexOf = [&order](const b2Body* b) { for (size_t i = 0; i < order.size(); ++i) if (order[i] == b) return (int)i; return -1; };
	int jointIndex = 0;
	for (int pass = 0; pass < 2; ++pass) // (gear joints last: they refer to other joints)
	{
		for (b2Joint* j = m_jointList; j; j = j->GetNext())
		{
			if ((j->GetType() == e_gearJoint) != (pass == 1)) continue;
			const b2Vec2 a = j->GetAnchorA(), b = j->GetAnchorB();
			b2Log("{\n  // joint %d: b2JointType(%d), bodies[%d] - bodies[%d], collideConnected %d\n", jointIndex, (int)j->GetType(), indexOf(j->GetBodyA()), indexOf(j->GetBodyB()), j->GetCollideConnected() ? 1 : 0);
			b2Log("  // world anchors (%.15lef, %.15lef) (%.15lef, %.15lef)\n}\n", a.x, a.y, b.x, b.y);
			++jointIndex;
		}
	}
	b2Log("b2Free(joints);\nb2Free(bodies);\njoints = nullptr;\nbodies = nullptr;\n");
}
