/* b2hip.h - C ABI of the MI355X-native b2World::Step() hot path (libb2hip.so).
 *
 * This is the drop-in boundary: plain C structs, pointers and sizes, no C++ and no torch types.
 * The host-side Box2D API mirror (box2d-mt_amd/host/Box2D/...) is a thin C++ layer that binds
 * exactly these entry points; INTEGRATION.md shows the binding a maintainer of the reference
 * would add inside b2World / b2Body.
 *
 * Reference interfaces replaced (all paths relative to the reference tree):
 *   b2hip_world_create / destroy     b2World::b2World, ~b2World            Box2D/Dynamics/b2World.cpp:446-520
 *   b2hip_create_body                b2World::CreateBody + b2Body::b2Body   b2World.cpp:548-582, b2Body.cpp:26-112
 *   b2hip_create_fixture             b2Body::CreateFixture, b2Fixture::Create/CreateProxies,
 *                                    b2Body::ResetMassData                  b2Body.cpp:182-226,310-385; b2Fixture.cpp:42-141
 *   b2hip_create_revolute_joint      b2World::CreateJoint (revolute)        b2World.cpp:672-760, Joints/b2RevoluteJoint.cpp:47-63
 *   b2hip_create_distance_joint      b2World::CreateJoint (distance)        b2World.cpp:672-760, Joints/b2DistanceJoint.cpp:51-63
 *   b2hip_create_prismatic_joint     b2World::CreateJoint (prismatic)       b2World.cpp:672-760, Joints/b2PrismaticJoint.cpp:98-128
 *   b2hip_create_weld_joint          b2World::CreateJoint (weld)            b2World.cpp:672-760, Joints/b2WeldJoint.cpp:46-56
 *   b2hip_create_wheel_joint         b2World::CreateJoint (wheel)           Joints/b2WheelJoint.cpp:47-77
 *   b2hip_create_rope_joint          b2World::CreateJoint (rope)            Joints/b2RopeJoint.cpp:34-46
 *   b2hip_create_friction_joint      b2World::CreateJoint (friction)        Joints/b2FrictionJoint.cpp:45-56
 *   b2hip_create_motor_joint         b2World::CreateJoint (motor)           Joints/b2MotorJoint.cpp:48-60
 *   b2hip_create_pulley_joint        b2World::CreateJoint (pulley)          Joints/b2PulleyJoint.cpp:62-79
 *   b2hip_create_mouse_joint         b2World::CreateJoint (mouse)           Joints/b2MouseJoint.cpp:36-55
 *   b2hip_create_gear_joint          b2World::CreateJoint (gear)            Joints/b2GearJoint.cpp:50-129
 *   b2hip_joint_set_target           b2MouseJoint::SetTarget                Joints/b2MouseJoint.cpp:57-64
 *   b2hip_joint_set_offsets          b2MotorJoint::SetLinearOffset / SetAngularOffset   Joints/b2MotorJoint.cpp:253-281
 *   b2hip_destroy_joint              b2World::DestroyJoint                  b2World.cpp:762-846
 *   b2hip_joint_set_motor            b2{Revolute,Prismatic,Wheel}Joint::EnableMotor / SetMotorSpeed / SetMaxMotor{Torque,Force}
 *                                                                           Joints/b2RevoluteJoint.cpp:418-452, b2PrismaticJoint.cpp:588-616
 *   b2hip_joint_set_limits           b2{Revolute,Prismatic}Joint::EnableLimit / SetLimits
 *                                                                           Joints/b2RevoluteJoint.cpp:459-500, b2PrismaticJoint.cpp:549-581
 *   b2hip_destroy_body / _fixture    b2World::DestroyBody, b2Body::DestroyFixture  b2World.cpp:585-670, b2Body.cpp:238-308
 *   b2hip_set_transform / _awake / _bullet, b2hip_apply_*_impulse           b2Body.cpp:451-473, 575-601; b2Body.h:690-718, 885-950
 *   b2hip_fixture_set_sensor / _thick / _filter, b2hip_fixture_refilter     b2Fixture.cpp:180-257
 *   b2hip_step                       b2World::Step                          b2World.cpp:1613-1710
 *   b2hip_collide                    b2World::Collide / b2ContactManager::Collide      b2World.cpp:1120-1141, b2ContactManager.cpp:177-230
 *   b2hip_solve                      b2World::Solve (islands + b2Island::Solve)        b2World.cpp:1166-1431, b2Island.cpp:184-396
 *   b2hip_sync_fixtures              b2World::SynchronizeFixtures                      b2World.cpp:1143-1164, b2ContactManager.cpp:315-364,441-452
 *   b2hip_find_new_contacts          b2World::FindNewContacts / b2BroadPhase::UpdatePairs / AddPair
 *                                                                                       b2World.cpp:1095-1118, b2BroadPhase.h:211-267, b2ContactManager.cpp:237-312,366-386
 *   b2hip_solve_toi                  b2World::SolveTOI                                 b2World.cpp:1026-1093, 851-1024
 *   b2hip_get_fat_aabbs              (served to) b2World::QueryAABB / RayCast          b2World.cpp:1740-1795, b2DynamicTree.h:168-287
 *   b2hip_get_body_states            b2Body::GetPosition/GetAngle/GetLinearVelocity/GetAngularVelocity/IsAwake  b2Body.h:516-700
 *   b2hip_enable/get_contact_events  b2ContactListener::BeginContact / EndContact     b2WorldCallbacks.h:88-104, b2ContactManager.cpp:420-438
 *   b2hip_set_pre_solve              b2ContactListener::PreSolve(Immediate)           b2WorldCallbacks.h:105-174, b2Contact.cpp:283-297
 *   b2hip_enable/get_post_solve      b2ContactListener::PostSolve(Immediate)          b2Island.cpp:532-570, b2ContactManager.cpp:454-470
 *   b2hip_set_contact_filter         b2ContactFilter::ShouldCollide                   b2WorldCallbacks.h:52-63, b2ContactManager.cpp:283-287
 *   b2hip_save / load_snapshot       (new: binary checkpoint; cf. b2World::Dump          b2World.cpp:2107-2164)
 *   b2hip_get_contacts               b2World::GetContactList + b2Contact::GetManifold  b2World.h:352-360, b2Contact.h:95-163
 *   b2hip_get_profile                b2World::GetProfile                    b2World.h:196-197, b2TimeStep.h:25-40
 *
 * Conventions: every function returns 0 on success and a negative b2hip_status on failure
 * (b2hip_last_error() gives the text); no exceptions cross the boundary; host buffers are owned by
 * the caller, device buffers by the world; all calls for one world come from one thread; work is
 * stream-ordered on the world's HIP stream and only the download / step calls block.
 * There is NO CPU fallback: if no HIP device is usable, b2hip_world_create fails.
 */
#ifndef B2HIP_H
#define B2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct b2hip_world b2hip_world;

typedef enum b2hip_status
{
	B2HIP_OK = 0,
	B2HIP_ERR_INVALID = -1,     /* bad argument / unknown id */
	B2HIP_ERR_HIP = -2,         /* a HIP runtime call failed */
	B2HIP_ERR_NO_DEVICE = -3,   /* no usable gfx950 device */
	B2HIP_ERR_UNSUPPORTED = -4, /* feature outside the device path */
	B2HIP_ERR_CAPACITY = -5     /* a device buffer overflowed and could not be regrown */
} b2hip_status;

enum { B2HIP_STATIC_BODY = 0, B2HIP_KINEMATIC_BODY = 1, B2HIP_DYNAMIC_BODY = 2 }; /* b2BodyType, b2Body.h:36-47 */
enum { B2HIP_SHAPE_CIRCLE = 0, B2HIP_SHAPE_EDGE = 1, B2HIP_SHAPE_POLYGON = 2,          /* b2Shape::Type, b2Shape.h:53-60 */
       /* ONE child of a b2ChainShape (b2ChainShape.h:32, GetChildEdge b2ChainShape.cpp:114-147): the record of an edge whose
        * ghost vertices are the child's neighbours. It collides, sweeps and ray-casts as that edge; its broad-phase AABB has no
        * radius (b2ChainShape.cpp:174-189). A chain fixture of n children is n consecutive fixtures of this type, created in
        * child order (the reference creates one proxy per child in that order, b2Fixture.cpp:126-141). */
       B2HIP_SHAPE_CHAIN = 3 };

typedef struct b2hip_world_def
{
	float gravity_x, gravity_y;
	int allow_sleep;      /* b2World::SetAllowSleeping      default 1 */
	int warm_starting;    /* b2World::SetWarmStarting       default 1 */
	int continuous;       /* b2World::SetContinuousPhysics  default 1 in the reference; continuous collision (TOI) runs on the device: b2hip_solve_toi */
	int sub_stepping;     /* b2World::SetSubStepping        default 0 ; a step call solves ONE TOI event and leaves the step open; the calls that
	                       *   follow run Collide and the next event, no island solve, until no event is left (b2World.cpp:1082-1086, 1668) */
	int auto_clear_forces;/* b2World::SetAutoClearForces    default 1 */
	int device;           /* HIP device ordinal, -1 = current */
} b2hip_world_def;

/* b2BodyDef (b2Body.h:52-129) */
typedef struct b2hip_body_def
{
	int type;
	float px, py, angle;
	float vx, vy, w;
	float linear_damping, angular_damping, gravity_scale;
	int allow_sleep, awake, fixed_rotation, bullet, active;
} b2hip_body_def;

/* A finished collision shape (what b2Shape::Clone would copy). Layout == device ShapeRec.
 *   circle : verts[0..1] = m_p, radius = m_radius
 *   edge   : verts[0..1] = m_vertex1, [2..3] = m_vertex2, [4..5] = m_vertex0, [6..7] = m_vertex3,
 *            count bit0 = m_hasVertex0, bit1 = m_hasVertex3, radius = b2_polygonRadius
 *   polygon: count, verts[2*count], normals[2*count], centroid, radius = b2_polygonRadius */
typedef struct b2hip_shape
{
	int32_t type;
	int32_t count;
	float radius;
	float pad;
	float centroid[2];
	float verts[16];
	float normals[16];
} b2hip_shape;

/* b2FixtureDef (b2Fixture.h:56-94) */
typedef struct b2hip_fixture_def
{
	float density, friction, restitution;
	uint16_t category_bits, mask_bits;
	int16_t group_index;
	int16_t pad;
	int is_sensor;
	int thick_shape;
} b2hip_fixture_def;

/* b2RevoluteJointDef (Joints/b2RevoluteJoint.h:35-85) */
typedef struct b2hip_revolute_joint_def
{
	int body_a, body_b;
	float local_anchor_a[2], local_anchor_b[2];
	float reference_angle;
	int enable_limit;
	float lower_angle, upper_angle;
	int enable_motor;
	float motor_speed, max_motor_torque;
	int collide_connected;
} b2hip_revolute_joint_def;

/* b2DistanceJointDef (Joints/b2DistanceJoint.h:31-68): rigid rod when frequency_hz == 0, damped spring otherwise */
typedef struct b2hip_distance_joint_def
{
	int body_a, body_b;
	float local_anchor_a[2], local_anchor_b[2];
	float length;
	float frequency_hz, damping_ratio;
	int collide_connected;
} b2hip_distance_joint_def;

/* b2PrismaticJointDef (Joints/b2PrismaticJoint.h:31-85): bodyB slides along local_axis_a of bodyA, no relative rotation */
typedef struct b2hip_prismatic_joint_def
{
	int body_a, body_b;
	float local_anchor_a[2], local_anchor_b[2];
	float local_axis_a[2];          /* normalised at creation, like b2PrismaticJoint's constructor does */
	float reference_angle;
	int enable_limit;
	float lower_translation, upper_translation;
	int enable_motor;
	float motor_speed, max_motor_force;
	int collide_connected;
} b2hip_prismatic_joint_def;

/* b2WeldJointDef (Joints/b2WeldJoint.h:28-60): rigid when frequency_hz == 0, else a soft angular spring */
typedef struct b2hip_weld_joint_def
{
	int body_a, body_b;
	float local_anchor_a[2], local_anchor_b[2];
	float reference_angle;
	float frequency_hz, damping_ratio;
	int collide_connected;
} b2hip_weld_joint_def;

/* b2WheelJointDef (Joints/b2WheelJoint.h:31-76): point-on-line along local_axis_a (used as given), suspension spring
 * along the axis (frequency_hz > 0) and a rotational motor */
typedef struct b2hip_wheel_joint_def
{
	int body_a, body_b;
	float local_anchor_a[2], local_anchor_b[2];
	float local_axis_a[2];
	float frequency_hz, damping_ratio;
	int enable_motor;
	float motor_speed, max_motor_torque;
	int collide_connected;
} b2hip_wheel_joint_def;

/* b2RopeJointDef (Joints/b2RopeJoint.h:28-52): the anchors may not get further apart than max_length */
typedef struct b2hip_rope_joint_def
{
	int body_a, body_b;
	float local_anchor_a[2], local_anchor_b[2];
	float max_length;
	int collide_connected;
} b2hip_rope_joint_def;

/* b2FrictionJointDef (Joints/b2FrictionJoint.h:26-51): top-down friction, force and torque capped */
typedef struct b2hip_friction_joint_def
{
	int body_a, body_b;
	float local_anchor_a[2], local_anchor_b[2];
	float max_force, max_torque;
	int collide_connected;
} b2hip_friction_joint_def;

/* b2PulleyJointDef (Joints/b2PulleyJoint.h:28-76): length_a + ratio * length_b stays constant; ground anchors in world space */
typedef struct b2hip_pulley_joint_def
{
	int body_a, body_b;
	float ground_anchor_a[2], ground_anchor_b[2];
	float local_anchor_a[2], local_anchor_b[2];
	float length_a, length_b;
	float ratio;
	int collide_connected;
} b2hip_pulley_joint_def;

/* b2MouseJointDef (Joints/b2MouseJoint.h:27-57): pulls the point of bodyB that lies at `target` (world) when the joint is
 * created towards wherever the target is moved; bodyA is only bookkeeping (the Testbed passes the ground body) */
typedef struct b2hip_mouse_joint_def
{
	int body_a, body_b;
	float target[2];
	float max_force;
	float frequency_hz, damping_ratio;
	int collide_connected;
} b2hip_mouse_joint_def;

/* b2GearJointDef (Joints/b2GearJoint.h:28-48): couples two existing revolute / prismatic joints (ids returned by their
 * create calls): coordinate1 + ratio * coordinate2 stays what it is at creation. The joint's bodies are the second bodies of the
 * two joints, as in the reference's constructor; the two joints must outlive the gear. */
typedef struct b2hip_gear_joint_def
{
	int joint1, joint2;
	float ratio;
	int collide_connected;
} b2hip_gear_joint_def;

/* b2MotorJointDef (Joints/b2MotorJoint.h:26-57): drives bodyB to linear_offset / angular_offset in bodyA's frame */
typedef struct b2hip_motor_joint_def
{
	int body_a, body_b;
	float linear_offset[2];
	float angular_offset;
	float max_force, max_torque;
	float correction_factor;
	int collide_connected;
} b2hip_motor_joint_def;

/* Host-visible body state after a step (40 bytes per body, one coalesced device->host copy). */
typedef struct b2hip_body_state
{
	float px, py;      /* b2Body::GetPosition  (m_xf.p)   */
	float angle;       /* b2Body::GetAngle     (m_sweep.a) */
	float vx, vy, w;   /* linear / angular velocity        */
	float cx, cy;      /* b2Body::GetWorldCenter (m_sweep.c) */
	uint32_t flags;    /* bit0-1 type, bit2 awake, bit3 autoSleep, bit4 bullet, bit5 fixedRotation, bit6 active */
	float sleep_time;
} b2hip_body_state;

#define B2HIP_BODY_AWAKE 0x4u

/* b2MassData + centre (b2Body::GetMass/GetInertia/GetLocalCenter) */
typedef struct b2hip_mass_data
{
	float mass, inertia;      /* inertia about the local origin, as b2Body::GetInertia returns it */
	float local_center[2];
	float inv_mass, inv_inertia;
} b2hip_mass_data;

/* One contact (b2Contact): fixture ids are the ids b2hip_create_fixture returned. */
typedef struct b2hip_contact
{
	int32_t fixture_a, fixture_b;
	int32_t body_a, body_b;
	uint32_t flags;            /* bit0 touching, bit1 enabled */
	int32_t manifold_type;     /* b2Manifold::Type */
	int32_t point_count;
	float local_normal[2];
	float local_point[2];
	float point_local[2][2];
	float normal_impulse[2];
	float tangent_impulse[2];
	uint32_t id_key[2];
	float friction, restitution;
	float tangent_speed;       /* b2Contact::GetTangentSpeed (b2Contact.h:157-160) */
} b2hip_contact;

/* Counters of the last step (device truth, read back with the body states). */
typedef struct b2hip_counters
{
	int32_t bodies, proxies, contacts, touching_contacts;
	int32_t islands, small_islands, large_islands;
	int32_t small_island_bodies, small_island_contacts, large_island_bodies, large_island_contacts;
	int32_t colors, moved_proxies, new_contacts, destroyed_contacts;
	int32_t solver_chunks;
	int32_t pos_iterations_large;
	int32_t overflow_flags;
	int32_t toi_events;              /* TOI sub-steps solved in the last step (b2World::StepSolveTOI calls) */
	int32_t toi_calls;               /* b2TimeOfImpact evaluations in the last step */
	int32_t toi_pending_first_pass;  /* contacts whose first-pass time of impact was < 1 */
	int32_t toi_serial_fallbacks;    /* steps (since creation) whose parallel TOI chains were redone by the serial event loop */
	/* block partition of the large islands (one workgroup solves one block with its bodies in LDS) */
	int32_t blocks;                  /* blocks of the current partition */
	int32_t cut_constraints;         /* constraints between bodies of two blocks in the last step */
	int32_t block_max_rows;          /* most constraints owned by one block in the last step */
	int32_t partitions;              /* partitions made since the world was created */
	int32_t block_solver_steps;      /* steps whose large islands were solved by the block solver */
	int32_t free_islands;            /* one-body islands without contact or joint in the last step (stepped without the island solver) */
	int32_t sweep_solver_steps;      /* steps whose (jointed / hub) large islands took the one-launch-per-sweep block kernel */
	/* hub bodies (more contacts than can be coloured): their constraints are swept in order by one wave (k_large_hub) */
	int32_t hub_constraints;         /* hub constraints in the last step */
	int32_t hub_fixpoint_rounds;     /* rounds of the fixed-point form of that sweep in the last step (all sweeps) */
	int32_t hub_serial_chunks;       /* chunks of 64 hub constraints swept lane after lane instead */
	int32_t toi_chain_contacts;      /* contacts (since creation) the parallel TOI chains left for their close-out to create in event order */
	int32_t toi_pre_solve_reruns;    /* runs of the TOI phase (since creation) repeated because a PreSolve changed its contact inside a sub-step */
	int32_t solver_recoveries;       /* large-island solves (since creation) run a second time, launch by launch, because a wait between the
	                                    workgroups of a resident / data-flow solver kernel timed out, or whose colouring ran out of colours (such constraints are
	                                    swept in order) or of rounds (finished grid-wide); the step itself succeeds (B2HIP_NO_RECOVER=1: it fails as before) */
} b2hip_counters;

const char* b2hip_last_error(void);
const char* b2hip_version(void);

int b2hip_world_create(const b2hip_world_def* def, b2hip_world** out);
void b2hip_world_destroy(b2hip_world* w);

int b2hip_set_gravity(b2hip_world* w, float gx, float gy);
/* b2World::ShiftOrigin (b2World.cpp:1862-1887): bodies, broad-phase boxes and the world-space anchors of mouse and pulley
 * joints move by -(x, y); between steps only. */
int b2hip_shift_origin(b2hip_world* w, float x, float y);
int b2hip_set_flags(b2hip_world* w, int allow_sleep, int warm_starting, int continuous, int sub_stepping);

/* Returns the new body / fixture / joint id (>= 0) or a negative status. */
int b2hip_create_body(b2hip_world* w, const b2hip_body_def* def);
int b2hip_create_fixture(b2hip_world* w, int body, const b2hip_fixture_def* def, const b2hip_shape* shape);
int b2hip_create_revolute_joint(b2hip_world* w, const b2hip_revolute_joint_def* def);
int b2hip_create_distance_joint(b2hip_world* w, const b2hip_distance_joint_def* def);
int b2hip_create_prismatic_joint(b2hip_world* w, const b2hip_prismatic_joint_def* def);
int b2hip_create_weld_joint(b2hip_world* w, const b2hip_weld_joint_def* def);
int b2hip_create_wheel_joint(b2hip_world* w, const b2hip_wheel_joint_def* def);
int b2hip_create_rope_joint(b2hip_world* w, const b2hip_rope_joint_def* def);
int b2hip_create_friction_joint(b2hip_world* w, const b2hip_friction_joint_def* def);
int b2hip_create_motor_joint(b2hip_world* w, const b2hip_motor_joint_def* def);
int b2hip_create_pulley_joint(b2hip_world* w, const b2hip_pulley_joint_def* def);
int b2hip_create_mouse_joint(b2hip_world* w, const b2hip_mouse_joint_def* def);
int b2hip_create_gear_joint(b2hip_world* w, const b2hip_gear_joint_def* def);
/* b2MouseJoint::SetTarget (b2MouseJoint.cpp:57-64): wakes bodyB when the target moves */
int b2hip_joint_set_target(b2hip_world* w, int joint, float x, float y);
/* b2MotorJoint::SetLinearOffset + SetAngularOffset (b2MotorJoint.cpp:253-281): wakes both bodies when something changes */
int b2hip_joint_set_offsets(b2hip_world* w, int joint, float linear_x, float linear_y, float angular);
/* b2World::DestroyJoint (b2World.cpp:762-846): wakes both bodies; contacts between them are filtered again if the joint
 * kept them from colliding. Joint ids are never reused (the id of a destroyed joint stays invalid). Destroy a gear joint
 * before the joints it couples. */
int b2hip_destroy_joint(b2hip_world* w, int joint);
/* Revolute / prismatic / wheel (motor only) joints between steps: EnableMotor + SetMotorSpeed + SetMaxMotorTorque|Force in one call, and
 * EnableLimit + SetLimits in one call. Like the reference's setters, a call that changes something wakes both bodies and
 * (limits) restarts the limit impulse from zero; a call that changes nothing does nothing. */
int b2hip_joint_set_motor(b2hip_world* w, int joint, int enable_motor, float motor_speed, float max_motor);
int b2hip_joint_set_limits(b2hip_world* w, int joint, int enable_limit, float lower, float upper);
/* b2Joint::GetReactionForce / GetReactionTorque(inv_dt) and the motor's share (b2RevoluteJoint::GetMotorTorque,
 * b2PrismaticJoint::GetMotorForce, b2WheelJoint::GetMotorTorque) after the last step, from the joint's accumulated impulses
 * (b2Joint.h:129-133; per type: b2RevoluteJoint.cpp:439-456, b2PrismaticJoint.cpp:502-510,618-621, ...).
 * out4 = force.x, force.y, torque, motor. One small device read per call (between steps). */
int b2hip_get_joint_reaction(b2hip_world* w, int joint, float inv_dt, float out4[4]);
/* b2RopeJoint::GetLimitState (b2RopeJoint.h:84; also the limit state of revolute / prismatic joints): 0 inactive, 1 at lower,
 * 2 at upper, 3 equal limits (b2LimitState, b2Joint.h:58-64) as the last step's solver left it; negative = error code. */
int b2hip_get_joint_limit_state(b2hip_world* w, int joint);

/* ---- Life cycle and mutators between steps (all refused inside a step, like the reference's locked world) -------------------
 * Ids are never reused: a destroyed body / fixture keeps its id (every getter reports it as destroyed), so ids handed
 * out earlier stay valid. Edits that touch the contact array (destroys, TOI candidacy, re-filtering) are applied in call
 * order by one device pass at the start of the next step, or when the contacts are read.
 *
 * b2World::DestroyBody (b2World.cpp:585-670): its joints (b2hip_destroy_joint each, newest first), then its contacts
 * (newest first; a touching one reports its EndContact), then its fixtures and their broad-phase proxies (newest first:
 * proxy ids are freed and reused exactly as the reference's tree does), then the body leaves m_nonStaticBodies (the last
 * non-static body takes its slot, which is the order islands are seeded in). */
int b2hip_destroy_body(b2hip_world* w, int body);
/* b2Body::DestroyFixture (b2Body.cpp:238-308): the fixture's contacts, its proxy, then ResetMassData */
int b2hip_destroy_fixture(b2hip_world* w, int fixture);
/* b2Body::SetTransform (b2Body.cpp:451-473): pose and sweep (origin included) are set, every fixture is synchronised with
 * zero displacement (a proxy that leaves its fat AABB is re-inserted and buffered as moved) */
int b2hip_set_transform(b2hip_world* w, int body, float x, float y, float angle);
/* b2Body::SetAwake (b2Body.h:690-718): true restarts the sleep timer; false also zeroes velocities and forces */
int b2hip_set_awake(b2hip_world* w, int body, int awake);
/* b2Body::SetActive (b2Body.cpp:496-544): an inactive body keeps its fixtures but has no broad-phase proxies and no contacts
 * (they are destroyed in its contact-list order) and is skipped by the island build, joints to it included; activating it
 * creates the proxies again, newest fixture first, with fresh proxy ids - contacts follow with the next pair update. */
int b2hip_set_active(b2hip_world* w, int body, int active);
/* b2Body::SetType (b2Body.cpp:118-188): type = B2HIP_STATIC_BODY / KINEMATIC / DYNAMIC. Mass data are recomputed, a body
 * that becomes static stops and its proxies are synchronised, the body moves between the world's static and non-static
 * lists (the island seed order follows), wakes, loses its forces and ALL its contacts; its proxies are touched so that
 * the next pair update forms the contacts its new type allows. */
int b2hip_set_type(b2hip_world* w, int body, int type);
/* b2Body::SetBullet (b2Body.cpp:575-601) + b2ContactManager::RecalculateToiCandidacy (b2ContactManager.cpp:566-640) */
int b2hip_set_bullet(b2hip_world* w, int body, int bullet);
/* b2Body::ApplyLinearImpulse / ApplyLinearImpulseToCenter (point = world centre) / ApplyAngularImpulse (b2Body.h:885-950) */
int b2hip_apply_linear_impulse(b2hip_world* w, int body, float ix, float iy, float px, float py, int wake);
int b2hip_apply_linear_impulse_to_center(b2hip_world* w, int body, float ix, float iy, int wake);
int b2hip_apply_angular_impulse(b2hip_world* w, int body, float impulse, int wake);
/* b2Fixture::SetSensor (b2Fixture.cpp:222-239: wakes the body, re-evaluates TOI candidacy), SetThickShape (:241-257),
 * SetFilterData + Refilter (:180-220: contacts of the fixture are filtered again by the next Collide, the proxy is
 * touched so that new pairs can form) */
int b2hip_fixture_set_sensor(b2hip_world* w, int fixture, int is_sensor);
int b2hip_fixture_set_thick(b2hip_world* w, int fixture, int thick_shape);
int b2hip_fixture_set_filter(b2hip_world* w, int fixture, uint16_t category_bits, uint16_t mask_bits, int16_t group_index);
int b2hip_fixture_refilter(b2hip_world* w, int fixture);
/* b2Fixture::SetDensity / SetFriction / SetRestitution (b2Fixture.h:306-334): the density is read by the next ResetMassData
 * (b2hip_set_mass_data(w, body, NULL)), friction and restitution by contacts created from now on. */
int b2hip_fixture_set_material(b2hip_world* w, int fixture, float density, float friction, float restitution);
/* b2Body::SetLinearDamping / SetAngularDamping / SetGravityScale (b2Body.h:620-648) */
int b2hip_set_body_damping(b2hip_world* w, int body, float linear_damping, float angular_damping, float gravity_scale);
/* b2Body::SetFixedRotation (b2Body.cpp:546-565) and b2Body::SetSleepingAllowed (b2Body.h:674-688) */
int b2hip_set_fixed_rotation(b2hip_world* w, int body, int flag);
int b2hip_set_sleeping_allowed(b2hip_world* w, int body, int flag);
/* b2WheelJoint / b2DistanceJoint / b2WeldJoint / b2MouseJoint ::SetFrequency / SetSpringFrequencyHz + SetDampingRatio
 * (e.g. b2WheelJoint.h:125-131): plain member writes, nobody is woken */
int b2hip_joint_set_spring(b2hip_world* w, int joint, float frequency_hz, float damping_ratio);
/* The plain scalar setters of the joint classes (assignments in the reference, no wake-up): b2DistanceJoint::SetLength,
 * b2RopeJoint::SetMaxLength (LENGTH); b2FrictionJoint / b2MotorJoint / b2MouseJoint::SetMaxForce (MAX_FORCE);
 * b2FrictionJoint / b2MotorJoint::SetMaxTorque (MAX_TORQUE); b2GearJoint::SetRatio (RATIO);
 * b2MotorJoint::SetCorrectionFactor (CORRECTION_FACTOR). A parameter the joint's type does not have: B2HIP_ERR_INVALID. */
enum { B2HIP_JOINT_LENGTH = 0, B2HIP_JOINT_MAX_FORCE = 1, B2HIP_JOINT_MAX_TORQUE = 2, B2HIP_JOINT_RATIO = 3, B2HIP_JOINT_CORRECTION_FACTOR = 4 };
int b2hip_joint_set_param(b2hip_world* w, int joint, int param, float value);
/* 1 if the id names a body / fixture that has been destroyed */
int b2hip_body_is_destroyed(const b2hip_world* w, int body);
int b2hip_fixture_is_destroyed(const b2hip_world* w, int fixture);

int b2hip_body_count(const b2hip_world* w);
int b2hip_fixture_count(const b2hip_world* w);
int b2hip_get_mass_data(const b2hip_world* w, int body, b2hip_mass_data* out);
/* b2Body::SetMassData (b2Body.cpp:387-424: dynamic bodies only; `inertia` about the body origin); mass_data == NULL:
 * b2Body::ResetMassData (b2Body.cpp:310-385: from the fixtures' shapes and densities). */
int b2hip_set_mass_data(b2hip_world* w, int body, const b2hip_mass_data* mass_data);

/* Force / impulse staging between steps (b2Body::ApplyForceToCenter, ApplyTorque, SetLinearVelocity ...). */
int b2hip_apply_force(b2hip_world* w, int body, float fx, float fy, float torque, int wake);
int b2hip_set_velocity(b2hip_world* w, int body, float vx, float vy, float omega);

/* One full b2World::Step on the device, then the body-state read-back. Blocking. */
int b2hip_step(b2hip_world* w, float dt, int velocity_iterations, int position_iterations);

/* The phases of Step as separate entry points (each blocks until its kernels are done) so that every
 * phase can be parity-tested and profiled alone. b2hip_step == begin, [find_new_contacts], collide,
 * solve, sync_fixtures, find_new_contacts, end. */
int b2hip_step_begin(b2hip_world* w, float dt, int velocity_iterations, int position_iterations);
int b2hip_collide(b2hip_world* w);
int b2hip_solve(b2hip_world* w);
int b2hip_sync_fixtures(b2hip_world* w);
int b2hip_find_new_contacts(b2hip_world* w);
/* b2World::SolveTOI (b2World.cpp:1026-1093): continuous collision, after the end-of-step pair update.
 * No-op unless the world was created / flagged with continuous = 1. */
int b2hip_solve_toi(b2hip_world* w);
int b2hip_step_end(b2hip_world* w);

/* Host mirror of the last read-back; valid until the next step. */
int b2hip_get_body_states(b2hip_world* w, int first, int count, b2hip_body_state* out);
int b2hip_contact_count(b2hip_world* w);
int b2hip_get_contacts(b2hip_world* w, int cap, b2hip_contact* out);

/* Contact events: the BeginContact / EndContact half of b2ContactListener (b2WorldCallbacks.h:88-174; generation sites
 * b2Contact.cpp:253-297, b2ContactManager.cpp:104-107; delivery order b2ContactManager.cpp:420-438). When enabled, every
 * step ends with the list of contacts whose touching state changed since the host was last told: all begins in
 * proxy-id-pair order, then all ends in proxy-id-pair order (the order in which the reference delivers its deferred
 * callbacks). One net event per contact and step: a contact that begins AND ends inside one step (continuous-collision
 * sub-steps) produces none, where the reference may call both. `contact_index` = index into b2hip_get_contacts of the
 * same step, or -1 when the contact was destroyed (its end event). Sensors report like any contact. */
typedef struct b2hip_contact_event
{
	int32_t fixture_a, fixture_b;
	int32_t kind;              /* 0 = begin, 1 = end */
	int32_t contact_index;
} b2hip_contact_event;
int b2hip_enable_contact_events(b2hip_world* w, int enable);
/* returns the number of events of the last step (negative on error); at most `cap` are written */
int b2hip_get_contact_events(b2hip_world* w, int cap, b2hip_contact_event* out);
/* ---- The other half of b2ContactListener, and b2ContactFilter: user code in the middle of a step ----------------------------
 * The physics runs on the device, user code on the host: each of these is a plain C callback the step calls on the
 * stepping thread at the point where the reference calls the listener / filter, between two device phases.
 * Installing one costs a device -> host round trip per step at that point (and only then).
 *
 * b2ContactFilter::ShouldCollide (b2WorldCallbacks.h:52-63; call sites b2ContactManager.cpp:283-287 AddPair, :195-203 the
 * re-filter of flagged contacts in Collide): called for every NEW candidate pair that has passed the body rules (same
 * body, existing contact, joints, one dynamic body) and for every contact flagged for re-filtering; return 0 to refuse.
 * An installed filter REPLACES the built-in category / mask / group rule, as a user b2ContactFilter replaces the default
 * one (call b2hip_default_should_collide from it to keep that rule). Not consulted for contacts created inside a
 * continuous-collision sub-step (those are made on the device with the built-in rule). */
typedef int (*b2hip_should_collide_fn)(void* user, int fixture_a, int fixture_b);
int b2hip_set_contact_filter(b2hip_world* w, b2hip_should_collide_fn fn, void* user);
/* ... or all pairs of one decision point in one call (fixture_pairs = count x {a, b}, verdict[i] = 0 refuses pair i): the
 * form for a caller that asks b2ContactFilter::ShouldCollide(fA, fB, threadId) from several threads (b2WorldCallbacks.h:57-62). */
typedef void (*b2hip_should_collide_batch_fn)(void* user, int count, const int32_t* fixture_pairs, int32_t* verdict);
int b2hip_set_contact_filter_batch(b2hip_world* w, b2hip_should_collide_batch_fn fn, void* user);
/* the built-in rule: b2ContactFilter::ShouldCollide (b2WorldCallbacks.cpp:24-38) on the fixtures' filter data */
int b2hip_default_should_collide(b2hip_world* w, int fixture_a, int fixture_b);

/* b2Manifold as the listener sees it (b2Collision.h:93-107) */
typedef struct b2hip_manifold
{
	int32_t type, point_count;
	float local_normal[2], local_point[2];
	float point_local[2][2];
	float normal_impulse[2], tangent_impulse[2];
	uint32_t id_key[2];
} b2hip_manifold;

/* b2ContactListener::PreSolve (b2WorldCallbacks.h:88-174; generation site b2Contact.cpp:283-297, delivery
 * b2ContactManager.cpp:431-434): after Collide, before the islands are built, once per touching non-sensor contact that
 * was updated in this step, in proxy-id-pair order, with the manifold of the previous step. Return 0 to disable the
 * contact for this step (b2Contact::SetEnabled(false): it is left out of the islands; the next Collide enables it again).
 * `contact_index` indexes b2hip_get_contacts. Contacts updated inside continuous-collision sub-steps are not reported. */
/* `material` (in / out): the contact's mixed friction and restitution (b2Contact.h:40-50) and its tangent speed, as
 * b2Contact::SetFriction / SetRestitution / SetTangentSpeed (b2Contact.h:129-160) may change them from inside the callback;
 * what the callback leaves there stays with the contact (this step's solver and every later one, until it is set again). */
typedef struct b2hip_contact_material
{
	float friction, restitution, tangent_speed;
} b2hip_contact_material;
typedef int (*b2hip_pre_solve_fn)(void* user, int contact_index, int fixture_a, int fixture_b,
	const b2hip_manifold* old_manifold, const b2hip_manifold* manifold, b2hip_contact_material* material);
int b2hip_set_pre_solve(b2hip_world* w, b2hip_pre_solve_fn fn, void* user);

/* The same delivery as ONE call per step with all records (in the deferred order): the form a caller uses that wants to run
 * b2ContactListener::PreSolveImmediate on the worker threads of its b2TaskExecutor, as the reference does during Collide
 * (b2WorldCallbacks.h:135-173, b2Contact.cpp:283-297), and the deferred PreSolve afterwards in order. The callee fills
 * `enabled` (in: 1) and `material` (in: the contact's values) of every record. Installing a batch function replaces the
 * per-record one. */
typedef struct b2hip_pre_solve_record
{
	int32_t contact_index, fixture_a, fixture_b;
	int32_t enabled;
	b2hip_manifold old_manifold, manifold;
	b2hip_contact_material material;
} b2hip_pre_solve_record;
typedef void (*b2hip_pre_solve_batch_fn)(void* user, int count, b2hip_pre_solve_record* records);
int b2hip_set_pre_solve_batch(b2hip_world* w, b2hip_pre_solve_batch_fn fn, void* user);

/* The listener calls the reference makes from INSIDE its TOI sub-steps (b2World::SolveTOI -> StepSolveTOI, b2World.cpp:866,946:
 * contact->Update(listener) on the TOI contact and on every contact of the two bodies it visits; b2Island::SolveTOI ->
 * Report, b2Island.cpp:527): the last step's calls in the reference's call order - event after event, inside an event the
 * TOI contact's Update, the visited contacts' Updates in walk order, then PostSolve for the sub-step's island in island
 * order. They come after the Collide / Solve callbacks of the step (b2hip_get_contact_events, PreSolve, b2hip_get_post_solve).
 * One record per call site; `kind` says which callbacks it stands for (an Update can be BeginContact + PreSolve at once):
 *   bit0 BeginContact, bit1 EndContact, bit2 PreSolve(old_manifold), bit3 PostSolve (manifold.normal_impulse / tangent_impulse
 *   = the sub-step solver's impulses, manifold.point_count = its point count).
 * Recorded while begin / end events, a PreSolve function or PostSolve records are switched on (each kind only if its
 * callback is), and the TOI phase then runs through the serial event loop (the order IS the serial order).
 * PreSolve is the exception, because what it does to its contact - a zero return (b2Contact::SetEnabled(false)), an edited
 * material - changes the sub-step that called it (b2World.cpp:873-881, 948-954: a contact switched off keeps the sweeps of
 * its bodies and stays out of the sub-step's island): the function installed with b2hip_set_pre_solve /
 * b2hip_set_pre_solve_batch (one record per call) is called BY THE STEP for every such Update, in the reference's order,
 * each exactly once, and its answer acts where the reference's does. The device's event loop is one kernel: an answer that
 * changes its contact sends the phase back to its snapshot for another run with the answers so far (one extra run per
 * changing answer, b2hip_counters.toi_pre_solve_reruns). World edits made from such a call reach the device before the next
 * step; a call that edits the world AND changes its contact is refused (B2HIP_ERR_INVALID). bit2 therefore never shows in
 * the records read here. */
typedef struct b2hip_toi_callback
{
	int32_t kind;
	int32_t contact_index, fixture_a, fixture_b;
	b2hip_manifold old_manifold, manifold;
	b2hip_contact_material material;
} b2hip_toi_callback;
int b2hip_get_toi_callbacks(b2hip_world* w, int cap, b2hip_toi_callback* out);

/* b2ContactListener::PostSolve (generation b2Island.cpp:532-570, delivery b2ContactManager.cpp:454-470): the impulses the
 * solver ended with, one record per contact constraint of every island solved in the last step, in proxy-id-pair order.
 * `count` is the solver's point count (1 when the block solver's conditioning guard dropped the second point,
 * b2ContactSolver.cpp:230-247). The sub-step islands of continuous collision are not reported. */
typedef struct b2hip_contact_impulse
{
	int32_t fixture_a, fixture_b;
	int32_t contact_index;
	int32_t count;
	float normal_impulses[2];
	float tangent_impulses[2];
} b2hip_contact_impulse;
int b2hip_enable_post_solve(b2hip_world* w, int enable);
/* returns the number of records of the last step (negative on error); at most `cap` are written */
int b2hip_get_post_solve(b2hip_world* w, int cap, b2hip_contact_impulse* out);

/* ---- One world over the GPUs of a node, sharded by island ----------------------------------------------
 * Every rank builds the SAME world with the same calls and steps it; Collide, the island build, the broad-phase and the TOI
 * phase run replicated, b2Island::Solve runs for the islands a rank owns (b2World.cpp:1236-1241: islands share only static
 * bodies), and after the solve the ranks all-gather the records of what they own (a "slab" per rank: 52 B per body, 20 B per
 * contact, 24 B per joint of ITS islands; every record carries its id).
 *
 * (1) Inside the library, RCCL over xGMI on the world's own stream - no host synchronisation, nothing to call per step:
 *       rank 0: b2hip_shard_unique_id(id)  ->  hand the 128 bytes to every rank (any means: a file, MPI, torch.distributed)
 *       every rank: b2hip_shard_connect(world, id, rank, count)   (one GPU per rank, at most 8 ranks; librccl is opened here)
 *       ... b2hip_step as usual (the drop-in b2World::Step needs no change).
 * (2) With a collective of the caller's (the CPU tests use gloo through torch.distributed), per step, inside
 *     b2hip_step_begin .. b2hip_step_end, between b2hip_solve and b2hip_sync_fixtures:
 *       b2hip_set_shard(world, rank, count) once;
 *       b2hip_shard_slab_words(world, words[count], count): the slab sizes of ALL ranks (every rank knows them: the island
 *         build is replicated) -> stride = max; b2hip_shard_export(slab, stride): this rank's slab (memory the library can
 *         address: DEVICE memory for libb2hip); all-gather of `stride` int32 per rank;
 *         b2hip_shard_import(all_slabs, stride): the other ranks' results enter the world.
 * box2d-mt_amd/python/sharding.py (ShardedWorld) drives either. With one rank (the default) nothing changes. */
int b2hip_set_shard(b2hip_world* w, int rank, int count);
int b2hip_shard_slab_words(b2hip_world* w, size_t* words_per_rank, int ranks);
int b2hip_shard_export(b2hip_world* w, void* slab, size_t words);
int b2hip_shard_import(b2hip_world* w, const void* all_slabs, size_t stride_words);
int b2hip_shard_unique_id(void* id128);
int b2hip_shard_connect(b2hip_world* w, const void* id128, int rank, int count);
/* bytes the last step's all-gather moved into this rank (count x stride x 4) */
int b2hip_shard_exchange_bytes(b2hip_world* w, size_t* bytes);

/* ---- ... or by SPATIAL OWNERSHIP (round 4; SURVEY.md section 8e: "persistent ownership by spatial cell -> GPU"; the reference
 * partitions the same phases over its threads: b2CollideTask b2World.cpp:100, b2BroadphaseSyncFixturesTask :120,
 * b2BroadphaseFindNewContactsTask :142, islands :1236-1241) -------------------------------------------------------------------
 * Every rank builds the SAME world with the same calls; ids mean the same everywhere. Every non-static body has one OWNER.
 * A rank evaluates manifolds, builds islands, solves, synchronises fixtures, searches pairs and runs TOI events for the
 * bodies it owns only; what it keeps of the others is the replicated tables of section 8e (pose / velocity rows, fat AABBs)
 * and the structure of the contact array (which contacts exist, in which order - so that every order the reference's
 * results depend on stays the unsharded world's). Per step it receives: the rows and fat AABBs of the bodies the others
 * moved (after Solve and after SolveTOI) and the new pairs the others found; when a new contact joins bodies of different
 * owners, the smaller side's connected component migrates (contact and joint content is shipped once).
 * Results are bit-identical to the unsharded world in every mode whose island solve is (reference-order tier, exact-order
 * mode); large islands in the coloured order stay in their parity class (the block partition is a rank's own).
 *
 *   b2hip_shard_spatial(world, rank, count, owners): between steps, once the bodies exist. owners[b2hip_body_count] = owning
 *     rank per body (static bodies: ignored), or NULL: strips of equal body count along x. The same on every rank.
 *     Bodies created later fall into the strip of their x. Contacts and joints that join bodies of different owners are
 *     resolved at the next step (the component goes to the rank that owns most of it).
 *   the collective: b2hip_shard_connect (RCCL, as above), or b2hip_set_shard_gather: `fn(user, send, bytes, recv)` must
 *     place every rank's `bytes` bytes at recv + rank * bytes on every rank (an all-gather over HOST memory; gloo in the
 *     CPU tests, a thread barrier in the one-process GPU tests) and return 0.
 *   b2hip_get_shard_stats: what this rank owns and what the exchanges moved.
 * Not supported in a spatially sharded world (refused with B2HIP_ERR_UNSUPPORTED): contact listeners / filters, sub-stepping. */
typedef int (*b2hip_all_gather_fn)(void* user, const void* send, size_t bytes, void* recv);
typedef struct b2hip_shard_stats
{
	int32_t rank, count;
	int32_t owned_bodies, owned_proxies;     /* non-static bodies this rank owns, and their proxies */
	int32_t owned_contacts;                  /* contacts whose content this rank maintains (the others are structure only) */
	int32_t islands_solved;                  /* islands of the last step (this rank's) */
	int32_t constraint_rows;                 /* solid contacts in this rank's islands in the last step */
	int64_t migrated_bodies;                 /* bodies that changed owner since the world was sharded */
	int64_t resolutions;                     /* times a straddling contact / joint made components merge */
	int64_t bytes_received_last_step;        /* all exchanges of the last step */
	int64_t pairs_sent;                      /* new pairs this rank found and sent, since the world was sharded */
	int64_t toi_redos;                       /* TOI phases run again because an event reached over an ownership boundary */
} b2hip_shard_stats;
int b2hip_shard_spatial(b2hip_world* w, int rank, int count, const uint8_t* owners);
int b2hip_set_shard_gather(b2hip_world* w, b2hip_all_gather_fn fn, void* user);
int b2hip_get_shard_stats(b2hip_world* w, b2hip_shard_stats* out);
/* Measurement hook (tools/gpu_spatial_share.py): mode 1 - keep the result of every collective of `w` in device memory; mode 2 -
 * `w` takes its collectives' results from the tape of `from` (a world of the same rank that recorded the same run) instead
 * of running a collective: one rank of a sharded world stepped alone on one GPU; mode 0 - off. */
int b2hip_shard_tape(b2hip_world* w, int mode, b2hip_world* from);
/* Lean exchange (the default): the rows of the bodies THIS rank owns as the last step left them, packed - what the step
 * brought to the host (1 / N of the table crosses PCIe; the table of all rows stays on the device until b2hip_get_body_states
 * asks for it, as with b2hip_set_lazy_readback). ids[k] = body, out[k] = its state; returns the number of rows (at most cap). */
int b2hip_get_own_body_states(b2hip_world* w, int cap, int32_t* ids, b2hip_body_state* out);
/* the owner table as it stands (owners[b2hip_body_count]); between steps */
int b2hip_get_body_owners(b2hip_world* w, int cap, uint8_t* owners);

/* Island label per body for the last step: -1 = not solved (asleep / static), else the smallest body id
 * of the island (island membership is compared as a set partition). */
int b2hip_get_island_labels(b2hip_world* w, int cap, int32_t* out);
/* Fat AABB of a fixture's proxy (b2BroadPhase::GetFatAABB). */
int b2hip_get_fat_aabb(b2hip_world* w, int fixture, float out4[4]);
/* The fat AABBs of fixtures [first, first + count) in one copy: what b2World::QueryAABB / RayCast walk
 * (b2World.cpp:1740-1795; the reference asks its dynamic tree, b2DynamicTree.h:168-287 - same boxes, other order). */
int b2hip_get_fat_aabbs(b2hip_world* w, int first, int count, float* out4n);

/* Test hook: FNV-1a hash of a group of device arrays (0 bodies, 1 contacts, 2 proxy AABBs, 3 contact impulses),
 * usable between phase calls to compare two worlds phase by phase. */
int b2hip_debug_hash(b2hip_world* w, int which, uint64_t* out);
/* With B2HIP_TRACE=1 in the environment at world creation, b2hip_solve records a (stage label, state hash)
 * per solver stage; returns 1 past the end. */
int b2hip_debug_trace(b2hip_world* w, int index, char* label, int label_cap, uint64_t* hash);
/* Raw read-back of one device array (see b2hip.hip for the ids); test / debugging only. */
int b2hip_debug_read(b2hip_world* w, int which, int first, int count, void* out);

/* World snapshot (checkpoint / resume; the reference only has the lossy text b2World::Dump, b2World.cpp:2107-2164).
 * Everything that survives a step: bodies, shapes, fixtures with their proxy ids and fat AABBs, joints with their
 * accumulated impulses, the contact array in creation order with manifolds, warm-start impulses, cached impacts,
 * colours and TOI slots, the move buffer and the counters. A world loaded from it continues bit for bit like the world
 * it was taken from. The blob is only meaningful to the same build of the library.
 *   b2hip_save_snapshot: writes at most `cap` bytes; *needed receives the snapshot's size (call with cap 0 to ask).
 *   b2hip_load_snapshot: creates a NEW world (destroy it with b2hip_world_destroy); `device` as in b2hip_world_def. */
int b2hip_save_snapshot(b2hip_world* w, void* buffer, size_t cap, size_t* needed);
int b2hip_load_snapshot(const void* buffer, size_t size, int device, b2hip_world** out);

/* The state read-back of a step (40 B per body, the reference's b2Body members the user reads between steps, b2Body.h:126-131
 * GetTransform / GetPosition / GetAngle / velocities). Default (0): every step ends with it - b2hip_step returns with the rows on
 * the host. Lazy (1): a step ends with the counters only and the rows stay in HBM until the first call that needs a body's
 * state (b2hip_get_body_states, an edit of a body, a snapshot) fetches them, once; a caller that steps several times between
 * looks - or never looks, a batch run that reads the last state only - does not pay 40 MB of PCIe per step at 1 M bodies.
 * Results are the same bit for bit. B2HIP_LAZY_READBACK=1 in the environment sets it for every world created afterwards. */
int b2hip_set_lazy_readback(b2hip_world* w, int enable);

/* 13 floats in b2Profile declaration order (b2TimeStep.h:25-40), milliseconds, from HIP events. */
int b2hip_get_profile(b2hip_world* w, float ms[13]);
int b2hip_get_counters(b2hip_world* w, b2hip_counters* out);

/* Measurement hooks used by bench.py: mean duration (ms) of the solver kernel launches of the last
 * step measured with HIP events on the world's stream, and the algorithmic bytes they processed
 * (Ct*(Nv*220 + Np*136 + 488) + B*240). */
int b2hip_get_solver_timing(b2hip_world* w, float* ms, double* algorithmic_bytes, int* constraints, int* bodies);

/* Per-launch timing of the dominant solver kernel (k_large_velocity for coloured large islands, k_solve_small
 * for in-LDS small islands): when enabled, every launch of it is bracketed by a HIP event pair on the world's
 * stream. b2hip_get_kernel_timing reports, for the last step, the kernel's name, the SUM of its launch
 * durations (ms), the launch count and the algorithmic bytes those launches processed (220 B per constraint
 * per velocity sweep, resp. the whole 8d formula for the fused small-island kernel). Off by default: the event
 * pairs cost host time, so the timed region of bench.py runs without them. */
/* `enable`: 0 off, 1 the dominant solver kernel (above); 2 k_collide, 3 k_sync_fixtures, 4 k_find_pairs_small - the bandwidth
 * kernels of the other phases (one event pair per launch). Their algorithmic bytes are the SURVEY 8d per-unit figures times
 * the units the caller states with b2hip_set_kernel_timing_units (it knows the scene: collide = 480 B x contacts between two
 * polygons (units_a) + 230 B x other contacts (units_b); sync fixtures = 250 B x proxies (units_a); pair search = 16 B x proxies
 * (units_a) + 8 B x candidate pairs (units_b)). */
int b2hip_set_kernel_timing(b2hip_world* w, int enable);
int b2hip_set_kernel_timing_units(b2hip_world* w, long long units_a, long long units_b);
int b2hip_get_kernel_timing(b2hip_world* w, char* name, int name_cap, float* total_ms, int* launches, double* algorithmic_bytes);

#ifdef __cplusplus
}
#endif

#endif
