// Drop-in header: b2World (reference: Box2D/Dynamics/b2World.h:46-470).
//
// Same construction, body / joint factory and Step(dt, velocityIterations, positionIterations,
// b2TaskExecutor&) signature as the reference (b2World.h:105-108). Step() runs the whole hot path
// on one MI355X through the C ABI of libb2hip.so (include/b2hip.h) and returns with every body's
// pose / velocity / awake state readable on the host. There is no CPU implementation of Step here:
// if no HIP device is present the constructor reports it and Step does nothing but assert.
#ifndef B2_WORLD_H
#define B2_WORLD_H

#include <atomic>
#include <utility>
#include <mutex>
#include "Box2D/Common/b2Math.h"
#include "Box2D/Common/b2Draw.h"
#include "Box2D/Common/b2BlockAllocator.h"
#include "Box2D/Common/b2StackAllocator.h"
#include "Box2D/Dynamics/b2TimeStep.h"
#include "Box2D/Dynamics/b2WorldCallbacks.h"
#include "Box2D/MT/b2TaskExecutor.h"
#include "Box2D/Collision/b2DynamicTree.h"
#include "Box2D/Dynamics/b2ContactManager.h"

#include <vector>
#include <stdint.h>

struct b2AABB;
struct b2BodyDef;
struct b2JointDef;
class b2Body;
class b2Fixture;
class b2Joint;
class b2Contact;
struct b2hip_world;
struct b2hip_body_state;

class b2World
{
public:
	b2World(const b2Vec2& gravity);
	~b2World();

	void SetDestructionListener(b2DestructionListener* listener) { m_destructionListener = listener; }
	/// A user filter replaces the built-in category / mask / group rule (b2World.h:69-72): it is asked on the stepping
	/// thread for every new candidate pair and every contact flagged for re-filtering (b2ContactManager.cpp:283-287, 195-203).
	void SetContactFilter(b2ContactFilter* filter);
	/// BeginContact / EndContact are delivered at the end of Step() (one net event per contact and step, begins before
	/// ends, each in proxy-id order); PreSolve between Collide and the island solve (SetEnabled(false) is honoured);
	/// PostSolve at the end of Step() with the solver's final impulses. All on the stepping thread, threadId 0.
	void SetContactListener(b2ContactListener* listener);
	/// Debug drawing (b2World.h:77, 120 of the reference; what a Testbed binary calls every frame): DrawDebugData walks the
	/// host mirrors - bodies, fixtures' shape records, joints' anchors, the fat AABBs as the device holds them - and calls the
	/// user's b2Draw; nothing of it touches the step.
	void SetDebugDraw(b2Draw* debugDraw) { m_debugDraw = debugDraw; }
	void DrawDebugData();
	/// b2World::Dump (b2World.h:248): the world as C++ statements that would build it again, through b2Log. Between steps.
	void Dump();

	b2Body* CreateBody(const b2BodyDef* def);
	/// b2World.cpp:585-670: joints (SayGoodbye), contacts, fixtures (SayGoodbye) and the body; the pointer is dead afterwards
	void DestroyBody(b2Body* body);
	/// Box2D-MT's sub-tree broad-phase knob (b2World.h:199-206): accepted and ignored (the device broad-phase is a hash grid)
	void SetSubTreeSize(float32 subTreeWidth, float32 subTreeHeight) { B2_NOT_USED(subTreeWidth); B2_NOT_USED(subTreeHeight); }
	b2Joint* CreateJoint(const b2JointDef* def);
	void DestroyJoint(b2Joint* joint);   // reference: b2World.cpp:762-846 (destroy a gear joint before the joints it couples)

	/// Take a time step: collide, solve islands, update the broad-phase - all on the device.
	void Step(float32 timeStep, int32 velocityIterations, int32 positionIterations, b2TaskExecutor& executor);

	void ClearForces();

	/// Every fixture (once per child proxy) whose fat AABB overlaps `aabb` (b2World.cpp:1751-1757); the callback returns false
	/// to stop. Served by a host-side tree over the device's fat AABBs (refitted on demand, O(log n) per query): the same SET
	/// as the reference's tree reports, in this tree's traversal order.
	void QueryAABB(b2QueryCallback* callback, const b2AABB& aabb);
	/// Ray cast with the reference's callback protocol (b2World.cpp:1785-1795, b2DynamicTree.h:203-287): return 0 to stop,
	/// a fraction to clip the ray, 1 to go on unclipped, -1 to ignore the fixture.
	void RayCast(b2RayCastCallback* callback, const b2Vec2& point1, const b2Vec2& point2);
	/// Statistics of that tree (b2World.h:199-206 of the reference): it holds the same fat AABBs as the reference's
	/// broad-phase tree but is built by this repo's insertion order, so the figures are of the same kind, not the same values.
	int32 GetTreeHeight() const;
	int32 GetTreeBalance() const;
	float32 GetTreeQuality() const;
	const b2ContactManager& GetContactManager() const { return m_contactManager; }

	b2Body* GetBodyList() { return m_bodyList; }
	const b2Body* GetBodyList() const { return m_bodyList; }
	b2Joint* GetJointList() { return m_jointList; }
	const b2Joint* GetJointList() const { return m_jointList; }
	/// Materialises the contact list from the device (one copy per call after a step).
	b2Contact* GetContactList();
	const b2Contact* GetContactList() const { return const_cast<b2World*>(this)->GetContactList(); }

	void SetAllowSleeping(bool flag);
	bool GetAllowSleeping() const { return m_allowSleep; }
	void SetWarmStarting(bool flag);
	bool GetWarmStarting() const { return m_warmStarting; }
	void SetContinuousPhysics(bool flag);
	bool GetContinuousPhysics() const { return m_continuousPhysics; }
	void SetSubStepping(bool flag);
	bool GetSubStepping() const { return m_subStepping; }

	int32 GetProxyCount() const;
	int32 GetBodyCount() const { return m_bodyCount; }
	int32 GetJointCount() const { return m_jointCount; }
	int32 GetContactCount() const;

	void SetGravity(const b2Vec2& gravity);
	b2Vec2 GetGravity() const { return m_gravity; }
	bool IsLocked() const { return m_locked; }
	void SetAutoClearForces(bool flag) { m_autoClearForces = flag; }
	bool GetAutoClearForces() const { return m_autoClearForces; }
	const b2Profile& GetProfile() const { return m_profile; }
	/// b2World.cpp:1862-1887: the origin moves to `newOrigin` (a no-op while the world is locked)
	void ShiftOrigin(const b2Vec2& newOrigin);
	/// b2World.h:244, 445-448: the time the caller spent waiting for its own locks, reported with the profile
	void SetLockingTime(float32 ms) { m_profile.locking = ms; }
	/// b2World.h:220: true while the step runs the listener's *Immediate callbacks and the contact filter on the executor's threads
	bool IsMtLocked() const { return m_mtLocked; }
	/// Cost estimates with which the reference sizes its solve tasks (b2World.h:158-168; defaults b2World.cpp:466-469). The
	/// device sizes its launches from the island census instead: the values are kept for the caller and change nothing.
	uint32 GetBodyCostScale() const { return m_bodyCost; }
	uint32 GetContactCostScale() const { return m_contactCost; }
	uint32 GetJointCostScale() const { return m_jointCost; }
	void SetBodyCostScale(uint32 bodyCost) { m_bodyCost = bodyCost; }
	void SetContactCostScale(uint32 contactCost) { m_contactCost = contactCost; }
	void SetJointCostScale(uint32 jointCost) { m_jointCost = jointCost; }
	void SetSolveTaskCostThreshold(uint32 cost) { m_solveTaskCostThreshold = cost; }
	uint32 GetSolveTaskCostThreshold() const { return m_solveTaskCostThreshold; }

	/// The C-ABI world this object wraps (include/b2hip.h).
	b2hip_world* GetDeviceWorld() { return m_hip; }

private:
	friend class b2Body;
	friend class b2Fixture;
	friend class b2Contact;
	void DeliverContactEvents();
	void DeliverPostSolve();
	void DeliverToiCallbacks();
	static int FilterTrampoline(void* user, int fixtureA, int fixtureB);
	static void FilterBatchTrampoline(void* user, int count, const int32_t* fixturePairs, int32_t* verdict);
	static void PreSolveBatchTrampoline(void* user, int count, struct b2hip_pre_solve_record* records);
	/// fn(index, threadId) for every index in [0, count): on the worker threads of the executor the running Step was given
	/// (each with its own threadId < GetThreadCount(), as the reference's *Immediate callbacks see it), inline otherwise
	void RunOnExecutor(uint32 count, void (*fn)(void* ctx, uint32 index, uint32 threadId), void* ctx);
	b2TaskExecutor* m_stepExecutor;
	static int PreSolveTrampoline(void* user, int contactIndex, int fixtureA, int fixtureB, const struct b2hip_manifold* oldManifold,
		const struct b2hip_manifold* manifold, struct b2hip_contact_material* material);
	const std::vector<b2AABB>& FatAABBs();
	friend class b2BroadPhase;
	void SyncShadowTree();             // the tree's leaves follow the device's fat AABBs (one read-back per step, on demand)
	b2DynamicTree m_shadowTree;
	std::vector<int32> m_shadowLeaf;   // device fixture id -> leaf (b2_nullNode: no proxy)
	bool m_shadowValid;
	b2ContactManager m_contactManager;

	void PushFlags();
	const b2hip_body_state& State(int32 id) const;
	void RefreshStates() const;

	b2hip_world* m_hip;
	b2Vec2 m_gravity;
	bool m_allowSleep, m_warmStarting, m_continuousPhysics, m_subStepping, m_autoClearForces, m_locked;
	bool m_mtLocked = false;
	uint32 m_bodyCost = 1, m_contactCost = 10, m_jointCost = 10, m_solveTaskCostThreshold = 100;
	b2Body* m_bodyList;
	b2Joint* m_jointList;
	int32 m_bodyCount, m_jointCount;
	std::vector<b2Body*> m_bodies;     // by device id
	std::vector<b2Fixture*> m_fixtures; // by device id
	std::vector<b2AABB> m_fatAABBs;    // one read-back per step, on demand (QueryAABB / RayCast)
	bool m_fatValid;
	b2DestructionListener* m_destructionListener;
	b2ContactFilter* m_contactFilter;
	b2ContactListener* m_contactListener;
	b2Draw* m_debugDraw = nullptr;
	void DrawShape(const b2Fixture* fixture, const b2Transform& xf, const b2Color& color);
	void DrawJoint(b2Joint* joint);
	b2Profile m_profile;
	b2BlockAllocator m_blockAllocator;
	mutable std::vector<b2hip_body_state> m_states;
	mutable std::atomic<bool> m_statesValid; // (read by user range tasks on several threads: RefreshStates locks)
	mutable std::mutex m_statesMutex;
	void TouchState(int32 id) const;
	std::vector<b2Contact> m_contactViews;
	bool m_contactsValid;
	std::vector<std::pair<int32, int32> > m_endedEarly; // fixture pairs whose EndContact was delivered by an edit between steps
	std::vector<b2ContactEdge> m_edgeViews; // b2Body::GetContactList: edges of one body, rebuilt per call
	void DestroyFixtureView(b2Fixture* f);
	void BindFixtures(b2Contact& c, int deviceFixtureA, int deviceFixtureB) const;
	void EndContactsOf(b2Body* body, b2Fixture* fixture);
};

#endif
