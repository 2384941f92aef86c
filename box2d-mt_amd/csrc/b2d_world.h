// b2d_world.h - HBM layout of one simulated world (structure of arrays) and the per-step
// device-side state block. Every kernel receives a `DW` by value (kernarg, scalar loads): the
// pointers below, plus a pointer to the mutable `DState` (counters, current contact buffer).
//
// Layout rationale (MI355X): every per-body / per-contact field group is a 16-byte vector so a
// wave reads it with one global_load_dwordx4 per lane (1 KiB per wave-instruction, fully
// coalesced when lanes walk consecutive indices). Contacts are kept dense and in creation order
// (stable compaction on destroy), because creation order is what defines the reference's
// per-body contact-list order and therefore its Gauss-Seidel order (b2ContactManager.cpp:531-553).
#ifndef B2D_WORLD_H
#define B2D_WORLD_H

#include "b2d_solver.h"
#include "b2d_joint.h"
#include "b2d_wave.h"

// body flag bits (device). Bits 0-1 hold the b2BodyType.
#define BF_TYPE_MASK 0x3u
#define BF_AWAKE 0x4u
#define BF_AUTOSLEEP 0x8u
#define BF_BULLET 0x10u
#define BF_FIXEDROT 0x20u
#define BF_ACTIVE 0x40u
#define BF_ISLAND 0x80u      // was in a solved island this step (b2Body::e_islandFlag)
#define BF_LARGE 0x100u      // its island is solved by the coloured (large island) path this step

#define BT_STATIC 0u
#define BT_KINEMATIC 1u
#define BT_DYNAMIC 2u

// contact flag bits (b2Contact.h:178-203 restated)
#define CF_TOUCHING 0x1u
#define CF_ENABLED 0x2u
#define CF_FILTER 0x4u
#define CF_TOI_CANDIDATE 0x8u
#define CF_SENSOR 0x10u      // either fixture is a sensor (cached at creation)
#define CF_DESTROY 0x20u     // marked by collide, removed by the compaction that follows
#define CF_ISLAND 0x40u      // already added to an island by the DFS
// continuous collision (b2Contact::e_toiFlag, m_toiCount; valid only inside one step's TOI phase)
#define CF_TOI 0x80u           // ContactArrays::mat.w holds a valid time of impact
#define CF_TOI_LISTED 0x100u   // already in DW::toiList
#define CF_TOI_PENDING 0x200u  // claimed for recomputation by the running TOI pass
#define CF_REPORTED 0x400u     // contact events on: the host has been told that this contact touches (k_contact_events)
#define CF_PRESOLVE 0x800u      // updated by this step's Collide, touching, not a sensor: b2ContactListener::PreSolve is due (b2Contact.cpp:283)
#define CF_VC_ONE_POINT 0x10000u // the solver's conditioning guard dropped the second manifold point this step (b2ContactSolver.cpp:230-247)
#define CF_USER_REJECT 0x20000u  // the user's contact filter refused this contact at its re-filtering (b2ContactManager.cpp:195-203)
#define CF_PRESOLVE_OFF 0x40000u  // the listener's last PreSolve switched this contact off: what a TOI sub-step assumes until it has been asked
#define CF_FOREIGN 0x80000u      // spatially sharded world: the bodies of this contact belong to another rank - the contact exists here (same
                                 // slot on every rank) but its manifold, impulses and touching bit are not maintained (b2d_kernels_spatial.h)
#define CF_TOI_COUNT_SHIFT 12  // bits 12..15: m_toiCount (0..9)
#define CF_TOI_COUNT_MASK 0xf000u
#define CF_TOI_STATE_MASK (CF_TOI | CF_TOI_LISTED | CF_TOI_PENDING | CF_TOI_COUNT_MASK)

// proxy filter1 packing: low 16 = groupIndex (int16), bit16 = sensor, bit17 = thick
#define PF_SENSOR 0x10000
#define PF_THICK 0x20000

#define SMALL_ISLAND_MAX_W 512   // an island is "small" if max(bodies, contacts, 1) <= this
#define SMALL_CHUNK_LANES 1024   // one workgroup solves one chunk of small islands (<= 1024 bodies, <= 1024 contacts)
#define TINY_ISLAND_MAX_W 128    // if every small island of the step is <= this, chunks are 256 lanes (lighter barriers)
#define SMALL_ISLAND_MAX_JOINTS 64 // a small island holds at most this many joints (one lane walks them in order); more: the coloured solver
#define TINY_CHUNK_LANES 256
#define MAX_COLORS 64
#define HUB_DEGREE 30            // a body with more solid contacts than this is a "hub" (the Tumbler's container): its constraints
                                 // are not coloured (two such bodies in contact could need deg + deg - 1 > 64 colours) but solved
                                 // one after the other by k_large_hub after the coloured constraints of every sweep
#define HUB_COLOR (MAX_COLORS - 1) // row group of the hub constraints
#define CENSUS_WG_MAX_BODIES 16384 // large-island bodies up to this many are grouped by block by k_block_census itself (one workgroup)
#define COLOR_SMALL_MAX 4096     // uncoloured constraints up to this many are coloured by one workgroup without a host round trip
#define TOI_NEW_LIST_MAX 1024     // new TOI candidates of one pair update up to this many are ranked from a list (k_toi_order_create)
#define COUNT_RANK_MAX 4096      // new-pair sets up to this size are ranked by counting, above by radix sort
#define SHARD_BIG_BODIES 4096    // islands above this size are dealt over the ranks one by one (in root-id order), smaller ones by a hash of their root
#define SHARD_BIG_MAX 1024       // ... at most this many per step (more: they fall back to the hash)
#define SHARD_MAX_RANKS 8        // GPUs of one node
#define SHARD_BODY_WORDS 13      // exchange record of a body: its id, c.xy, a, sleepTime, v.xy, w, awake, xf.p.xy, xf.q.sc
#define SHARD_CONTACT_WORDS 5    // ... of a contact: its index, the four warm-start impulses
#define SHARD_JOINT_WORDS 6      // ... of a joint: its id, impulse x, y (wheel: spring impulse), z, motor impulse, limit state
// Block partition of the large islands (b2d_kernels_solve_blocks.h): every body of a large island has a home block, one
// workgroup solves one block with its bodies in LDS. A constraint between bodies of two blocks is a CUT constraint: it owns
// a colour of the upper range, so that on every body the cut constraints come last in a sweep.
#define CUT_COLOR_BASE 32        // colours [0, 32) interior constraints, [32, 63) cut constraints, 63 = HUB_COLOR
#define BLOCK_LANES 1024         // workgroup size of k_solve_blocks = rows (constraints) a block can hold
#define ROOT_PEN_SLOTS 5          // penetration maxima of the block solver's position iterations in flight (b2d_kernels_solve_blocks.h)
#define BLOCK_MAX_BODIES 1024    // home bodies a block can hold (LDS rows)
#define MAX_BLOCKS 1024          // blocks of one partition
#define BLK_SLOT 32              // ints between two blocks' counters in blkRows / blkCursor: a 128-byte line each (atomics on words of one line queue)
#define BLOCK_TARGET_DEG 1500    // a block is closed when the contact degrees of its bodies add up to this (rows ~ half of it)

struct ContactArrays
{
	int4* ids;        // proxyA, proxyB, bodyA, bodyB  (A/B after the type-table swap of b2Contact::Create)
	uint64_t* key;    // (min proxyKey << 32) | max proxyKey : b2ContactProxyIds ordering key
	uint32_t* flags;
	float4* mat;      // friction, restitution, tangentSpeed, toi
	float4* man0;     // localNormal.xy, localPoint.xy
	float4* man1;     // points[0].localPoint.xy, points[1].localPoint.xy
	float4* imp;      // normalImpulse0, tangentImpulse0, normalImpulse1, tangentImpulse1
	int4* man3;       // id0.key, id1.key, manifold type, pointCount
	int* color;       // persistent constraint colour (large-island solver), -1 = none yet
	int* mgr;         // TOI candidates: slot in the reference's contact array (b2Contact::m_managerIndex), else -1
};

struct Counters
{
	int nContacts;       // live contacts
	int nDestroy;        // marked by collide this step
	int nTouching;
	int nMoves;          // entries of the move buffer
	int nMovesSeen;      // ... as the last pair update found it
	int candRounds[32];  // rounds of 64 candidates the pair search of that update went through (spread over 32 words; the host
	                     // picks the grid's cell - full or half the limit - for the next step from candidates per moved proxy)
	int nPairs;          // candidate pairs emitted by the pair finder
	int nNewContacts;    // unique new contacts created
	int nRoots;
	int nSIslands, nSBodies, nSContacts, nSW, nChunks;
	int nLIslands, nLBodies, nLContacts;
	int nColors, nUncolored, colorRounds;
	int nLargeProxies;
	int nLargeMoves;      // moved proxies wider than a grid cell (k_find_pairs_small lists them for k_find_pairs_large)
	int posItersLarge;
	int allLargeDone;
	int needRecolor;
	int maxSmallW;       // largest max(bodies, contacts) among the small islands of this step
	int chunkW;          // chunk granularity chosen for this step (TINY_ISLAND_MAX_W or SMALL_ISLAND_MAX_W)
	int overflow;        // bit0 contacts, bit1 pairs, bit2 colours, bit3 moves
	int nIslands;
	int nToiList;        // contacts with a cached time of impact < 1 (pending TOI events)
	int nToiEvents;      // TOI sub-steps solved this step
	int nToiCalls;       // b2TimeOfImpact evaluations this step
	int toiBase;         // contact count when the TOI adjacency was built (later contacts form the tail)
	int nToiLog;         // records in DW::toiLog this step
	int toiIncomplete;   // the TOI phase ended at its event cap (sub-stepping): the step is not complete (b2World::m_stepComplete)
	int toiOverflow;     // bit0 candidates, bit1 moves, bit2 pairs, bit3 recompute list, bit4 TOI list
	int nToiOrder;       // persistent: TOI-candidate contacts alive (b2ContactManager::m_toiCount)
	int nToiDestroy;     // TOI candidates destroyed by the running collide
	int nNewToiCand;     // TOI candidates among the contacts the running pair update creates (k_create_contacts lists them in DW::toiNewList)
	int toiUnsafe;       // the parallel TOI chains met a case only the serial event loop reproduces (bits: b2d_kernels_toi_chains.h)
	int nToiGroups;      // dynamic bodies with a pending impact
	int nToiMoved;       // proxies re-inserted by the TOI chains / components
	int nToiNewPairs;    // pairs the chains found between two moving bodies (created in event order when the chains are done)
	int nToiChainCreated; // contacts that close-out created in this step
	uint32_t cellExtBits; // float bits of the largest fat-AABB extent among the grid-sized proxies, see gridLimit()
	int nToiDomains;     // components with a pending impact
	int nToiPartial;     // pending impacts of the components that are replayed serially (DW::toiDomList)
	int toiAnyFailed;    // some component of this phase has to be replayed serially (DW::toiDomFailed has a 1): k_toi_dom_rollback has work
	int nContactsSnap, nToiOrderSnap; // contact count / TOI slot count when k_toi_snapshot was taken
	int nEvents;         // contact events of this step (DW::evKey / evInfo), see k_contact_events
	int nUncolList;      // entries of DW::uncolList (large-island constraints without a colour)
	int nCompact;        // entries of DW::compactList (constraints of the colour class under compaction this step)
	int compactClass;    // colour class whose constraints may move to a lower free colour this step (0: none)
	int compactCursor;   // persistent: the class visited last (one per step, from the highest down to 1, then over again)
	int compactIdle;     // persistent: classes visited in a row without a constraint moving
	int compactMoved;    // constraints k_color_small moved down in this step's visit (read by the next step's colorCheckBegin)
	int compactTick;     // persistent: steps, for the slow beat of an idle colouring
	int maxDegree;       // largest number of solid touching contacts on one non-static body this step
	int maxDegreePlain;  // ... among the bodies that are no hubs (<= HUB_DEGREE): the fewest colours a colouring of their constraints can have
	int gridFresh;       // the hash grid holds every proxy's fat AABB as of now: built by this step's pair update (k_bp_clear / k_bp_build), nothing has moved a proxy since (k_step_begin, k_sync_fixtures reset it) - the TOI phase then queries it as it is
	int nHubRows;        // hub constraints of this step
	int nHubWide;        // ... the first so many of hubList are constraints of the PRIMARY hub with different non-hub partners: one fixed point (k_sweep_end)
	int hubEpoch;        // persistent: steps counted for the tags of DW::hubFirst
	int colorRows[MAX_COLORS]; // census of the large-island constraints by colour as k_color_check found it (the host picks the tail colours of k_sweep_end from it)
	int hubRounds;       // fixed-point rounds k_large_hub ran this step (all sweeps, all chunks)
	int hubSerialChunks; // chunks of 64 hub rows it solved lane after lane instead
	int chunkLanes;      // workgroup size of the small-island solver chosen for this step (TINY_CHUNK_LANES or SMALL_CHUNK_LANES)
	// block partition (persistent: nBlocks, partitions; per step: the rest)
	int nBlocks;         // blocks of the current partition (0 = none yet)
	int partitions;      // partitions made so far (diagnostics)
	int nOrphanRows;     // large-island constraints with a body that has no home block this step
	int blkMaxRows;      // most rows owned by one block this step
	int blkMaxBodies;    // most home bodies in one block this step
	int nCutRows;        // cut constraints this step
	int blkTargetDeg;    // degree budget per block the current partition was made with
	int blkLanes;        // workgroup size (= rows / home bodies a block may hold) the current partition was made for
	int partitionAge;    // steps since the current partition was made (persistent; kept here so that a snapshot carries it)
	int partitionCooldown; // steps for which no new partition is attempted (the last attempts did not fit); persistent
	uint32_t colorMaskLo, colorMaskHi; // colours that own at least one large-island constraint this step
	int nPreSolve;       // PreSolve records of this step's Collide (DW::preRecs)
	int nPostSolve;      // PostSolve records of this step's Solve (DW::postRecs)
	int nFilterList;     // contacts flagged for re-filtering, listed for the user's contact filter (DW::filterList)
	int nBigIslands;     // islands with more than SHARD_BIG_BODIES bodies this step (sharded worlds only)
	// sharded worlds: what every rank's slab of the exchange holds this step - bodies, solid contacts, joints of the islands it
	// owns. Every rank counts ALL ranks' (the island build is replicated), so the hosts size the all-gather without talking.
	int shardBodies[SHARD_MAX_RANKS], shardContacts[SHARD_MAX_RANKS], shardJoints[SHARD_MAX_RANKS];
	int shardCursor[3];  // append cursors of k_shard_export
	int nRemoteIslands;  // islands of this step that another rank solves
	int nSerialOrphans;  // constraints swept in order this step because a body of theirs has no home block (rowIsSerial)
	int compactBlocksDone; // workgroups of k_compact_contacts that have finished (the last one switches the contact buffers)
	int endBlocksDone;     // tiles of 256 bodies the last marks-driven k_end_step looked at (diagnostics; its last workgroup puts it back to 0 behind the read-back)
	int rowsSkipped;       // k_end_step left the state rows out: the host will finish the pair update and read back again
	int collideBlocksDone; // ... of k_collide (the last one runs toiOrderDestroy)
	int chainBlocksDone;   // ... of k_toi_chains (the last one runs toiChainsEnd)
	int edgesBlocksDone;   // ... of k_island_edges (the last one publishes the census when it is the island build's last kernel)
	int nFreeIslands;    // one-body islands without contacts or joints, stepped by k_island_classify itself
	int nSmallJointed;   // small islands of this step that hold joints (none: the lean k_solve_small runs)
	// spatially sharded worlds (b2d_kernels_spatial.h)
	int nStraddle;       // contacts between non-static bodies of different owners (k_sp_flag_contacts): resolved before anybody updates them
	int nStraddleJoints; // ... joints
	int nResolve;        // components the running resolution merges
	int nMigrated;       // bodies whose owner the last resolution changed
	int spContacts[SHARD_MAX_RANKS], spJoints[SHARD_MAX_RANKS]; // contact / joint records every rank ships in the running resolution
	int spMigBodies[SHARD_MAX_RANKS];                           // ... body rows (the bodies that leave it)
	int spBodies[SHARD_MAX_RANKS], spProxies[SHARD_MAX_RANKS];  // non-static bodies and their proxies per owner (k_sp_owner_census)
	int spToiCreated;    // contacts this rank's TOI phase created (the tail of its contact array until the ranks have merged their tails)
	int spToiStraddle;   // ... of them with a body of another rank (an event reached over an ownership boundary: refused)
	int spOwnRows;       // rows k_end_step packed into DW::spOwnOut this step (the bodies this rank owns)
};

// What b2ContactListener::PreSolve is told about one contact (gathered after Collide, before the compaction of destroyed
// contacts: `info.x` is the contact's index AFTER it), and what PostSolve is told.
struct PreSolveRec
{
	int4 info;               // contact index, proxy A, proxy B, -
	unsigned long long key;  // proxy-id pair: the reference's deferred-callback order
	unsigned long long pad;
	float4 o0, o1, oimp;     // old manifold (as ContactArrays::man0 / man1 / imp)
	int4 o3;
	float4 n0, n1, nimp;     // new manifold
	int4 n3;
	float4 mat;              // the contact's mixed friction, restitution, tangent speed (b2Contact.h:40-50, 157) as the callback may edit them
};
// One listener call the reference makes from a TOI sub-step (b2World.cpp:866,946: contact->Update(listener); b2Island.cpp:527 ->
// Report), logged by the serial TOI loop in call order (DW::toiLog, Counters::nToiLog) while a listener is installed.
struct ToiLogRec
{
	int4 info;               // kind bits (1 BeginContact, 2 EndContact, 4 PreSolve, 8 PostSolve), contact index, proxy A, proxy B
	float4 o0, o1, oimp;     // old manifold (PreSolve)
	int4 o3;
	float4 n0, n1, nimp;     // new manifold ; PostSolve: nimp = the solver's impulses, n3.w = its point count
	int4 n3;
	float4 mat;
};
struct PostSolveRec
{
	int4 info;               // contact index, proxy A, proxy B, solver point count
	unsigned long long key;
	unsigned long long pad;
	float4 imp;              // normalImpulse0, tangentImpulse0, normalImpulse1, tangentImpulse1
};

struct DState
{
	Counters c;
	int cur;             // which ContactArrays is live
	int stamps[6];       // phase stamps of the resident large-island solver (copied from its barrier words by k_end_step)
	// wall_clock64 (100 MHz) at: end of k_block_census, start of k_island_dfs / k_color_small / k_solve_blocks - what the host's
	// census read-back in the middle of the step costs (tools/gpu_sync_gap.py)
	unsigned long long gapClock[4];
	// b2Profile: wall_clock64 at the start of the first kernel of each phase (slots = the PH_* points of b2hip.hip). Written
	// by b2dPhaseStamp, which every kernel calls first: a time stamp costs nothing on the stream (a hipEventRecord between
	// two kernels is a packet of its own - thirteen of them were 90 us of the 10 011-box pyramid's 760 us step)
	unsigned long long phaseClock[16];
	int pubCount;        // publications of the census so far (b2dPublishCensus; the host counts along)
	// (host copies only: the number of the publication / read-back this copy is; what lies before it is copied in 16-byte pieces)
	int dbgCensus[8];    // (diagnostics, B2HIP_HANDOVER_WHY: the first large-island body k_block_census found without a block, and why)
	alignas(16) int pubSeq;
	int pubPad[3];
};

// Where DState sits behind the n state rows of a read-back (floats from the start of the buffer; 16-byte aligned).
#define B2D_STATE_TAIL(n) ((((size_t)(n) * 10) + 3) & ~(size_t)3)

struct StepParams
{
	float dt, inv_dt, dtRatio;
	int velIters, posIters;
	int warmStarting, allowSleep;
	V2 gravity;
};

struct DW
{
	DState* st;
	uint32_t stampMask;   // phase stamps the next kernel on the main stream takes (b2dPhaseStamp); 0 for every other launch
	int nBodies, nProxies, nJoints, nShapes;
	int capContacts, capPairs, capMoves;
	int serialOrphans;    // 1: constraints of bodies without a home block are swept in order with the hub constraints
	int noFreeBodies;     // B2HIP_NO_FREE_BODIES=1: one-body islands go through the small-island solver like any other
	int testMaxColors, testColorRounds; // B2HIP_TEST_MAX_COLORS / B2HIP_TEST_COLOR_ROUNDS: fewer colours per range / rounds of k_color_small than the built-in limits (tests of the recoveries)
	int testSpinMax;      // B2HIP_TEST_SPIN_MAX: the spin limit of the waits between workgroups (0: PERSIST_SPIN_MAX / DATAFLOW_SPIN_MAX) - forces the time-out the recovery path exists for
	int restPoll;         // how the rest rows wait (dataflowRun's pollSleep): 1 short naps (round 5), >= 4 back off by distance (B2HIP_REST_POLL)
	int hubSerial;        // B2HIP_HUB_SERIAL=1: hub rows lane after lane only (validation of the fixed-point path)
	int smallMaxW;        // islands up to this size take the exact-order in-LDS solver (default TINY_ISLAND_MAX_W = 128; B2HIP_SMALL_MAX_W up to 512)
	int bigChunks;        // 1: always use 1024-lane chunks for the small-island solver (B2HIP_BIG_CHUNKS)
	uint32_t htMask;      // contact-key hash table size - 1
	uint32_t gridMask;    // broad-phase hash grid size - 1
	float cellSize, invCellSize;

	// ---- bodies [nBodies], index = creation order ------------------------------------------
	float4* b_pos;    // sweep.c.xy, sweep.a, sleepTime
	float4* b_pos0;   // sweep.c0.xy, sweep.a0, alpha0
	float4* b_vel;    // v.xy, w, -
	float4* b_xf;     // xf.p.xy, xf.q.s, xf.q.c
	float4* b_mass;   // invMass, invI, localCenter.xy        (constant between edits)
	float4* b_damp;   // linearDamping, angularDamping, gravityScale, -
	float4* b_force;  // force.xy, torque, -
	uint32_t* b_flags;
	int* b_wake;      // wake requests gathered during collide / contact creation
	int* b_order;     // per body: its slot in the reference's m_nonStaticBodies (b2World.cpp:573, 662-667: appended at creation,
	                  // the last one moves into the slot of a destroyed one) - island seeds are taken in that order (:1207-1221)
	int* orderBody;   // slot -> body

	// ---- proxies [nProxies], one per fixture (circle / edge / polygon) -----------------------
	float4* p_fat;    // fat AABB lower.xy, upper.xy (b2DynamicTree node aabb)
	int* p_body;
	int* p_shape;
	int* p_key;       // proxy id the reference's tree would hand out: the deterministic ordering key
	uint32_t* p_filter0; // categoryBits | maskBits << 16
	int* p_filter1;      // groupIndex | sensor / thick bits
	float2* p_mat;       // friction, restitution
	const ShapeRec* shapes;
	int* b_proxyHead;    // per body: its newest proxy (b2Body::m_fixtureList), -1 = none
	int* p_next;         // next older proxy of the same body

	// ---- contacts ---------------------------------------------------------------------------
	ContactArrays ca[2];
	uint64_t* ht_keys;   // open-addressing set of live contact keys

	// ---- joints -----------------------------------------------------------------------------
	RevoluteJoint* joints;
	GearRec* gears;      // gear joints' own records (JointRec::enableLimit indexes it)
	int* jadjStart;      // per body: its joint edges, newest first (b2World::CreateJoint pushes at the list head)
	int* jadj;
	int* rootJointStart; // per root: segment of lj_list (exclusive scan of rootJoints)
	int* rootJointCursor;
	int* lj_list;        // joint ids grouped by island, in the island's solve order
	int* rootJointOkay;  // per root: AND of the joints' position-solve results of the running iteration

	// ---- island build -----------------------------------------------------------------------
	int* parent;         // union-find over non-static bodies; after flatten: island root per body
	int* rootSeed;       // per root: lowest m_nonStaticBodies slot (b_order) among its awake bodies (INT_MAX = island asleep)
	int* rootBodies;     // per root: non-static body count
	int* rootContacts;   // per root: solid touching contact count
	int* rootJoints;
	int4* rootScanIn;    // per body slot: (nb, nc, w, 1) for small solved roots else 0
	int4* rootScanOut;
	int* rootIsland;     // per root: small-island index, or -2 for large, -1 not solved
	int* deg;            // per body: number of solid touching contacts (CSR)
	int* adjStart;       // exclusive scan of deg
	int* adjCursor;
	int* adj;            // contact indices grouped by body
	int2* adjSlot;       // per contact: its place in the adjacency segments of its two bodies (k_island_union), -1 for a static body
	// small islands
	int* si_root;        // [nSIslands]
	int* si_bodyStart;   // [nSIslands + 1]
	int* si_contactStart;
	int* si_wStart;
	int* si_maxLevel;
	int* si_bodies;      // body ids in the reference's DFS order, island after island
	int* si_contacts;    // contact indices in the reference's discovery order
	int* si_level;       // dependency level of each contact inside its island (1-based)
	int* si_stack;       // DFS stack scratch (same segmentation as si_bodies)
	int* si_lastLevel;   // per island-body slot scratch
	int* b_slot;         // per body: slot in si_bodies (small islands)
	int* b_island;       // per body: small island index
	int* chunkFirst;     // [nChunks] first small island of each solver chunk
	// large islands
	int* li_bodies;      // body ids (any order)
	int* li_contacts;    // contact indices (any order)
	int* li_roots;
	int* li_color;       // per large contact slot
	int* colorCount;     // [MAX_COLORS + 1]
	int* colorStart;     // [MAX_COLORS + 1]
	int* colorCursor;
	int* compactList;    // large contact slots of colour Counters::compactClass (at most COLOR_SMALL_MAX listed)
	int* uncolList;      // large contact slots that have no colour yet (at most COLOR_SMALL_MAX listed)
	int* hubRowOf;       // per contact: its constraint row if it is a hub constraint this step
	float4* hubDelta;    // per hub constraint (hubList order): the change it made to its hub's row in the last sweep (k_large_hub's first guess)
	int* hubList;        // hub constraint rows in contact-index order (the deterministic visiting order of k_large_hub)
	unsigned long long* hubMeta;  // [0] the primary hub of this step: (its degree << 32) | body id, 0 = no hub (k_island_flatten)
	unsigned long long* hubFirst; // per body: (Counters::hubEpoch << 32) | ~(lowest contact index among its constraints with the primary hub)
	int hubWide;         // 1: hubList starts with the rows k_sweep_end takes as one fixed point (Counters::nHubWide); 0: k_large_hub's order (B2HIP_HUB_WIDE=0)
	int* li_sorted;      // large contact slots grouped by colour
	int4* li_ref;        // per colour-sorted row: contact index, bodyA, bodyB (static bodies as -(id+1)), island root
	uint32_t* bodyClaim;
	uint64_t* bodyColorMask;
	uint64_t* bodyActive;   // per body: colours of its constraints in THIS step's large-island solve (dataflow solver)
	float4* solveSnapBody;  // what k_solver_snapshot keeps of the large islands' bodies (6 rows each), ...
	float4* solveSnapImp; uint32_t* solveSnapCFlags; // ... of their contacts (impulses, flags) ...
	struct JointRec* solveSnapJoints; struct GearRec* solveSnapGears; // ... and of their joints
	uint64_t* bodyRest;     // per body: the REST colours (>= k_color_fill's restFirst) among its constraints of this step (k_large_rest)
	int eventsOn;           // contact events requested by the host (b2hip_enable_contact_events)
	unsigned long long* evKey; // per event: proxy-key pair of the contact (the reference's deferred-callback sort key)
	int4* evInfo;           // per event: (fixtureA, fixtureB, kind 0 = begin / 1 = end, contact index or -1 if destroyed)
	int* dfRank;            // per body: DF_RANKS mailbox slots, [rank] = slot of the body's rank-th constraint (k_solve_mailbox)
	float4* dfInbox;        // per large-island constraint row: two tagged 16-byte slots (body A, body B)
	float4* b_posv;         // per body: (c.xy, a, version) rows of the dataflow solver's position phase
	float* lc;           // large-island constraint rows, field-major: lc[field * capContacts + slot]
	int* b_rowDirty;        // per body: its read-back row (or its sweep's alpha0) was written behind SynchronizeFixtures - by contact creation's wake-ups or the TOI phase; k_end_step looks at these bodies only when the rows left early (b2hip_host_phases.h: startEarlyRows) and clears the marks
	float4* warmDelta;   // per large-island constraint row: 4 x float4 - what its warm start subtracts from body A (points 0, 1) and adds to body B (k_large_init -> k_large_warm)
	uint32_t* rootPen;   // per root: max penetration of the running position iteration (bits of -minSeparation)
	int* rootDone;       // per root: positionSolved
	uint32_t* rootSleepMin;
	// block partition of the large islands
	int* b_blk1;         // per body, persistent: home block + 1 (0 = none)
	int* b_adopt;        // per body, per step: block + 1 offered to a body without one by a neighbour (max over the neighbours)
	int* b_adoptStage;   // three stage buffers of k_block_adopt (nBodies each)
	int* blkRows;        // per block: rows it owns this step
	int* blkRowStart;    // [nBlocks + 1] exclusive scan of blkRows
	int* blkCursor;      // per block: fill cursor of k_color_fill
	int* blkBodyCount;   // home bodies per block (counted by k_color_check, scanned by k_block_census); [block * BLK_SLOT], like blkRows
	int* blkBodyCursor;  // ... slots handed out so far (k_color_fill)
	int* blkBodyStart;   // [nBlocks + 1] home bodies of each block (segments of blkBodies)
	int* blkBodies;      // large-island bodies grouped by home block
	int* rowColor;       // per block-sorted row: its colour
	float4* b_cutv;      // per body: (v.xy, w, tag) exchange row of the cut constraints, velocity phase (positions: b_posv)
	int blockSort;       // k_color_fill groups the rows by owner block (k_solve_blocks) instead of by colour
	// island sharding over the ranks of one node (b2d_kernels_shard.h): every rank holds the whole world, solves the islands
	// it owns and takes the others' results from one exchange per step
	int shardRank, shardCount;
	int* bigRoots;       // roots of the islands with more than SHARD_BIG_BODIES bodies (dealt over the ranks in root-id order)
	// ... or by spatial ownership (b2d_kernels_spatial.h): work and contact content partitioned by the owner of the bodies
	int spatial;         // 1: DW::b_owner decides who evaluates, solves and moves a body and its contacts
	uint8_t* b_owner;    // per body: the rank that owns it (the same table on every rank; static bodies: unused)
	uint8_t* spNewOwner; // per body: its owner after the running resolution
	uint8_t* spAwake;    // per body: the awake bit the other ranks hold for it (lean exchange: a row travels when it changes)
	int spFullRows;      // 1: E1 / E4 carry the rows of every body a rank moved (every rank holds the whole world's state: tests,
	                     // callers that read any body anywhere); 0: only what the others' work needs - fat AABBs, awake bits
	int* spStraddle;     // contact indices of Counters::nStraddle
	int capStraddle;
	int* spCount;        // per component under resolution: bodies per owner [SP_RESOLVE_MAX][SHARD_MAX_RANKS]
	int* spTarget;       // per component under resolution: its new owner
	int* spOwnOut;       // pinned host memory: the rows of the bodies this rank owns, packed (11 words: id + b2hip_body_state), lean exchange
	int spOwnCap;
	int4* spTailKey;     // per contact created inside a TOI phase: (alpha bits, event key hi, lo, -) of the event that created it -
	                     // the reference's creation order across ranks (k_sp_merge_tails)
	// listener / filter bridge (include/b2hip.h: b2hip_set_contact_filter, b2hip_set_pre_solve, b2hip_enable_post_solve)
	int userFilter;      // a user contact filter is installed: the built-in category / mask / group rule is not applied
	int preSolveOn, postSolveOn;
	float4* pre_o0;      // per contact: the manifold before this step's Collide (valid where CF_PRESOLVE is set)
	float4* pre_o1;
	float4* pre_oimp;
	int4* pre_o3;
	PreSolveRec* preRecs;
	PostSolveRec* postRecs;
	ToiLogRec* toiLog;      // listener calls from TOI sub-steps, in call order (null: no listener)
	int capToiLog;
	int toiEventCap;     // b2World::SetSubStepping: events per call (0: no cap)
	int toiContinue;     // this call continues a step an earlier call left incomplete (no island solve, no first pass)
	const int4* toiVerdict; // what the listener's PreSolve answered for the Updates logged so far in this phase: x bit0 asked,
	int nToiVerdict;        // bit1 switched the contact off, yzw friction / restitution / tangent speed (bits); [0, nToiVerdict)
	int* filterList;     // contact indices flagged CF_FILTER (listed for the user's filter before Collide)

	// ---- broad-phase ------------------------------------------------------------------------
	int* moveBuf;        // proxy indices whose fat AABB changed / were created
	int* gridCount;      // per hash cell
	int* gridStart;
	int* gridCursor;
	unsigned long long* arriveTree; // ARRIVE_SITES two-level arrival trees (b2dTreeArrive), all words 0 between launches
	int* gridItems;      // proxy indices grouped by cell
	float4* gridFat;     // ... and their fat AABBs as k_grid_fill found them (the pair search's copy: the TOI phase moves p_fat)
	int* largeProxies;   // proxies larger than a cell
	int* largeMoves;     // ... those of them in the move buffer
	uint64_t* pairKey;   // candidate pairs: key
	int2* pairProxy;     // proxy indices (lo-key proxy, hi-key proxy)
	uint64_t* pairKey2;  // sort double buffer
	int2* pairProxy2;
	int* pairFirst;      // 1 if first occurrence of its key
	int* pairRank;       // rank among unique keys

	// ---- continuous collision ----------------------------------------------------------------
	int* toiList;        // contact indices whose cached TOI is < 1
	int* toiPos2c;       // slot of the reference's TOI partition -> contact index (inverse of ContactArrays::mgr)
	int* toiDestroyList; // TOI candidates marked for destruction by collide
	int* toiNewList;     // TOI candidates among the new contacts of the running pair update (contact indices, at most TOI_NEW_LIST_MAX listed)
	int* b_toiGroup;     // per body: chain index + 1 while it owns a TOI chain
	int* toiGroups;      // dynamic bodies with a pending impact
	// TOI components (b2d_kernels_toi_domains.h): connected components of {non-static bodies, contacts between them}
	int* toiParent;      // per body: component label (union-find, flattened)
	int* toiDomOf;       // per body (label): component index + 1 while the component has a pending impact, else 0
	int* toiDomRoot;     // per component: its label
	int* toiDomCount;    // per component: contacts in it = capacity of its pending list
	int* toiDomBase;     // per component: start of its slice of toiDomList
	int* toiDomFill;     // per component: pending impacts listed so far
	int* toiDomList;     // pending lists of all components, back to back; afterwards the list of the serial replay
	int* toiDomFailed;   // per component: it met a new contact / another component and has to be replayed serially
	int* toiDomEvents;   // per component: events it counted (taken back if it is replayed)
	float4* toiHull;     // per proxy: hull of the fat AABBs it has had in this TOI phase (valid for proxies in toiMoved)
	int* toiGroupCount;  // per chain: contacts gathered for it
	int* toiGroupList;   // per chain: CHAIN_ADJ_MAX contact indices
	int* toiMoved;       // proxies re-inserted by the chains
	int gridHalf;        // the grid's cell is half the limit (dense scenes) instead of the limit itself (b2d_kernels_broadphase.h: gridCell)
	int noOwnIdBlocks;   // B2HIP_NO_OWN_ID_BLOCKS=1: a large-island body without a block or an offer stays an orphan (round 4: the island is partitioned again)
	int noChainCreate;   // B2HIP_TOI_NO_CHAIN_CREATE=1: every new pair a chain meets sends the phase to the serial loop (comparison)
	int* toiNew;         // TOI_NEWPAIR_MAX x 8 ints: pairs found by the chains (alpha bits, event key hi / lo, proxy lo / hi)
	float4* snapBody;    // 5 rows per body: pos, pos0, vel, xf, flags (state before the chains)
	float4* snapFat;     // fat AABBs before the chains

	// ---- generic scratch ----------------------------------------------------------------------
	int* scanTmp;        // block sums for the scan utility
	int* radixHist;
	int* keepFlag;       // contact compaction
	int* keepScan;

	// ---- read-back ------------------------------------------------------------------------------
	float* stateOut;     // 10 x 4 bytes per body (b2hip_body_state)
};

#if defined(__HIPCC__)
// "The host will keep the block partition this step": what phaseSolve's test for a new partition comes to on the census
// (slightly stricter: it does not know the workgroup size the host would choose). k_color_small, queued behind the census
// before the host has seen it, only runs if this holds - colours handed out against a partition that is about to be
// replaced would differ from the ones the host-ordered sequence (partition, check, colour) gives - and the host, which
// evaluates the same function on the same counters, knows whether it ran.
__host__ __device__ inline bool b2dPartitionSettled(const Counters& c)
{
	if (c.nBlocks == 0 || c.nOrphanRows > 0 || c.blkMaxRows > c.blkLanes || c.blkMaxBodies > c.blkLanes || c.nSerialOrphans > 2048) return false;
	if (c.partitionAge > 240 && (4 * c.nCutRows > c.nLContacts || 2 * c.nLContacts < 900)) return false;
	return true;
}

// What one workgroup hands to another workgroup of the SAME launch (the "last workgroup finishes the job" kernels): stores
// and loads that go past the XCD's L2 (sc1), so that no fence is needed - a release fence at agent scope writes the whole
// L2 back (measured: a __threadfence per workgroup made k_collide 50 us longer on the 10 011-box pyramid).
//   producer: b2dStoreAgent* ... then b2dLastBlockArrive() ;  the workgroup it elects reads with b2dLoadAgent*
__device__ __forceinline__ void b2dStoreAgentI(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int b2dLoadAgentI(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void b2dStoreAgent4(float4* p, float4 v)
{
	typedef float f4 __attribute__((ext_vector_type(4)));
	f4 q;
	q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
	// (s_nop 1: the store's data registers must not be written by the next two instructions - see stRow in b2d_handover.h)
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ float4 b2dLoadAgent4(const float4* p)
{
	typedef float f4 __attribute__((ext_vector_type(4)));
	f4 r;
	asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
	return make_float4(r.x, r.y, r.z, r.w);
}
// ---- arrival of the workgroups of a launch ------------------------------------------------------------------------------
// "The workgroup that finishes last does the close-out" and "every workgroup adds its count to a counter" both need one
// atomic per workgroup on ONE word. On this part such atomics are executed at the memory side (the eight XCDs' L2s are not
// coherent with one another), one after the other, ~30 ns each: measured with k_island_classify on a million bodies - 78 us
// without its two census adds, 88 us with 512 workgroups, 205 us with 2 048, 325 us with 8 192. So the workgroups arrive in
// two levels: workgroup b at slot b % 32 (32 words on 32 different 128-byte lines: 32 queues side by side), and the one that
// completes its slot's count carries the slot's totals on to the root word. The longest queue is gridDim / 32 + 32 atomics
// instead of gridDim. One 64-bit word holds the count (bits 48+) and two 24-bit sums, so a census travels with the arrival;
// whoever completes a word puts it back to 0 - nobody else touches it again in this launch.
#define TREE_GROUPS 32
#define TREE_STRIDE 16 // 64-bit words between two slots: 128 bytes
#define TREE_WORDS ((TREE_GROUPS + 1) * TREE_STRIDE)
#define TREE_SUM_MAX 0xffffffu // (callers whose sums may exceed this use plain atomics instead: see treeSumsFit)
#define ARRIVE_COLLIDE 0
#define ARRIVE_COMPACT 1
#define ARRIVE_EDGES 2
#define ARRIVE_CHAINS 3
#define ARRIVE_END_STEP 4
#define ARRIVE_CLASSIFY 5
#define ARRIVE_TOI_FIRST 6
#define ARRIVE_COLOR_CHECK 7
#define ARRIVE_PAIRS 8
#define ARRIVE_SITES 9
// ONE thread per workgroup, every workgroup of the (one-dimensional) grid exactly once. True in the workgroup that arrives
// last, with the sums of all workgroups' v0 / v1.
__device__ __forceinline__ bool b2dTreeArrive(unsigned long long* tree, unsigned v0, unsigned v1, unsigned* t0, unsigned* t1)
{
	const unsigned nb = gridDim.x;
	const unsigned groups = nb < TREE_GROUPS ? nb : TREE_GROUPS;
	const unsigned g = blockIdx.x % groups;
	const unsigned members = nb / groups + (g < nb % groups ? 1u : 0u);
	const unsigned long long one = 1ull << 48;
	unsigned long long* slot = tree + (size_t)g * TREE_STRIDE;
	const unsigned long long add = one | ((unsigned long long)(v1 & TREE_SUM_MAX) << 24) | (unsigned long long)(v0 & TREE_SUM_MAX);
	unsigned long long seen = __hip_atomic_fetch_add(slot, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
	if ((unsigned)(seen >> 48) != members) return false;
	__hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	unsigned long long* root = tree + (size_t)TREE_GROUPS * TREE_STRIDE;
	const unsigned long long add2 = one | (seen & 0xffffffffffffull);
	seen = __hip_atomic_fetch_add(root, add2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add2;
	if ((unsigned)(seen >> 48) != groups) return false;
	__hip_atomic_store(root, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	*t0 = (unsigned)(seen & TREE_SUM_MAX);
	*t1 = (unsigned)((seen >> 24) & TREE_SUM_MAX);
	return true;
}

// Every thread of the workgroup calls it at the end of its work; true in the workgroup that arrived last (all of them
// have then completed the stores they made before arriving). v0 / v1: this WORKGROUP's contribution to two sums (the value
// thread 0 passes counts), delivered in *t0 / *t1 to the elected workgroup (all its threads).
__device__ __forceinline__ bool b2dLastBlockArrive(const DW& W, int site, unsigned v0 = 0u, unsigned v1 = 0u, unsigned* t0 = nullptr, unsigned* t1 = nullptr)
{
	__shared__ int s_lastBlock;
	__shared__ unsigned s_treeSums[2];
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0)
	{
		unsigned a = 0u, b = 0u;
		s_lastBlock = b2dTreeArrive(W.arriveTree + (size_t)site * TREE_WORDS, v0, v1, &a, &b) ? 1 : 0;
		s_treeSums[0] = a;
		s_treeSums[1] = b;
	}
	__syncthreads();
	if (t0) *t0 = s_treeSums[0];
	if (t1) *t1 = s_treeSums[1];
	return s_lastBlock != 0;
}

// Two counters every lane of a kernel adds to at its end (a census): summed over the workgroup, carried by the arrival tree,
// added to the counters by the workgroup that arrives last. Nothing else may write the two counters in this launch. Every
// lane of every workgroup calls. fit = the sums stay below 2^24 (the caller's bound: bodies, contacts); else plain atomics.
__device__ __forceinline__ void b2dBlockTreeAdd2(const DW& W, int site, int* c0, int v0, int* c1, int v1, bool fit)
{
	__shared__ int s_treeAdd[2];
	if (threadIdx.x == 0) { s_treeAdd[0] = 0; s_treeAdd[1] = 0; }
	__syncthreads();
	v0 = waveSumInt(v0);
	v1 = waveSumInt(v1);
	if (waveLane() == 0)
	{
		if (v0) atomicAdd(&s_treeAdd[0], v0);
		if (v1) atomicAdd(&s_treeAdd[1], v1);
	}
	__syncthreads();
	if (threadIdx.x != 0) return;
	if (!fit)
	{
		if (s_treeAdd[0]) atomicAdd(c0, s_treeAdd[0]);
		if (s_treeAdd[1]) atomicAdd(c1, s_treeAdd[1]);
		return;
	}
	unsigned t0 = 0u, t1 = 0u;
	if (b2dTreeArrive(W.arriveTree + (size_t)site * TREE_WORDS, (unsigned)s_treeAdd[0], (unsigned)s_treeAdd[1], &t0, &t1))
	{
		if (t0) atomicAdd(c0, (int)t0);
		if (t1) atomicAdd(c1, (int)t1);
	}
}

// The island census for the host, which is polling for it (b2hip.hip: awaitCensus): every counter goes straight into the
// pinned host copy `pub`, the number of this publication last. Called by ALL threads of one workgroup, after everything
// the counters depend on (the island build's last kernel, its last workgroup). Cheaper than a copy behind the kernel plus
// a stream synchronisation, and the stream can go on (k_color_small is already queued) while the host decides.
// `seqGiven` >= 0: the publication's number comes from the host (a SECOND buffer with a count of its own: the state behind
// k_color_small - two publications into one buffer within a few microseconds raced with the host, which polls for the
// first one's number and copies the buffer: round 6, a step in a few hundred failed with "census was not published").
__device__ __forceinline__ void b2dPublishCensus(const DW& W, DState* pub, int seqGiven = -1)
{
	DState* S = W.st;
	// (what lane 0 has just stored into the counters with ordinary stores is written back before anybody reads it past the L2)
	if (threadIdx.x == 0) __threadfence();
	__syncthreads();
	// (16 bytes per lane: the whole block leaves as one or two store instructions of the first wave - word by word it was
	// two hundred separate writes across PCIe)
	static_assert(offsetof(DState, pubSeq) % 16 == 0, "DState: the published part is copied in 16-byte pieces");
	const float4* src = (const float4*)S;
	float4* dst = (float4*)pub;
	for (int k = (int)threadIdx.x; k < (int)(offsetof(DState, pubSeq) / 16); k += (int)blockDim.x) dst[k] = b2dLoadAgent4(&src[k]);
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0)
	{
		int seq = seqGiven;
		if (seqGiven < 0)
		{
			seq = (S->pubCount + 1) & 0x3fffffff;
			S->pubCount = seq;
		}
		__hip_atomic_store(&pub->pubSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// First statement of every kernel: the start of this kernel is the boundary of the phases the host named in stampMask.
__device__ __forceinline__ void b2dPhaseStamp(const DW& W)
{
	if (W.stampMask != 0u && blockIdx.x == 0 && threadIdx.x == 0)
	{
		const unsigned long long t = wall_clock64();
		for (uint32_t m = W.stampMask; m != 0u; m &= m - 1u) W.st->phaseClock[__ffs((int)m) - 1] = t;
	}
}
#endif

#endif
