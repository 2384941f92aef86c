// b2hip.hip - libb2hip.so: host bookkeeping + step orchestration + the C ABI declared in include/b2hip.h.
//
// The world lives in HBM (b2d_world.h). The host keeps (a) the immutable per-body / per-fixture
// parameters it was given, (b) a mirror of the dynamic body state refreshed by the mandatory
// read-back at the end of each step, and (c) the deterministic proxy-id allocator that reproduces
// the ids the reference's dynamic tree would hand out (they define every deterministic ordering,
// b2ContactManager.cpp:64-92). All physics runs in HIP kernels; there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h> // (types only: the library is opened with dlopen by b2hip_shard_connect)
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/b2hip.h"
#include "b2d_kernels_toi_chains.h"
#include "b2d_kernels_toi_domains.h"
// (The three earlier resident large-island solvers of rounds 1 - 2 - grid barrier per colour, polled body rows, pushed
// mailboxes - lived on as a test build until round 3 and were removed in round 4: k_solve_blocks / k_blocks_sweep are
// cross-checked against the launch-per-colour kernels, tests/test_gpu_parity.py. What is left of them compiles out.)
#include "b2d_handover.h"
#define B2HIP_HAVE_VALIDATION_SOLVERS 0
#include "b2d_kernels_solve_blocks.h"
#include "b2d_kernels_sweep_end.h"
#include "b2d_kernels_edit.h"
#include "b2d_kernels_shard.h"
#include "b2d_kernels_spatial.h"
#include "b2d_scan.h"
#include "b2d_shape_geom.h"

static thread_local std::string g_lastError;

static int setError(int code, const std::string& msg)
{
	g_lastError = msg;
	return code;
}

#define HIP_TRY(expr)                                                                                   \
	do                                                                                                  \
	{                                                                                                   \
		hipError_t _e = (expr);                                                                         \
		if (_e != hipSuccess)                                                                           \
		{                                                                                               \
			return setError(B2HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
		}                                                                                               \
	} while (0)

// Device array that keeps its content when it grows.
// (set when librccl is opened, b2hip_shard_connect: releases a world's communicator)
static void (*g_rcclDestroy)(void* comm) = nullptr;

template <typename T>
struct DevArray
{
	T* p = nullptr;
	size_t cap = 0;
	int ensure(size_t n, hipStream_t stream, bool keep = true, bool zeroNew = true)
	{
		if (n <= cap) return 0;
		size_t ncap = cap ? cap : 64;
		while (ncap < n) ncap *= 2;
		T* np = nullptr;
		HIP_TRY(hipMalloc((void**)&np, ncap * sizeof(T)));
		if (zeroNew) HIP_TRY(hipMemsetAsync(np, 0, ncap * sizeof(T), stream));
		if (keep && p && cap) HIP_TRY(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, stream));
		if (p)
		{
			HIP_TRY(hipStreamSynchronize(stream));
			HIP_TRY(hipFree(p));
		}
		p = np;
		cap = ncap;
		return 0;
	}
	void release()
	{
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

struct HostBody
{
	int type;
	uint32_t flags;
	float px, py, qs, qc;   // m_xf
	float cx, cy, a;        // m_sweep.c, a
	float c0x, c0y, a0;
	float lcx, lcy;         // m_sweep.localCenter
	float vx, vy, w;
	float fx, fy, torque;
	float mass, I, invMass, invI;
	float linearDamping, angularDamping, gravityScale;
	float sleepTime;
	int worldIndex;   // slot in b2hip_world::nonStatic (the reference's m_nonStaticBodies), -1 for static bodies
	int dead;         // destroyed (b2World::DestroyBody): the id stays, the body takes no part in anything any more
	int resetSweep;   // SetTransform: the sweep origin (c0, a0) is rewritten from the host mirror at the next upload
	std::vector<int> fixtures; // creation order (the reference's list is newest first)
	bool dirty;
	uint32_t pullEpoch;  // == b2hip_world::mirrorEpoch: this row has been refreshed from (or is newer than) h_state
	uint32_t forceEpoch; // == b2hip_world::stepEpoch: fx, fy, torque were applied since the last step (auto-clear worlds)
};

struct HostFixture
{
	int body;
	int shape;
	float density, friction, restitution;
	uint16_t categoryBits, maskBits;
	int16_t groupIndex;
	bool isSensor, thick;
	int proxyKey;
	float fat[4];
	bool dead;        // destroyed (b2Body::DestroyFixture / DestroyBody): the id stays, the proxy is gone
	bool noProxy;     // the body is inactive (b2Body::SetActive(false)): the fixture lives on without a broad-phase proxy
};

struct GraphSeg
{
	hipGraph_t graph = nullptr;
	hipGraphExec_t exec = nullptr;
	uint64_t sig = 0;
};

struct FreeUnit
{
	int leaf;
};

struct b2hip_world
{
	b2hip_world_def def;
	int device;
	hipStream_t stream;
	bool debugSync;

	std::vector<HostBody> bodies;
	std::vector<HostFixture> fixtures;
	std::vector<ShapeRec> shapes;
	std::map<std::string, int> shapeIndex;
	std::vector<RevoluteJoint> joints;

	// proxy id allocator (b2DynamicTree::AllocateNode / FreeNode, b2DynamicTree.cpp:53-99)
	int nextNode;
	int leafCount;
	std::vector<FreeUnit> freeUnits;

	// what has been uploaded so far
	size_t upBodies, upFixtures, upShapes, upJoints;
	std::vector<int> pendingMoves;
	std::vector<int> dirtyList;   // bodies whose host mirror is newer than the device rows
	std::mutex dirtyMutex;        // the per-body setters may run on several user threads, one body each (ManyBodies.h:39-64)
	size_t stateCount;            // bodies covered by the last read-back in h_state
	uint32_t mirrorEpoch;         // bumped by every read-back into h_state (HostBody::pullEpoch)
	uint32_t stepEpoch;           // bumped by every step (HostBody::forceEpoch)
	bool newFixture;
	float inv_dt0;
	bool stepActive;
	bool callbackWindow;          // inside the step, while the PreSolve callbacks run: mutators are accepted (and applied right after)
	bool failed = false;          // a step failed half-way: the device state is inconsistent, every later call says so
	std::string failedWhy;
	StepParams sp;

	// device
	DW dw;
	DevArray<DState> d_state;
	DevArray<float4> b_pos, b_pos0, b_vel, b_xf, b_mass, b_damp, b_force;
	DevArray<uint32_t> b_flags;
	DevArray<int> b_wake;
	DevArray<float4> p_fat;
	DevArray<int> p_body, p_shape, p_key, p_filter1;
	DevArray<uint32_t> p_filter0;
	DevArray<float2> p_mat;
	DevArray<int> b_proxyHead, p_next, toiList, toiPos2c, toiDestroyList, toiNewList, b_toiGroup, toiGroups, toiGroupCount, toiGroupList, toiMoved, toiNew, toiParent, toiDomOf, toiDomRoot, toiDomCount, toiDomBase, toiDomFill, toiDomList, toiDomFailed, toiDomEvents;
	DevArray<float4> toiHull;
	DevArray<float4> snapBody, snapFat;
	DevArray<ShapeRec> d_shapes;
	DevArray<int4> c_ids[2];
	DevArray<uint64_t> c_key[2];
	DevArray<uint32_t> c_flags[2];
	DevArray<float4> c_mat[2], c_man0[2], c_man1[2], c_imp[2];
	DevArray<int4> c_man3[2];
	DevArray<int> c_color[2], c_mgr[2];
	DevArray<int4> li_ref;
	DevArray<uint64_t> ht_keys;
	DevArray<RevoluteJoint> d_joints;
	DevArray<GearRec> d_gears;
	DevArray<int> jadjStart, jadj, rootJointStart, rootJointCursor, lj_list, rootJointOkay;
	std::vector<std::pair<int, int> > pendingFilter; // body pairs whose contacts must be re-filtered (new joint)
	int nMouseJoints = 0;
	size_t jadjBodies = (size_t)-1, jadjJoints = (size_t)-1; // what the device's per-body joint lists were last built for
	std::vector<GearRec> gears;   // gear joints' own records, appended like joints (the device copy keeps the impulses)
	size_t upGears = 0;
	std::vector<std::pair<int, int> > jointEdits;    // (joint, 1 = also clear the limit impulse, 2 = also the anchors / offsets): members changed by a setter
	DevArray<int> parent, rootSeed, rootBodies, rootContacts, rootJoints, rootIsland, deg, adjStart, adjCursor, adj;
	DevArray<int2> adjSlot;
	DevArray<int4> rootScanIn, rootScanOut;
	DevArray<int> si_root, si_bodyStart, si_contactStart, si_wStart, si_maxLevel, si_bodies, si_contacts, si_level,
		si_stack, si_lastLevel, b_slot, b_island, chunkFirst;
	DevArray<int> li_bodies, li_contacts, li_roots, li_color, colorCount, colorStart, colorCursor, li_sorted;
	DevArray<uint32_t> bodyClaim, rootPen, rootSleepMin;
	DevArray<uint64_t> bodyColorMask, bodyActive, bodyRest;
	DevArray<float4> b_posv, dfInbox;
	DevArray<int> dfRank;
	DevArray<unsigned long long> evKey;
	DevArray<int4> evInfo;
	bool eventsOn = false;
	std::vector<b2hip_contact_event> events; // of the last step, in delivery order
	std::vector<b2hip_toi_callback> toiCallbacks; // listener calls of the last step's TOI sub-steps, in call order
	DevArray<ToiLogRec> toiLog;
	DevArray<int4> toiVerdict; // PreSolve answers for the TOI phase's log slots (DW::toiVerdict)
	DevArray<int> uncolList, compactList, hubRowOf, hubList;
	DevArray<float4> hubDelta;
	DevArray<unsigned long long> hubMeta, hubFirst;
	DevArray<int> rootDone;
	DevArray<float> lc;
	DevArray<int> moveBuf, gridCount, gridStart, gridCursor, gridItems, largeProxies, largeMoves;
	DevArray<float4> gridFat;
	DevArray<unsigned long long> arriveTree;
	DevArray<uint64_t> pairKey, pairKey2;
	DevArray<int2> pairProxy, pairProxy2;
	DevArray<int> pairFirst, pairRank;
	DevArray<int> scanTmp, radixHist, radixHistScan, keepFlag, keepScan;
	DevArray<int4> scanTmp4;
	void* shardComm = nullptr;   // ncclComm_t of a connected sharded world (b2hip_shard_connect)
	DevArray<int> shardSend, shardRecv; // this rank's slab / all ranks' slabs
	size_t shardExchangeBytes = 0;
	bool shardLoopback = false;  // B2HIP_SHARD_LOOPBACK=1: a communicator of ONE rank still runs export -> ncclAllGather -> import (self-test on a one-GPU box)
	// spatial ownership (b2d_kernels_spatial.h; b2hip_shard_spatial)
	bool spatial = false;
	DevArray<uint8_t> b_owner, spNewOwner, spAwake;
	bool spFullRows = false;       // B2HIP_SHARD_FULL_ROWS=1 / b2hip_shard_full_rows: every rank holds every body's current row
	int spRowCap = 1024, spProxyCap = 4096;
	int spIdle[6] = {0, 0, 0, 0, 0, 0}; // exchanges in a row in which a capacity was four times what any rank needed (spCapDecay) // lean E1: records per rank (grown alike on every rank when a header says so)
	DevArray<int> spStraddle, spCount, spTarget, spSend, spRecv;
	std::vector<uint8_t> spOwners; // the owner table as the host last knew it (assignment; refreshed after every resolution)
	bool spOwnersDirty = false;    // owners assigned / bodies created since the table was uploaded
	float spBounds[SHARD_MAX_RANKS + 1] = {0}; // strips along x the owners were dealt by (bodies created later fall into them)
	b2hip_all_gather_fn gatherFn = nullptr; // the caller's all-gather (gloo, tests); null with a connected RCCL communicator
	void* gatherUser = nullptr;
	int* spHost = nullptr;          // pinned staging of the caller's all-gather
	const int* spSendWiped = nullptr; // == spSend.p: its header has been wiped by the last import launch (spPrepareSend)
	int* spHdrHost = nullptr;       // pinned: [0] sequence number, [1] extra word, [2..] the headers of all ranks' slabs (spReadHeaders)
	int* spHdrDev = nullptr;
	int spHdrSeq = 0;
	int* spOwnHost = nullptr;       // pinned: the packed rows of this rank's bodies (id + b2hip_body_state), written by k_end_step
	int* spOwnDev = nullptr;        // ... its device address
	size_t spOwnCapRows = 0;
	size_t spHostWords = 0;
	int spPairCap = 2048, spToiBodyCap = 256, spToiProxyCap = 512; // (records per rank; grown alike on every rank when a header says so)
	int spOwned[SHARD_MAX_RANKS] = {0}, spOwnedProxies[SHARD_MAX_RANKS] = {0};
	long long spMigratedTotal = 0, spResolves = 0, spPairsSent = 0;
	size_t spBytesStep = 0;         // bytes this rank received in the exchanges of the last step
	int spContactsBeforeToi = 0, spToiOrderBefore = 0, spTailCap = 64;
	int spToiUnsafe = 0;            // this rank's Counters::toiUnsafe as its E4 header showed it
	bool spToiSettled = false;      // this step's phase has been through its fallback already
	DevArray<int4> spTailKey;
	DevArray<int2> spVirt;          // body pairs a TOI event would have joined over an ownership boundary (k_sp_tail_pairs)
	// measurement hook (b2hip_shard_tape): the results of this rank's collectives kept in device memory / taken from another
	// world's tape instead of a collective - one rank of a sharded world stepped alone on one GPU (tools/gpu_spatial_share.py)
	bool spTapeRecord = false;
	std::vector<std::pair<int*, size_t> > spTape;
	b2hip_world* spTapeFrom = nullptr;
	size_t spTapeCursor = 0;
	long long spToiRedos = 0;
	size_t spUp = 0;                // bodies the device's owner table covers
	DevArray<int> scanFlags;     // status words of the single-pass scans (b2d_scan.h)
	ScanFlags scanCtx;           // ... with their epoch and the abort word (refreshed by ensureCapacity)
	DevArray<float> stateOut;
	DevArray<int> gridBar;       // grid barrier state of the persistent solver
	int dfEpoch;
	bool solverRows, solverLocal, solverMailbox, noSideStream, profileDetail;
	int collideStage = -1;       // B2HIP_COLLIDE_STAGE=0 / 1: never / always stage the shape records through LDS (default: by the record count)
	int collideSortEnv = -1;     // B2HIP_COLLIDE_SORT=0 / 1: k_collide never / always sorts the contacts of a tile by shape-pair class in LDS
	hipStream_t stream2 = nullptr; // small-island solver beside the large-island one
	hipEvent_t evFork = nullptr, evJoin = nullptr;
	int dfLanesForced, dfSleep, nCU; // k_solve_dataflow: workgroup size, poll back-off, co-resident workgroups
	int persistMaxWG;            // co-resident workgroups of k_solve_persistent on this device (0 = do not use it)
	int persistSteps;            // steps solved by the persistent kernel (diagnostics)
	DevArray<int> consts; // [0] nBodies, [1] gridSize, [2] radix hist count, [3] sorted-pair count
	DevArray<unsigned long long> filterPairs; // sorted body-pair keys of the joints created / destroyed since the last step
	std::vector<int> nonStatic;       // the reference's m_nonStaticBodies: body ids in its order (island seed order)
	bool orderDirty = false;
	DevArray<int> b_order, orderBody;
	DevArray<int> bigRoots;           // sharded worlds: roots of this step's big islands
	// edits of existing fixtures / contacts between steps (b2d_kernels_edit.h)
	std::vector<int2> editOps;        // queued contact-array ops, in call order
	std::vector<int> proxyEdits;      // fixtures whose device proxy row (filter words, body) must be rewritten
	std::vector<int> fatEdits;        // ... and the ones among them whose fat AABB the host has moved (SetTransform)
	bool proxyListsStale = false;     // a fixture was destroyed: b_proxyHead / p_next need a rebuild
	DevArray<int2> d_editOps;
	// listener / filter bridge: user callbacks in the middle of a step (include/b2hip.h)
	b2hip_should_collide_fn filterFn = nullptr;
	void* filterUser = nullptr;
	bool refilterPending = false;   // some contact may carry CF_FILTER (joint created / destroyed, fixture re-filtered)
	b2hip_pre_solve_fn preSolveFn = nullptr;
	b2hip_pre_solve_batch_fn preSolveBatchFn = nullptr;
	b2hip_should_collide_batch_fn filterBatchFn = nullptr;
	void* preSolveUser = nullptr;
	bool postSolveOn = false;
	std::vector<b2hip_contact_impulse> postSolve; // of the last step, in delivery order
	DevArray<float4> pre_o0, pre_o1, pre_oimp;
	DevArray<int4> pre_o3;
	DevArray<PreSolveRec> preRecs;
	DevArray<PostSolveRec> postRecs;
	DevArray<int> filterList, hostList; // hostList: indices uploaded by the host (contacts to disable / reject, pairs to drop)
	// block partition of the large islands (b2d_kernels_solve_blocks.h)
	DevArray<int> b_adoptStage;
	DevArray<int> b_blk1, b_adopt, blkRows, blkRowStart, blkCursor, blkBodyStart, blkBodies, rowColor, blkBodyCount, blkBodyCursor;
	DevArray<float4> b_cutv;
	int blocksMaxWG = 0;         // co-resident workgroups of k_solve_blocks on this device (0 = do not use it)
	int sweepMaxWG[3] = { 0, 0, 0 }; // ... of k_blocks_sweep<256 / 512 / 1024>
	int hubWaves = 8;            // waves of k_large_hub (B2HIP_HUB_WAVES=1: one)
	bool sweepEnd = true;        // k_sweep_end closes every sweep of the launch-per-colour solver (B2HIP_NO_SWEEP_END=1: round 4's launches)
	bool sweepTail = true;       // ... and takes the small colours (B2HIP_NO_TAIL=1: a launch per colour)
	int tailRowsMax = 1024;      // a colour with at most this many rows in the step's census is a tail colour (B2HIP_TAIL_ROWS): one
	                             // round of the workgroup. Measured on the settled Tumbler (profiles/r05_b): a round costs the
	                             // workgroup ~4.5 us - the same chain of dependent loads a launch pays - so a colour of 8 000
	                             // rows is 9 rounds = 40 us against 5.8 us as a launch of its own (tail colours up to 8 192 rows:
	                             // 4.94 ms per step; none: 3.87; round 4's launches: 4.57)
	int recolorSlack = 2;        // colour afresh when the colours in use exceed the last fresh colouring's by more than this (B2HIP_RECOLOR_SLACK; -1: every 64th step as in round 4)
	int freshColors = 0;         // colours the last colouring from scratch of a partition-less world needed (0: none yet); in the snapshot's hints
	bool freshColorsPending = false;
	bool restFlow = true;        // the small colours of a sweep as data flow per body in one launch (k_large_rest; B2HIP_NO_REST=1: launches / tail)
	int restRowsMax = 65536;     // ... the highest colours that hold at most this many rows together (B2HIP_REST_ROWS). Measured on the
	                             // settled Tumbler (profiles/r05_i_rest_rows_sweep.txt: the solver family per step, 21 colours): none 2.06 ms /
	                             // 260 launches per step; 16 384 rows 2.01 / 236; 50 000 1.89 / 188; 80 000 1.86 / 164; 180 000 1.88 / 116 -
	                             // a hop costs more the more lanes poll
	int lastTailFirst = 0, lastRestFirst = 0, lastSweepLaunches = 0; // diagnostics of the last step
	long long launchCount = 0;   // kernels launched on the main stream so far (LAUNCH)
	long long familyLaunchesAtStart = 0; int familyLaunches = 0; // ... by the large-island solver family in the last step (timing mode 5)
	int largeHintSteps = 120;    // > 0: the world has had large islands lately (k_color_check / k_block_census run with the island build)
	int serialOrphansNext = 0;   // DW::serialOrphans of the next step
	int adoptSticky = 0;
	bool adoptPasses = false;    // the last step had orphan constraints (or made a partition): run k_block_adopt this step
	bool traceLaunches = false;  // B2HIP_TRACE_LAUNCHES=1 (with B2HIP_DEBUG): every kernel's name before the stream is drained behind it
	bool tracePartition = false; // B2HIP_TRACE_PARTITION=1: why a partition was made, on stderr
	bool gridHalf = false;       // the hash grid's cell is half the limit: chosen from the candidates per moved proxy of the last pair update
	bool gridForced = false;     // B2HIP_GRID_HALF=0 / 1 fixes it
	int pairsLargeSticky = 0;    // steps for which the pair update still reads its pair count back before it sorts
	int toiPreSolveReruns = 0; // runs of the TOI phase repeated because a PreSolve changed its contact inside a sub-step
	int toiGridRetries = 0;      // steps whose chains were run again with the hash grid instead of serially
	int toiChainContacts = 0;    // contacts created by the close-out of the parallel TOI chains since the world was made
	int recolorCountdown = 0;    // ... and steps until such islands are coloured afresh (phaseSolve)
	bool blocksTooBig = false;   // the large islands hold more constraints than any block solver takes: no partition (phaseSolve)
	bool noSweepBlocks = false;  // B2HIP_NO_SWEEP_BLOCKS=1: jointed / hub islands stay on the launch-per-colour kernels
	int dfWipedAt = 0;           // dfEpoch >> 14 at the last wipe of the hand-over rows
	int sweepSteps = 0;          // steps whose large islands went through k_blocks_sweep
	int blockLanes = 0;          // forced workgroup size of k_solve_blocks (B2HIP_BLOCK_LANES), 0 = chosen per partition
	bool noBlocks = false;       // B2HIP_NO_BLOCKS=1: no block partition (large islands through the launch-per-colour kernels)
	int blockSteps = 0;          // steps solved by k_solve_blocks (diagnostics)

	// pinned host buffers
	float* h_state;             // pinned, coherent: k_end_step writes the read-back into it (d_hstate = its device address)
	float* d_hstate = nullptr;
	int stateSeq = 0;            // sequence number of the last read-back asked for (awaitState)
	size_t h_stateCap;
	DState* h_dstate;
	DState* h_pub = nullptr;     // where k_block_census publishes the island census (pinned, coherent); polled by awaitCensus
	DState* d_pub = nullptr;     // ... its device address
	int pubSeq = 0;
	bool noCensusPoll = false;   // B2HIP_NO_CENSUS_POLL=1: copy + stream synchronisation instead (for comparison)
	bool noStatePoll = false;    // B2HIP_NO_STATE_POLL=1: the same for the read-back at the end of the step
	// b2hip_set_lazy_readback: a step ends with the counters only; the 40 B per body stay on the device until a body's state
	// is asked for (rowsPending: h_state's rows are older than the device's; fetched once, by whoever asks first)
	bool lazyReadback = false;
	std::atomic<bool> rowsPending{false};
	// The rows travel only where they differ from what the host's buffer holds: stateOut is the device's copy of h_state's rows
	// (k_end_step, rowMode), valid while these three are what they were when it was last written in full.
	const float* shadowDev = nullptr;
	const float* shadowHost = nullptr;
	size_t shadowRows = 0;
	// ... which lets most of a large world's rows leave early, behind SynchronizeFixtures, on a stream of their own while the
	// pair update and the TOI phase run (startEarlyRows); the launch at the end of the step sends what changed since.
	hipStream_t rowStream = nullptr;
	hipEvent_t rowFork = nullptr, rowJoin = nullptr;
	bool rowsForked = false, rowsEarlyPending = false;
	int earlyRowsMin = 65536;       // bodies from which the early launch pays (B2HIP_EARLY_ROWS_MIN; 0 = never)
	std::mutex rowsMutex;
	bool blocksThisStep = false; // the large islands of this step went through k_solve_blocks

	Counters last;        // counters of the last completed step
	int lastContacts;
	float profile[13];
	hipEvent_t ev[13];
	float solverMs;
	double solverBytes;
	int solverConstraints, solverBodies;
	int forceLarge;
	// optional per-launch timing of the dominant solver kernel
	size_t pairCapHint = 0; // pair-buffer size asked for after an overflow (growPairBuffers)
	int constsUploaded[2] = { -1, -1 };
	int* constsUploadedAt = nullptr;
	int toiSyncSticky = 0; // steps for which the TOI phase decides from a read-back again (see phaseToi)
	int toiGridSticky = 0; // steps for which the TOI chains still get a rebuilt hash grid
	bool toiChainsHadGrid = false; // the chains of this step ran with the grid (else a moved proxy is all "unsafe" means)
	bool toiSnapshotTaken = false; // this step's TOI phase saved the state it started from (k_toi_snapshot)
	std::vector<int4> toiVerdicts; // this step's PreSolve answers per TOI log slot (the device's copy: DW::toiVerdict)
	// b2World::SetSubStepping (b2World.h:183; b2World.cpp:1082-1086, 1668): with the flag on a step call solves one TOI event
	// and leaves the step open; the calls that follow run Collide and the next event but no island solve, until no event is left
	bool stepComplete = true;  // b2World::m_stepComplete
	bool stepSolves = true;    // this call runs Solve (it started from a complete step)
	bool toiCountersFresh = false, toiSpeculative = false, toiSyncOnly = false, toiNoDomains = false;
	bool toiRan, toiEventValid, toiChains, toiSerialOnly, kernelTimingLaunches, solverBarriers, colorSmallPending;
	bool useGraphs;              // replay the host-decision-free launch sequences as hipGraphs (B2HIP_GRAPHS=1)
	int graphCaptures;
	GraphSeg segCollide, segIslands, segPairs;
	int hubSteps;
	int toiFallbacks;                                  // steps whose TOI chains had to be redone serially
	bool debugTrace;                                   // B2HIP_TRACE=1: hash the body state after every solver stage
	std::vector<std::pair<std::string, uint64_t> > trace;
	DevArray<float4> dbgPreVel, dbgVel;
	DevArray<int> dbgLi;
	int kernelTiming;
	long long ktUnitsA, ktUnitsB; // units behind the bandwidth kernels' byte counts (set by b2hip_set_kernel_timing_units)
	std::vector<hipEvent_t> ktEvents;
	int ktUsed;          // events recorded this step (pairs)
	int ktKind;          // 0 none, 1 k_large_velocity, 2 k_solve_small
	float ktMs;
	int ktLaunches;
	double ktBytes;
};

// Every entry point that touches the device runs with the world's device current and puts the caller's device back
// afterwards: two worlds on different GPUs in one process, or a step from another thread, stay on their own device.
struct DeviceGuard
{
	int prev = -1;
	bool switched = false;
	explicit DeviceGuard(int device)
	{
		if (device < 0 || hipGetDevice(&prev) != hipSuccess || prev == device) return;
		switched = hipSetDevice(device) == hipSuccess;
	}
	~DeviceGuard()
	{
		if (switched) (void)hipSetDevice(prev);
	}
	DeviceGuard(const DeviceGuard&) = delete;
	DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DEVICE_GUARD(w) DeviceGuard _deviceGuard((w)->device)

// ------------------------------------------------------------------------------------------------
static int nextPow2(size_t n)
{
	size_t p = 64;
	while (p < n) p <<= 1;
	return (int)p;
}

// The read-back buffer h_state IS the host mirror of the dynamic state; a HostBody is refreshed from it only
// when the host is about to edit that body (no O(bodies) host loop per step).
static void ensureRows(b2hip_world* w);
static void pullBody(b2hip_world* w, int i)
{
	if ((size_t)i >= w->stateCount || w->h_state == nullptr) return;
	HostBody& b = w->bodies[i];
	if (b.pullEpoch != w->mirrorEpoch) ensureRows(w);
	// once pulled, the host row is the newer one until the next read-back (an upload in between - a contact read flushes
	// the edits made so far - does not make h_state any fresher)
	if (b.pullEpoch == w->mirrorEpoch) return;
	b.pullEpoch = w->mirrorEpoch;
	const float* o = w->h_state + 10 * (size_t)i;
	b.px = o[0]; b.py = o[1]; b.a = o[2];
	b.vx = o[3]; b.vy = o[4]; b.w = o[5];
	b.cx = o[6]; b.cy = o[7];
	uint32_t f;
	memcpy(&f, o + 8, 4);
	b.flags = (b.flags & ~0x7fu) | (f & 0x7fu);
	b.sleepTime = o[9];
	b.c0x = b.cx; b.c0y = b.cy; b.a0 = b.a;
	b.qs = sinf(b.a);
	b.qc = cosf(b.a);
	if (w->def.auto_clear_forces && b.forceEpoch != w->stepEpoch) { b.fx = b.fy = b.torque = 0.0f; }
	b.forceEpoch = w->stepEpoch;
}

static void markDirty(b2hip_world* w, int i)
{
	HostBody& b = w->bodies[i];
	if (b.dirty) return;
	pullBody(w, i);
	b.dirty = true;
	std::lock_guard<std::mutex> lock(w->dirtyMutex);
	w->dirtyList.push_back(i);
}

// b2Body::SetAwake(true) (b2Body.h:690-718): the flag is set and the sleep timer restarts whether the body was asleep or not
// (a slowly dragged mouse joint keeps its body awake this way). A static body's timer is never read: only its flag matters.
static void setAwake(b2hip_world* w, int i)
{
	if (w->bodies[i].type == B2HIP_STATIC_BODY && (w->bodies[i].flags & BF_AWAKE) != 0) return;
	markDirty(w, i);
	HostBody& b = w->bodies[i];
	b.flags |= BF_AWAKE;
	b.sleepTime = 0.0f;
}

// The ids the reference's b2DynamicTree hands out (AllocateNode / FreeNode, b2DynamicTree.cpp:53-99): a LIFO free list of
// node ids in front of a growing pool. CreateProxy takes one node for the leaf and - unless the tree is empty - InsertLeaf
// one more for the new internal parent; DestroyProxy gives back the parent RemoveLeaf drops (unless the leaf was the root)
// and then the leaf, so the next CreateProxy reuses exactly that leaf id. Which id the internal node had is never
// observable (only leaves are proxies): it sits in the list as a marker (-1).
static int allocProxyKey(b2hip_world* w)
{
	int key;
	if (!w->freeUnits.empty())
	{
		key = w->freeUnits.back().leaf;
		if (key < 0) return -1; // an internal node's id would become a leaf id (the tree was emptied and refilled): not modelled
		w->freeUnits.pop_back();
		if (w->leafCount > 0)
		{
			// InsertLeaf's parent node comes off the free list as well, or from the pool
			if (!w->freeUnits.empty()) w->freeUnits.pop_back();
			else w->nextNode++;
		}
	}
	else
	{
		key = w->nextNode++;
		if (w->leafCount > 0) w->nextNode++; // the internal parent node InsertLeaf allocates
	}
	w->leafCount++;
	return key;
}

// b2DynamicTree::DestroyProxy (b2DynamicTree.cpp:121-128): RemoveLeaf frees the parent (if the leaf is not the root), then the leaf
static void freeProxyKey(b2hip_world* w, int key)
{
	FreeUnit u;
	if (w->leafCount > 1)
	{
		u.leaf = -1;
		w->freeUnits.push_back(u);
	}
	u.leaf = key;
	w->freeUnits.push_back(u);
	w->leafCount--;
}

static int internShape(b2hip_world* w, const ShapeRec& s)
{
	std::string bytes((const char*)&s, sizeof(ShapeRec));
	std::map<std::string, int>::iterator it = w->shapeIndex.find(bytes);
	if (it != w->shapeIndex.end()) return it->second;
	int idx = (int)w->shapes.size();
	w->shapes.push_back(s);
	w->shapeIndex[bytes] = idx;
	return idx;
}

// Host evaluation of shape AABB / mass uses the same header the kernels use (b2d_collide.h), built
// for the host by hipcc; host libm sinf/cosf == b2dSin/b2dCos bit for bit (see b2d_math.h).
static Xf hostXf(const HostBody& b)
{
	Xf xf;
	xf.p = v2(b.px, b.py);
	xf.q.s = b.qs;
	xf.q.c = b.qc;
	return xf;
}

// fixture mass: the shared geometry module (b2d_shape_geom.h), the same routine the drop-in host shape classes call
static void shapeMass(const ShapeRec& s, float density, float* massOut, V2* centerOut, float* IOut)
{
	const MassProps mp = b2dShapeMass(&s, density);
	*massOut = mp.mass;
	*centerOut = mp.center;
	*IOut = mp.inertia;
}

// b2Body::ResetMassData (b2Body.cpp:310-385)
static void resetMassData(b2hip_world* w, HostBody& b)
{
	b.mass = 0.0f;
	b.invMass = 0.0f;
	b.I = 0.0f;
	b.invI = 0.0f;
	b.lcx = b.lcy = 0.0f;
	if (b.type == B2HIP_STATIC_BODY || b.type == B2HIP_KINEMATIC_BODY)
	{
		b.c0x = b.cx = b.px;
		b.c0y = b.cy = b.py;
		b.a0 = b.a;
		return;
	}
	V2 localCenter = v2(0.0f, 0.0f);
	// the reference walks its fixture list newest first
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
	{
		const HostFixture& f = w->fixtures[b.fixtures[k]];
		if (f.density == 0.0f) continue;
		float mass, I;
		V2 center;
		shapeMass(w->shapes[f.shape], f.density, &mass, &center, &I);
		b.mass += mass;
		localCenter += mass * center;
		b.I += I;
	}
	if (b.mass > 0.0f)
	{
		b.invMass = 1.0f / b.mass;
		localCenter *= b.invMass;
	}
	else
	{
		b.mass = 1.0f;
		b.invMass = 1.0f;
	}
	if (b.I > 0.0f && (b.flags & BF_FIXEDROT) == 0)
	{
		b.I -= b.mass * b2dDot(localCenter, localCenter);
		b.invI = 1.0f / b.I;
	}
	else
	{
		b.I = 0.0f;
		b.invI = 0.0f;
	}
	V2 oldCenter = v2(b.cx, b.cy);
	b.lcx = localCenter.x;
	b.lcy = localCenter.y;
	V2 c = b2dMulXV(hostXf(b), localCenter);
	b.c0x = b.cx = c.x;
	b.c0y = b.cy = c.y;
	V2 dv = b2dCrossSV(b.w, c - oldCenter);
	b.vx += dv.x;
	b.vy += dv.y;
}

// ------------------------------------------------------------------------------------------------
static int syncCheck(b2hip_world* w, const char* what)
{
	if (!w->debugSync) return 0;
	if (w->traceLaunches) { fprintf(stderr, "[b2hip] %s\n", what); fflush(stderr); } // (B2HIP_TRACE_LAUNCHES=1: which launch hangs?)
	hipError_t e = hipStreamSynchronize(w->stream);
	if (e == hipSuccess) e = hipGetLastError();
	if (e != hipSuccess) return setError(B2HIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
	return 0;
}

// b2Profile without events: stampPhase(w, k) asks the NEXT kernel launched on the main stream to note the device clock in
// DState::phaseClock[k] as it starts (b2dPhaseStamp, first statement of every kernel that takes the DW block).
static inline void stampPhase(b2hip_world* w, int slot)
{
	if (w->profileDetail) w->dw.stampMask |= 1u << slot;
}

template <typename A, typename... R>
static inline void stampsTaken(b2hip_world* w, const A&, const R&...)
{
	if (std::is_same<typename std::decay<A>::type, DW>::value) w->dw.stampMask = 0u;
}

// A launch that the runtime refuses (bad configuration, wrong device current, lost context) is reported at once:
// hipGetLastError needs no synchronisation. With B2HIP_DEBUG the stream is drained after every launch as well.
#define LAUNCH(w, kernel, grid, block, ...)                                                   \
	do                                                                                        \
	{                                                                                         \
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (w)->stream, __VA_ARGS__);     \
		(w)->launchCount += 1;                                                                \
		stampsTaken((w), __VA_ARGS__);                                                        \
		hipError_t _le = hipGetLastError();                                                   \
		if (_le != hipSuccess) return setError(B2HIP_ERR_HIP, std::string(#kernel) + " launch: " + hipGetErrorString(_le)); \
		int _rc = syncCheck((w), #kernel);                                                    \
		if (_rc) return _rc;                                                                  \
	} while (0)

// Same on an explicit stream (the small-island side stream, see phaseSolve).
#define LAUNCH_ON(w, strm, kernel, grid, block, ...)                                          \
	do                                                                                        \
	{                                                                                         \
		const uint32_t _sm = (w)->dw.stampMask;                                               \
		if ((strm) != (w)->stream) (w)->dw.stampMask = 0u; /* phase stamps belong to the main stream */ \
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (strm), __VA_ARGS__);          \
		if ((strm) != (w)->stream) (w)->dw.stampMask = _sm; else stampsTaken((w), __VA_ARGS__); \
		hipError_t _le = hipGetLastError();                                                   \
		if (_le != hipSuccess) return setError(B2HIP_ERR_HIP, std::string(#kernel) + " launch: " + hipGetErrorString(_le)); \
		if ((w)->debugSync)                                                                   \
		{                                                                                     \
			hipError_t _e = hipStreamSynchronize(strm);                                       \
			if (_e == hipSuccess) _e = hipGetLastError();                                     \
			if (_e != hipSuccess) return setError(B2HIP_ERR_HIP, std::string(#kernel) + ": " + hipGetErrorString(_e)); \
		}                                                                                     \
	} while (0)

// ---- hipGraph segments ------------------------------------------------------------------------------
// The step is ~55 kernels of 2-5 us: issued one by one the host (~3.5 us per launch) is the bottleneck between two
// read-backs. The three launch sequences that contain no host decision (collide + compaction, island build up to the
// census read-back, end-of-step pair update) are captured once per world layout and replayed as one graph launch each.
// A segment is re-captured when anything baked into the kernel arguments changes (the DW pointer block, capacities).
static uint64_t segSignature(const b2hip_world* w, uint64_t extra)
{
	uint64_t h = 1469598103934665603ull ^ extra;
	h = (h ^ (uint64_t)(uintptr_t)w->scanTmp4.p) * 1099511628211ull;
	h = (h ^ (uint64_t)(uintptr_t)w->consts.p) * 1099511628211ull;
	h = (h ^ (uint64_t)(uintptr_t)w->stream) * 1099511628211ull;
	const unsigned char* p = (const unsigned char*)&w->dw;
	for (size_t i = 0; i < sizeof(DW); ++i)
	{
		h ^= p[i];
		h *= 1099511628211ull;
	}
	return h;
}

template <typename F>
static int runSegment(b2hip_world* w, GraphSeg& seg, uint64_t extra, F launches)
{
	if (!w->useGraphs || w->debugSync || w->debugTrace) return launches();
	const uint64_t sig = segSignature(w, extra);
	if (!seg.exec || seg.sig != sig)
	{
		if (seg.exec) (void)hipGraphExecDestroy(seg.exec);
		if (seg.graph) (void)hipGraphDestroy(seg.graph);
		seg.exec = nullptr;
		seg.graph = nullptr;
		HIP_TRY(hipStreamBeginCapture(w->stream, hipStreamCaptureModeThreadLocal));
		const int rc = launches();
		hipError_t e = hipStreamEndCapture(w->stream, &seg.graph);
		if (rc) return rc;
		if (e != hipSuccess) return setError(B2HIP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
		HIP_TRY(hipGraphInstantiate(&seg.exec, seg.graph, nullptr, nullptr, 0));
		seg.sig = sig;
		w->graphCaptures += 1;
	}
	HIP_TRY(hipGraphLaunch(seg.exec, w->stream));
	w->dw.stampMask = 0u; // (taken by the first kernel of the segment: the mask is part of the segment's signature)
	return 0;
}

static inline bool hasFilter(const b2hip_world* w) { return w->filterFn != nullptr || w->filterBatchFn != nullptr; }
static inline bool hasPreSolve(const b2hip_world* w) { return w->preSolveFn != nullptr || w->preSolveBatchFn != nullptr; }
// any listener callback switched on: the TOI sub-steps log their calls (b2hip_get_toi_callbacks) and run in serial order
static inline bool listenerOn(const b2hip_world* w) { return w->eventsOn || hasPreSolve(w) || w->postSolveOn; }

static int ktRecord(b2hip_world* w)
{
	if (!w->kernelTiming) return 0;
	if ((size_t)w->ktUsed >= w->ktEvents.size())
	{
		hipEvent_t e;
		HIP_TRY(hipEventCreate(&e));
		w->ktEvents.push_back(e);
	}
	HIP_TRY(hipEventRecord(w->ktEvents[w->ktUsed++], w->stream));
	return 0;
}

// b2hip_set_kernel_timing modes 2 / 3 / 4: an event pair around k_collide / k_sync_fixtures / k_find_pairs_small
static int ktBracket(b2hip_world* w, int mode, int kind)
{
	if (w->kernelTiming != mode) return 0;
	w->ktKind = kind;
	return ktRecord(w);
}

static int gridFor(size_t n, int block = 256, int maxBlocks = 2048)
{
	size_t g = (n + block - 1) / block;
	if (g < 1) g = 1;
	if (g > (size_t)maxBlocks) g = maxBlocks;
	return (int)g;
}

static int readState(b2hip_world* w)
{
	HIP_TRY(hipMemcpyAsync(w->h_dstate, w->d_state.p, sizeof(DState), hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	return 0;
}

// The island census as k_block_census published it under sequence number w->pubSeq (straight into pinned host memory):
// the host polls the number instead of queueing a copy and synchronising the stream - which also lets the stream run on
// (k_color_small, queued behind the census) while the host sizes the solver launches.
// The polling loop of awaitCensus / awaitState: until *seq == want. Like hipStreamSynchronize it waits as long as the stream
// is busy (a step of a pathological world can take a minute); it gives up only if the stream reports an error, or has
// drained and the number still is not there two seconds later (the publishing kernel did not run: a bug, not a wait).
static int pollPublished(b2hip_world* w, volatile const int* seq, int want, const char* what)
{
	bool drained = false;
	std::chrono::steady_clock::time_point drainedAt;
	// ... and, as a backstop, after a generous wall-clock deadline (B2HIP_STEP_DEADLINE_S, default 300 s): every device-side
	// wait is bounded (PERSIST_SPIN_MAX, SCAN_SPIN_MAX), so a stream that stays busy that long is lost, and the caller gets an
	// error and a failed world instead of a Step() that never returns.
	static const double deadlineS = getenv("B2HIP_STEP_DEADLINE_S") ? atof(getenv("B2HIP_STEP_DEADLINE_S")) : 300.0;
	const auto startedAt = std::chrono::steady_clock::now();
	for (unsigned spins = 1; *seq != want; ++spins)
	{
		if ((spins & 0x3fff) == 0)
		{
			const hipError_t q = hipStreamQuery(w->stream);
			if (q != hipSuccess && q != hipErrorNotReady) return setError(B2HIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
			if (q == hipSuccess)
			{
				const auto now = std::chrono::steady_clock::now();
				if (!drained) { drained = true; drainedAt = now; }
				else if (now - drainedAt > std::chrono::seconds(2)) return setError(B2HIP_ERR_HIP, std::string(what) + " was not published (the stream has drained)");
			}
			else drained = false;
			if (std::chrono::duration<double>(std::chrono::steady_clock::now() - startedAt).count() > deadlineS)
				return setError(B2HIP_ERR_HIP, std::string(what) + ": the device did not finish the step within the deadline (B2HIP_STEP_DEADLINE_S)");
		}
#if defined(__x86_64__)
		__builtin_ia32_pause();
#endif
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	return 0;
}

static int awaitCensus(b2hip_world* w)
{
	if (int rc = pollPublished(w, (volatile const int*)&w->h_pub->pubSeq, w->pubSeq, "island census")) return rc;
	memcpy(w->h_dstate, w->h_pub, offsetof(DState, pubSeq));
	return 0;
}

// Size every buffer for the current topology and a contact / pair budget; refresh the kernarg block.
static int ensureCapacity(b2hip_world* w, size_t needContacts)
{
	hipStream_t s = w->stream;
	const size_t nb = std::max<size_t>(w->bodies.size(), 1);
	const size_t np = std::max<size_t>(w->fixtures.size(), 1);
	int rc = 0;
#define ENS(arr, n) do { rc = w->arr.ensure((n), s); if (rc) return rc; } while (0)
	ENS(d_state, 1);
	ENS(b_pos, nb); ENS(b_pos0, nb); ENS(b_vel, nb); ENS(b_xf, nb); ENS(b_mass, nb); ENS(b_damp, nb); ENS(b_force, nb);
	ENS(b_flags, nb); ENS(b_wake, nb); ENS(b_order, nb); ENS(orderBody, nb); ENS(bigRoots, SHARD_BIG_MAX);
	ENS(p_fat, np); ENS(p_body, np); ENS(p_shape, np); ENS(p_key, np); ENS(p_filter0, np); ENS(p_filter1, np); ENS(p_mat, np);
	ENS(b_proxyHead, nb); ENS(p_next, np);
	ENS(d_shapes, std::max<size_t>(w->shapes.size(), 1));
	ENS(d_joints, std::max<size_t>(w->joints.size(), 1));
	ENS(d_gears, std::max<size_t>(w->gears.size(), 1));
	ENS(jadjStart, nb + 2); ENS(jadj, 2 * w->joints.size() + 2); ENS(rootJointStart, nb + 2); ENS(rootJointCursor, nb);
	ENS(lj_list, w->joints.size() + 2); ENS(rootJointOkay, nb);
	const size_t capPairs = std::max<size_t>(std::max<size_t>(8 * np + 4096, w->pairKey.cap), w->pairCapHint);
	const size_t capContacts = std::max<size_t>(needContacts + capPairs, 1024);
	for (int k = 0; k < 2; ++k)
	{
		ENS(c_ids[k], capContacts); ENS(c_key[k], capContacts); ENS(c_flags[k], capContacts); ENS(c_mat[k], capContacts);
		ENS(c_man0[k], capContacts); ENS(c_man1[k], capContacts); ENS(c_imp[k], capContacts); ENS(c_man3[k], capContacts);
		ENS(c_color[k], capContacts); ENS(c_mgr[k], capContacts);
	}
	const size_t cc = w->c_ids[0].cap; // actual (power of two) capacity
	// hash set: at most 50 % load
	{
		size_t want = (size_t)nextPow2(2 * cc);
		if (w->ht_keys.cap < want)
		{
			rc = w->ht_keys.ensure(want, s, false);
			if (rc) return rc;
		}
	}
	ENS(parent, nb); ENS(rootSeed, nb); ENS(rootBodies, nb); ENS(rootContacts, nb); ENS(rootJoints, nb); ENS(rootIsland, nb);
	ENS(deg, nb + 1); ENS(adjStart, nb + 2); ENS(adjCursor, nb); ENS(adj, 2 * cc); ENS(adjSlot, cc);
	ENS(rootScanIn, nb + 1); ENS(rootScanOut, nb + 2);
	ENS(si_root, nb + 1); ENS(si_bodyStart, nb + 2); ENS(si_contactStart, nb + 2); ENS(si_wStart, nb + 2); ENS(si_maxLevel, nb + 1);
	ENS(si_bodies, nb); ENS(si_contacts, cc); ENS(si_level, cc); ENS(si_stack, nb); ENS(si_lastLevel, nb);
	ENS(b_slot, nb); ENS(b_island, nb); ENS(chunkFirst, (nb + cc) / (TINY_CHUNK_LANES / 2) + 4);
	ENS(li_bodies, nb); ENS(li_contacts, cc); ENS(li_roots, nb); ENS(li_color, cc);
	ENS(colorCount, cc + 2 + COLOR_SLOT_PADDED * COLOR_SLOT_STRIDE); ENS(colorStart, cc + 2); ENS(colorCursor, cc + 2 + COLOR_SLOT_PADDED * COLOR_SLOT_STRIDE); ENS(li_sorted, cc); ENS(li_ref, cc); // (colorSlot: the first 65 colour counters on a line each)
	ENS(bodyClaim, nb); ENS(bodyColorMask, nb); ENS(bodyActive, nb); ENS(bodyRest, nb); ENS(b_posv, nb); ENS(dfRank, B2HIP_HAVE_VALIDATION_SOLVERS ? nb * 32 : 1); ENS(dfInbox, B2HIP_HAVE_VALIDATION_SOLVERS ? 2 * cc : 1); /* (mailbox tables of the test build's k_solve_mailbox: DF_RANKS = 32 slots per body) */ ENS(evKey, cc); ENS(evInfo, cc); ENS(uncolList, COLOR_SMALL_MAX); ENS(compactList, COLOR_SMALL_MAX); ENS(hubRowOf, cc); ENS(hubList, cc); ENS(hubDelta, cc); ENS(hubMeta, 8); ENS(hubFirst, nb); ENS(rootPen, ROOT_PEN_SLOTS * nb); ENS(rootDone, nb); ENS(rootSleepMin, nb);
	if (w->lc.cap < (size_t)LC_WORDS * cc)
	{
		rc = w->lc.ensure((size_t)LC_WORDS * cc, s, false, false);
		if (rc) return rc;
	}
	ENS(moveBuf, 2 * np + 64);
	const size_t gridSize = (size_t)nextPow2(2 * np);
	ENS(gridCount, gridSize); ENS(gridStart, gridSize + 2); ENS(gridCursor, gridSize); ENS(gridItems, np); ENS(gridFat, np); ENS(arriveTree, (size_t)ARRIVE_SITES * TREE_WORDS); ENS(largeProxies, np); ENS(largeMoves, 2 * np + 64);
	ENS(pairKey, capPairs); ENS(pairKey2, capPairs); ENS(pairProxy, capPairs); ENS(pairProxy2, capPairs);
	ENS(pairFirst, capPairs + 1); ENS(pairRank, capPairs + 2);
	const size_t maxScanN = std::max(std::max(nb + 2, gridSize + 2), std::max(cc + 2, capPairs + 2));
	const size_t radixTiles = capPairs / RADIX_TILE + 2;
	ENS(radixHist, 256 * radixTiles + 2); ENS(radixHistScan, 256 * radixTiles + 4);
	ENS(scanTmp, 3 * (std::max(maxScanN, 256 * radixTiles) / SCAN_TILE + 8));
	ENS(scanTmp4, 3 * (maxScanN / SCAN_TILE + 8));
	ENS(scanFlags, std::max(maxScanN, 256 * radixTiles) / SCAN_TILE + 8);
	ENS(keepFlag, cc + 1); ENS(keepScan, cc + 2);
	ENS(toiList, cc); ENS(toiPos2c, cc); ENS(toiDestroyList, cc); ENS(toiNewList, TOI_NEW_LIST_MAX);
	ENS(b_toiGroup, nb); ENS(toiGroups, nb); ENS(toiGroupCount, std::min<size_t>(nb, TOI_GROUPS_MAX)); ENS(toiGroupList, std::min<size_t>(nb, TOI_GROUPS_MAX) * CHAIN_ADJ_MAX); ENS(toiMoved, TOI_MOVED_ALL_MAX); ENS(toiNew, 8 * TOI_NEWPAIR_MAX); ENS(toiParent, nb); ENS(toiDomOf, nb); ENS(toiDomRoot, TOI_DOMAINS_MAX); ENS(toiDomCount, TOI_DOMAINS_MAX); ENS(toiDomBase, TOI_DOMAINS_MAX); ENS(toiDomFill, TOI_DOMAINS_MAX); ENS(toiDomFailed, TOI_DOMAINS_MAX); ENS(toiDomEvents, TOI_DOMAINS_MAX); ENS(toiDomList, cc); ENS(toiHull, np); ENS(snapBody, 5 * nb); ENS(snapFat, np);
	{
		// listener bridge buffers: contact-sized only while the callback that needs them is installed
		const size_t nPre = hasPreSolve(w) ? cc : 1, nPost = w->postSolveOn ? cc : 1, nFil = hasFilter(w) ? cc : 1;
		ENS(pre_o0, nPre); ENS(pre_o1, nPre); ENS(pre_oimp, nPre); ENS(pre_o3, nPre); ENS(preRecs, nPre);
		ENS(postRecs, nPost); ENS(filterList, nFil);
		ENS(toiLog, listenerOn(w) && w->def.continuous ? cc : 1);
		ENS(toiVerdict, hasPreSolve(w) && w->def.continuous ? cc : 1);
		ENS(hostList, std::max<size_t>(std::max(4 * nPre, nFil), hasFilter(w) ? capPairs : 1)); // (PreSolve material edits: 4 words each)
	}
	ENS(b_blk1, nb); ENS(b_adopt, nb); ENS(b_adoptStage, 3 * nb); ENS(blkRows, (size_t)(MAX_BLOCKS + 2) * BLK_SLOT); ENS(blkRowStart, MAX_BLOCKS + 2); ENS(blkCursor, (size_t)(MAX_BLOCKS + 2) * BLK_SLOT); ENS(blkBodyCount, MAX_BLOCKS + 2); ENS(blkBodyCursor, MAX_BLOCKS + 2);
	ENS(blkBodyStart, MAX_BLOCKS + 2); ENS(blkBodies, nb); ENS(rowColor, cc); ENS(b_cutv, nb);
	ENS(b_owner, w->spatial ? nb : 1); ENS(spNewOwner, w->spatial ? nb : 1); ENS(spAwake, w->spatial ? nb : 1); ENS(spStraddle, w->spatial ? std::max<size_t>(w->spStraddle.cap, 4096) : 1);
	ENS(spCount, w->spatial ? (size_t)SP_RESOLVE_MAX * SHARD_MAX_RANKS : 1); ENS(spTarget, w->spatial ? SP_RESOLVE_MAX : 1);
	ENS(spTailKey, w->spatial ? capContacts : 1); ENS(spVirt, w->spatial ? SP_TAIL_MAX + 1 : 1);
	ENS(stateOut, 12 * nb + sizeof(DState) / sizeof(float) + 4); // (+ the counters, behind the rows: one copy to the host per step)
	ENS(consts, 16);
	ENS(gridBar, 32);
#undef ENS
	w->scanCtx.words = w->scanFlags.p; // (a grown array keeps its words: the epoch goes on)
	w->scanCtx.count = w->scanFlags.cap;
	w->scanCtx.abortWord = &w->d_state.p->c.overflow;
	if (w->h_stateCap < 12 * nb + sizeof(DState) / sizeof(float) + 4)
	{
		// (the rows of the last read-back are the host's mirror of every body it has not edited: they move along)
		ensureRows(w);
		float* old = w->h_state;
		w->h_state = nullptr;
		w->h_stateCap = 12 * nb * 2 + sizeof(DState) / sizeof(float) + 4;
		HIP_TRY(hipHostMalloc((void**)&w->h_state, w->h_stateCap * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void**)&w->d_hstate, w->h_state, 0));
		if (old)
		{
			memcpy(w->h_state, old, w->stateCount * 10 * sizeof(float));
			(void)hipHostFree(old);
		}
	}

	DW& d = w->dw;
	d.st = w->d_state.p;
	d.nBodies = (int)w->bodies.size();
	d.nProxies = (int)w->fixtures.size();
	d.nJoints = (int)w->joints.size();
	d.nShapes = (int)w->shapes.size();
	d.bigChunks = getenv("B2HIP_BIG_CHUNKS") != nullptr ? 1 : 0;
	// Exact order costs ~1 us per DEPENDENT constraint (a GPU lane against a CPU core on a chain): a 210-box pyramid is
	// ~300 levels x 12 sweeps = 3.9 ms in k_solve_small, ~0.1 ms as one block of k_solve_blocks. Islands up to 128 (bodies
	// or contacts) are walked in the reference's order, bit-exact; B2HIP_SMALL_MAX_W (<= 512) moves the line.
	d.smallMaxW = TINY_ISLAND_MAX_W;
	d.noFreeBodies = getenv("B2HIP_NO_FREE_BODIES") && atoi(getenv("B2HIP_NO_FREE_BODIES")) ? 1 : 0;
	d.hubSerial = getenv("B2HIP_HUB_SERIAL") && atoi(getenv("B2HIP_HUB_SERIAL")) ? 1 : 0;
	w->hubWaves = getenv("B2HIP_HUB_WAVES") && atoi(getenv("B2HIP_HUB_WAVES")) == 1 ? 1 : 8; // (1: the one-wave form, for comparison)
	// The end of a sweep over islands that run launch per colour - tail colours, hub rows, joints, the verdict of a position
	// iteration - in one single-workgroup launch (k_sweep_end). B2HIP_NO_SWEEP_END=1: the launches of round 4 (k_large_hub,
	// k_large_joints, k_large_pos_end); B2HIP_NO_TAIL=1: every colour a launch of its own; B2HIP_HUB_WIDE=0: the hub rows in
	// k_large_hub's order and scheme (chunks of 64) inside k_sweep_end - what the comparisons in tests/ use. Asking for a
	// number of hub waves or the serial hub sweep means k_large_hub.
	w->sweepEnd = !(getenv("B2HIP_NO_SWEEP_END") && atoi(getenv("B2HIP_NO_SWEEP_END"))) && !getenv("B2HIP_HUB_WAVES");
	w->sweepTail = w->sweepEnd && !(getenv("B2HIP_NO_TAIL") && atoi(getenv("B2HIP_NO_TAIL")));
	w->recolorSlack = getenv("B2HIP_RECOLOR_SLACK") ? atoi(getenv("B2HIP_RECOLOR_SLACK")) : 2;
	w->restFlow = w->sweepEnd && !(getenv("B2HIP_NO_REST") && atoi(getenv("B2HIP_NO_REST")));
	w->restRowsMax = getenv("B2HIP_REST_ROWS") ? std::min(atoi(getenv("B2HIP_REST_ROWS")), REST_ROWS_MAX - COLOR_SMALL_MAX) : 65536;
	w->tailRowsMax = getenv("B2HIP_TAIL_ROWS") ? atoi(getenv("B2HIP_TAIL_ROWS")) : SWEEP_END_LANES;
	d.hubWide = (w->sweepEnd && !d.hubSerial && !(getenv("B2HIP_HUB_WIDE") && atoi(getenv("B2HIP_HUB_WIDE")) == 0)) ? 1 : 0;
	if (const char* e = getenv("B2HIP_SMALL_MAX_W")) d.smallMaxW = std::max(1, std::min((int)SMALL_ISLAND_MAX_W, atoi(e)));
	d.capContacts = (int)cc;
	d.capPairs = (int)w->pairKey.cap;
	d.capMoves = (int)w->moveBuf.cap;
	d.htMask = (uint32_t)(w->ht_keys.cap - 1);
	d.gridMask = (uint32_t)(gridSize - 1);
	d.b_pos = w->b_pos.p; d.b_pos0 = w->b_pos0.p; d.b_vel = w->b_vel.p; d.b_xf = w->b_xf.p; d.b_mass = w->b_mass.p;
	d.b_damp = w->b_damp.p; d.b_force = w->b_force.p; d.b_flags = w->b_flags.p; d.b_wake = w->b_wake.p;
	d.b_order = w->b_order.p; d.orderBody = w->orderBody.p; d.bigRoots = w->bigRoots.p;
	d.p_fat = w->p_fat.p; d.p_body = w->p_body.p; d.p_shape = w->p_shape.p; d.p_key = w->p_key.p;
	d.p_filter0 = w->p_filter0.p; d.p_filter1 = w->p_filter1.p; d.p_mat = w->p_mat.p; d.shapes = w->d_shapes.p;
	for (int k = 0; k < 2; ++k)
	{
		d.ca[k].ids = w->c_ids[k].p; d.ca[k].key = w->c_key[k].p; d.ca[k].flags = w->c_flags[k].p; d.ca[k].mat = w->c_mat[k].p;
		d.ca[k].man0 = w->c_man0[k].p; d.ca[k].man1 = w->c_man1[k].p; d.ca[k].imp = w->c_imp[k].p; d.ca[k].man3 = w->c_man3[k].p;
		d.ca[k].color = w->c_color[k].p; d.ca[k].mgr = w->c_mgr[k].p;
	}
	d.ht_keys = w->ht_keys.p;
	d.joints = w->d_joints.p;
	d.gears = w->d_gears.p;
	d.jadjStart = w->jadjStart.p; d.jadj = w->jadj.p; d.rootJointStart = w->rootJointStart.p;
	d.rootJointCursor = w->rootJointCursor.p; d.lj_list = w->lj_list.p; d.rootJointOkay = w->rootJointOkay.p;
	d.parent = w->parent.p; d.rootSeed = w->rootSeed.p; d.rootBodies = w->rootBodies.p; d.rootContacts = w->rootContacts.p;
	d.rootJoints = w->rootJoints.p; d.rootScanIn = w->rootScanIn.p; d.rootScanOut = w->rootScanOut.p; d.rootIsland = w->rootIsland.p;
	d.deg = w->deg.p; d.adjStart = w->adjStart.p; d.adjCursor = w->adjCursor.p; d.adj = w->adj.p; d.adjSlot = w->adjSlot.p;
	d.si_root = w->si_root.p; d.si_bodyStart = w->si_bodyStart.p; d.si_contactStart = w->si_contactStart.p; d.si_wStart = w->si_wStart.p;
	d.si_maxLevel = w->si_maxLevel.p; d.si_bodies = w->si_bodies.p; d.si_contacts = w->si_contacts.p; d.si_level = w->si_level.p;
	d.si_stack = w->si_stack.p; d.si_lastLevel = w->si_lastLevel.p; d.b_slot = w->b_slot.p; d.b_island = w->b_island.p;
	d.chunkFirst = w->chunkFirst.p;
	d.li_bodies = w->li_bodies.p; d.li_contacts = w->li_contacts.p; d.li_roots = w->li_roots.p; d.li_color = w->li_color.p;
	d.colorCount = w->colorCount.p; d.colorStart = w->colorStart.p; d.colorCursor = w->colorCursor.p; d.li_sorted = w->li_sorted.p; d.li_ref = w->li_ref.p;
	d.bodyClaim = w->bodyClaim.p; d.bodyColorMask = w->bodyColorMask.p; d.bodyActive = w->bodyActive.p; d.bodyRest = w->bodyRest.p; d.b_posv = w->b_posv.p; d.dfRank = w->dfRank.p; d.dfInbox = w->dfInbox.p; d.evKey = w->evKey.p; d.evInfo = w->evInfo.p; d.eventsOn = w->eventsOn ? 1 : 0; d.uncolList = w->uncolList.p; d.compactList = w->compactList.p; d.hubRowOf = w->hubRowOf.p; d.hubList = w->hubList.p; d.hubDelta = w->hubDelta.p; d.hubMeta = w->hubMeta.p; d.hubFirst = w->hubFirst.p; d.lc = w->lc.p; d.rootPen = w->rootPen.p;
	d.rootDone = w->rootDone.p; d.rootSleepMin = w->rootSleepMin.p;
	d.moveBuf = w->moveBuf.p; d.gridCount = w->gridCount.p; d.gridStart = w->gridStart.p; d.gridCursor = w->gridCursor.p;
	d.gridItems = w->gridItems.p; d.gridFat = w->gridFat.p; d.arriveTree = w->arriveTree.p; d.largeProxies = w->largeProxies.p; d.largeMoves = w->largeMoves.p;
	d.pairKey = w->pairKey.p; d.pairProxy = w->pairProxy.p; d.pairKey2 = w->pairKey2.p; d.pairProxy2 = w->pairProxy2.p;
	d.pairFirst = w->pairFirst.p; d.pairRank = w->pairRank.p;
	d.scanTmp = w->scanTmp.p; d.radixHist = w->radixHist.p; d.keepFlag = w->keepFlag.p; d.keepScan = w->keepScan.p;
	d.stateOut = w->stateOut.p;
	d.b_proxyHead = w->b_proxyHead.p; d.p_next = w->p_next.p; d.toiList = w->toiList.p;
	d.toiPos2c = w->toiPos2c.p; d.toiDestroyList = w->toiDestroyList.p; d.toiNewList = w->toiNewList.p;
	d.b_toiGroup = w->b_toiGroup.p; d.toiGroups = w->toiGroups.p; d.toiGroupCount = w->toiGroupCount.p; d.toiGroupList = w->toiGroupList.p; d.toiMoved = w->toiMoved.p; d.toiNew = w->toiNew.p; d.toiParent = w->toiParent.p; d.toiDomOf = w->toiDomOf.p; d.toiDomRoot = w->toiDomRoot.p; d.toiDomCount = w->toiDomCount.p; d.toiDomBase = w->toiDomBase.p; d.toiDomFill = w->toiDomFill.p; d.toiDomFailed = w->toiDomFailed.p; d.toiDomEvents = w->toiDomEvents.p; d.toiDomList = w->toiDomList.p; d.toiHull = w->toiHull.p;
	d.snapBody = w->snapBody.p; d.snapFat = w->snapFat.p;
	d.b_blk1 = w->b_blk1.p; d.b_adopt = w->b_adopt.p; d.b_adoptStage = w->b_adoptStage.p; d.blkRows = w->blkRows.p; d.blkRowStart = w->blkRowStart.p; d.blkCursor = w->blkCursor.p; d.blkBodyCount = w->blkBodyCount.p; d.blkBodyCursor = w->blkBodyCursor.p;
	d.blkBodyStart = w->blkBodyStart.p; d.blkBodies = w->blkBodies.p; d.rowColor = w->rowColor.p; d.b_cutv = w->b_cutv.p;
	d.spatial = w->spatial ? 1 : 0; d.b_owner = w->b_owner.p; d.spNewOwner = w->spNewOwner.p; d.spAwake = w->spAwake.p; d.spFullRows = w->spFullRows ? 1 : 0; d.spStraddle = w->spStraddle.p;
	d.capStraddle = (int)w->spStraddle.cap; d.spCount = w->spCount.p; d.spTarget = w->spTarget.p; d.spTailKey = w->spTailKey.p;
	if (w->spatial && w->spOwnCapRows < nb)
	{
		if (w->spOwnHost) { HIP_TRY(hipStreamSynchronize(s)); (void)hipHostFree(w->spOwnHost); }
		w->spOwnHost = nullptr;
		w->spOwnCapRows = 2 * nb;
		HIP_TRY(hipHostMalloc((void**)&w->spOwnHost, w->spOwnCapRows * 11 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void**)&w->spOwnDev, w->spOwnHost, 0));
	}
	d.spOwnOut = w->spatial ? w->spOwnDev : nullptr; d.spOwnCap = (int)w->spOwnCapRows;
	d.userFilter = hasFilter(w) ? 1 : 0; d.preSolveOn = hasPreSolve(w) ? 1 : 0; d.postSolveOn = w->postSolveOn ? 1 : 0;
	d.pre_o0 = w->pre_o0.p; d.pre_o1 = w->pre_o1.p; d.pre_oimp = w->pre_oimp.p; d.pre_o3 = w->pre_o3.p;
	d.preRecs = w->preRecs.p; d.postRecs = w->postRecs.p; d.filterList = w->filterList.p;
	d.toiLog = listenerOn(w) && w->def.continuous ? w->toiLog.p : nullptr;
	d.capToiLog = (int)std::min<size_t>(w->toiLog.cap, 0x7fffff);
	d.toiVerdict = hasPreSolve(w) && w->def.continuous ? w->toiVerdict.p : nullptr;
	d.nToiVerdict = std::min((int)w->toiVerdicts.size(), (int)std::min<size_t>(w->toiVerdict.cap, 0x7fffff));
	return 0;
}

// Upload bodies / fixtures / shapes / joints created or edited since the last step.
static int flushEdits(b2hip_world* w)
{
	hipStream_t s = w->stream;
	int rc = ensureCapacity(w, (size_t)w->lastContacts);
	if (rc) return rc;

	// ---- bodies: every dirty body gets all its rows rewritten from the host mirror ---------------
	const size_t nb = w->bodies.size();
	std::sort(w->dirtyList.begin(), w->dirtyList.end());
	w->dirtyList.erase(std::unique(w->dirtyList.begin(), w->dirtyList.end()), w->dirtyList.end());
	std::vector<float4> pos, pos0, vel, xf, mass, damp, force;
	std::vector<uint32_t> flags;
	size_t di = 0;
	while (di < w->dirtyList.size())
	{
		const size_t i = (size_t)w->dirtyList[di];
		size_t j = i;
		pos.clear(); pos0.clear(); vel.clear(); xf.clear(); mass.clear(); damp.clear(); force.clear(); flags.clear();
		while (di < w->dirtyList.size() && (size_t)w->dirtyList[di] == j)
		{
			HostBody& b = w->bodies[j];
			pos.push_back(make_float4(b.cx, b.cy, b.a, b.sleepTime));
			pos0.push_back(make_float4(b.c0x, b.c0y, b.a0, 0.0f));
			vel.push_back(make_float4(b.vx, b.vy, b.w, 0.0f));
			xf.push_back(make_float4(b.px, b.py, b.qs, b.qc));
			mass.push_back(make_float4(b.invMass, b.invI, b.lcx, b.lcy));
			damp.push_back(make_float4(b.linearDamping, b.angularDamping, b.gravityScale, 0.0f));
			force.push_back(make_float4(b.fx, b.fy, b.torque, 0.0f));
			flags.push_back((b.flags & ~BF_TYPE_MASK) | (uint32_t)b.type);
			b.dirty = false;
			++j;
			++di;
		}
		const size_t cnt = j - i;
		HIP_TRY(hipMemcpyAsync(w->b_pos.p + i, pos.data(), cnt * sizeof(float4), hipMemcpyHostToDevice, s));
		// the sweep origin (c0, a0, alpha0) is device-owned state: only NEW bodies get it from the host mirror, an edited
		// body keeps the one the last solve left (the read-back does not carry it, and TOI needs the true one)
		if (i + cnt > w->upBodies)
		{
			const size_t first = std::max(i, w->upBodies);
			HIP_TRY(hipMemcpyAsync(w->b_pos0.p + first, pos0.data() + (first - i), (i + cnt - first) * sizeof(float4), hipMemcpyHostToDevice, s));
		}
		for (size_t k = 0; k < cnt; ++k)
		{
			// b2Body::SetTransform moves the sweep origin too (b2Body.cpp:463-467)
			if (w->bodies[i + k].resetSweep && i + k < w->upBodies)
				HIP_TRY(hipMemcpyAsync(w->b_pos0.p + i + k, pos0.data() + k, sizeof(float4), hipMemcpyHostToDevice, s));
			w->bodies[i + k].resetSweep = 0;
		}
		HIP_TRY(hipMemcpyAsync(w->b_vel.p + i, vel.data(), cnt * sizeof(float4), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->b_xf.p + i, xf.data(), cnt * sizeof(float4), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->b_mass.p + i, mass.data(), cnt * sizeof(float4), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->b_damp.p + i, damp.data(), cnt * sizeof(float4), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->b_force.p + i, force.data(), cnt * sizeof(float4), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->b_flags.p + i, flags.data(), cnt * sizeof(uint32_t), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s)); // staging vectors are reused
	}
	w->dirtyList.clear();
	w->upBodies = nb;

	// ---- island seed order (m_nonStaticBodies): rewritten whole when a non-static body was created or destroyed
	if (w->orderDirty)
	{
		std::vector<int> order(nb, 0x7fffffff);
		for (size_t k = 0; k < w->nonStatic.size(); ++k) order[(size_t)w->nonStatic[k]] = (int)k;
		HIP_TRY(hipMemcpyAsync(w->b_order.p, order.data(), nb * sizeof(int), hipMemcpyHostToDevice, s));
		if (!w->nonStatic.empty()) HIP_TRY(hipMemcpyAsync(w->orderBody.p, w->nonStatic.data(), w->nonStatic.size() * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		w->orderDirty = false;
	}

	// ---- shapes / joints: small tables, rewritten whole when they grew -----------------------------
	if (w->upShapes != w->shapes.size())
	{
		HIP_TRY(hipMemcpyAsync(w->d_shapes.p, w->shapes.data(), w->shapes.size() * sizeof(ShapeRec), hipMemcpyHostToDevice, s));
		w->upShapes = w->shapes.size();
	}
	if (w->upJoints != w->joints.size())
	{
		// new joints are appended; the device keeps the persistent impulses of the ones it already has
		const size_t first = w->upJoints, cnt = w->joints.size() - first;
		HIP_TRY(hipMemcpyAsync(w->d_joints.p + first, w->joints.data() + first, cnt * sizeof(RevoluteJoint), hipMemcpyHostToDevice, s));
		w->upJoints = w->joints.size();
	}
	if (w->upGears != w->gears.size())
	{
		const size_t first = w->upGears, cnt = w->gears.size() - first;
		HIP_TRY(hipMemcpyAsync(w->d_gears.p + first, w->gears.data() + first, cnt * sizeof(GearRec), hipMemcpyHostToDevice, s));
		w->upGears = w->gears.size();
	}
	if (w->nMouseJoints > 0)
	{
		// a mouse joint reads bodyB's mass (b2MouseJoint.cpp:110), which a fixture added later changes
		for (size_t k = 0; k < w->upJoints; ++k)
		{
			JointRec& j = w->joints[k];
			if (j.type == B2D_JOINT_MOUSE && j.bodyMass != w->bodies[j.bodyB].mass)
			{
				j.bodyMass = w->bodies[j.bodyB].mass;
				w->jointEdits.push_back(std::make_pair((int)k, 2));
			}
		}
	}
	for (size_t k = 0; k < w->jointEdits.size(); ++k)
	{
		// setters touch the six limit / motor members only (contiguous); everything else in the device record is solver state
		const int id = w->jointEdits[k].first;
		const size_t off = offsetof(JointRec, enableLimit), len = offsetof(JointRec, collideConnected) - off;
		HIP_TRY(hipMemcpyAsync((char*)(w->d_joints.p + id) + off, (const char*)&w->joints[id] + off, len, hipMemcpyHostToDevice, s));
		if (w->jointEdits[k].second == 2)
		{
			const size_t o2 = offsetof(JointRec, localAnchorA), l2 = offsetof(JointRec, enableLimit) - o2;
			HIP_TRY(hipMemcpyAsync((char*)(w->d_joints.p + id) + o2, (const char*)&w->joints[id] + o2, l2, hipMemcpyHostToDevice, s));
		}
		if (w->jointEdits[k].second == 3)
		{
			HIP_TRY(hipMemcpyAsync((char*)(w->d_joints.p + id) + offsetof(JointRec, type), &w->joints[id].type, sizeof(int), hipMemcpyHostToDevice, s));
		}
		if (w->jointEdits[k].second == 1)
		{
			static const float zero = 0.0f;
			HIP_TRY(hipMemcpyAsync((char*)(w->d_joints.p + id) + offsetof(JointRec, impulseZ), &zero, sizeof(float), hipMemcpyHostToDevice, s));
		}
	}
	w->jointEdits.clear();
	if (w->jadjBodies != w->bodies.size() || w->jadjJoints != w->joints.size())
	{
		// per-body joint edges, newest first (b2World.cpp:697-710): CSR by counting, rebuilt only when bodies or joints were added
		const size_t nbod = w->bodies.size();
		w->jadjBodies = nbod;
		w->jadjJoints = w->joints.size();
		std::vector<int> start(nbod + 1, 0), adj;
		for (size_t j = 0; j < w->joints.size(); ++j)
		{
			if (w->joints[j].type == B2D_JOINT_DEAD) continue;
			start[w->joints[j].bodyA + 1] += 1;
			if (w->joints[j].bodyB != w->joints[j].bodyA) start[w->joints[j].bodyB + 1] += 1;
		}
		for (size_t b = 0; b < nbod; ++b) start[b + 1] += start[b];
		adj.resize((size_t)start[nbod]);
		std::vector<int> cursor(start.begin(), start.end() - 1);
		for (int j = (int)w->joints.size() - 1; j >= 0; --j)
		{
			if (w->joints[j].type == B2D_JOINT_DEAD) continue;
			adj[(size_t)cursor[w->joints[j].bodyA]++] = j;
			if (w->joints[j].bodyB != w->joints[j].bodyA) adj[(size_t)cursor[w->joints[j].bodyB]++] = j;
		}
		HIP_TRY(hipMemcpyAsync(w->jadjStart.p, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice, s));
		if (!adj.empty()) HIP_TRY(hipMemcpyAsync(w->jadj.p, adj.data(), adj.size() * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
	}

	// ---- new proxies -----------------------------------------------------------------------------
	const size_t np = w->fixtures.size();
	if (w->proxyListsStale && w->upFixtures == np && np > 0)
	{
		// a fixture was destroyed: the per-body proxy lists (newest first) are rebuilt without it
		std::vector<int> head(w->bodies.size(), -1), next(np, -1);
		for (size_t k = 0; k < np; ++k)
		{
			if (w->fixtures[k].dead || w->fixtures[k].noProxy) continue;
			const int b = w->fixtures[k].body;
			next[k] = head[b];
			head[b] = (int)k;
		}
		HIP_TRY(hipMemcpyAsync(w->b_proxyHead.p, head.data(), head.size() * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_next.p, next.data(), np * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		w->proxyListsStale = false;
	}
	if (w->upFixtures < np)
	{
		const size_t first = w->upFixtures, cnt = np - first;
		std::vector<float4> fat(cnt);
		std::vector<int> body(cnt), shape(cnt), key(cnt), f1(cnt);
		std::vector<uint32_t> f0(cnt);
		std::vector<float2> mat(cnt);
		for (size_t k = 0; k < cnt; ++k)
		{
			const HostFixture& f = w->fixtures[first + k];
			fat[k] = make_float4(f.fat[0], f.fat[1], f.fat[2], f.fat[3]);
			body[k] = (f.dead || f.noProxy) ? -1 : f.body;
			shape[k] = f.shape;
			key[k] = f.proxyKey;
			f0[k] = (uint32_t)f.categoryBits | ((uint32_t)f.maskBits << 16);
			f1[k] = ((int)(uint16_t)f.groupIndex) | (f.isSensor ? PF_SENSOR : 0) | (f.thick ? PF_THICK : 0);
			mat[k] = make_float2(f.friction, f.restitution);
		}
		HIP_TRY(hipMemcpyAsync(w->p_fat.p + first, fat.data(), cnt * sizeof(float4), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_body.p + first, body.data(), cnt * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_shape.p + first, shape.data(), cnt * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_key.p + first, key.data(), cnt * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_filter0.p + first, f0.data(), cnt * sizeof(uint32_t), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_filter1.p + first, f1.data(), cnt * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_mat.p + first, mat.data(), cnt * sizeof(float2), hipMemcpyHostToDevice, s));
		// per-body proxy lists, newest first like b2Body::m_fixtureList (b2Body.cpp:203-205)
		std::vector<int> head(w->bodies.size(), -1), next(np, -1);
		for (size_t k = 0; k < np; ++k)
		{
			if (w->fixtures[k].dead || w->fixtures[k].noProxy) continue;
			const int b = w->fixtures[k].body;
			next[k] = head[b];
			head[b] = (int)k;
		}
		HIP_TRY(hipMemcpyAsync(w->b_proxyHead.p, head.data(), head.size() * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(w->p_next.p, next.data(), np * sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		w->upFixtures = np;
		w->proxyListsStale = false;

		// Broad-phase cell: 1.5 x the largest fat extent among non-static proxies, ignoring outliers
		// (> 8 x median), which are handled by the brute-force "large proxy" path.
		std::vector<float> ext;
		for (size_t k = 0; k < np; ++k)
		{
			const HostFixture& f = w->fixtures[k];
			if (f.dead || f.noProxy || w->bodies[f.body].type == B2HIP_STATIC_BODY) continue;
			ext.push_back(std::max(f.fat[2] - f.fat[0], f.fat[3] - f.fat[1]));
		}
		float cell = 1.0f;
		if (!ext.empty())
		{
			std::sort(ext.begin(), ext.end());
			float median = ext[ext.size() / 2];
			float mx = median;
			for (size_t k = 0; k < ext.size(); ++k)
			{
				if (ext[k] <= 8.0f * median) mx = std::max(mx, ext[k]);
			}
			cell = 1.5f * mx;
		}
		w->dw.cellSize = cell;
		w->dw.invCellSize = 1.0f / cell;
	}

	// ---- edited proxies of fixtures the device already has: fat AABB (SetTransform), filter words (SetFilterData,
	// SetSensor, SetThickShape), owner (-1: the fixture was destroyed)
	if (!w->proxyEdits.empty())
	{
		std::sort(w->proxyEdits.begin(), w->proxyEdits.end());
		w->proxyEdits.erase(std::unique(w->proxyEdits.begin(), w->proxyEdits.end()), w->proxyEdits.end());
		for (size_t k = 0; k < w->proxyEdits.size(); ++k)
		{
			const int id = w->proxyEdits[k];
			if ((size_t)id >= w->upFixtures) continue; // (a new fixture: uploaded whole above)
			const HostFixture& f = w->fixtures[id];
			const float4 fat = make_float4(f.fat[0], f.fat[1], f.fat[2], f.fat[3]);
			const int body = (f.dead || f.noProxy) ? -1 : f.body;
			const int key = f.proxyKey;
			const uint32_t f0 = (uint32_t)f.categoryBits | ((uint32_t)f.maskBits << 16);
			const int f1 = ((int)(uint16_t)f.groupIndex) | (f.isSensor ? PF_SENSOR : 0) | (f.thick ? PF_THICK : 0);
			// (the fat AABB of an uploaded fixture is device state: the host copy is only current if SetTransform wrote it)
			if (std::find(w->fatEdits.begin(), w->fatEdits.end(), id) != w->fatEdits.end())
				HIP_TRY(hipMemcpy(w->p_fat.p + id, &fat, sizeof(float4), hipMemcpyHostToDevice));
			HIP_TRY(hipMemcpy(w->p_body.p + id, &body, sizeof(int), hipMemcpyHostToDevice));
			HIP_TRY(hipMemcpy(w->p_key.p + id, &key, sizeof(int), hipMemcpyHostToDevice)); // (a re-activated body's proxies have new ids)
			HIP_TRY(hipMemcpy(w->p_filter0.p + id, &f0, sizeof(uint32_t), hipMemcpyHostToDevice));
			HIP_TRY(hipMemcpy(w->p_filter1.p + id, &f1, sizeof(int), hipMemcpyHostToDevice));
			const float2 mat = make_float2(f.friction, f.restitution); // (b2Fixture::SetFriction / SetRestitution: for contacts created from now on)
			HIP_TRY(hipMemcpy(w->p_mat.p + id, &mat, sizeof(float2), hipMemcpyHostToDevice));
		}
		w->proxyEdits.clear();
		w->fatEdits.clear();
	}

	// ---- move buffer: proxies created since the last step (b2BroadPhase::CreateProxy buffers a move)
	if (!w->pendingMoves.empty())
	{
		rc = readState(w);
		if (rc) return rc;
		int have = w->h_dstate->c.nMoves;
		HIP_TRY(hipMemcpyAsync(w->moveBuf.p + have, w->pendingMoves.data(), w->pendingMoves.size() * sizeof(int), hipMemcpyHostToDevice, s));
		int total = have + (int)w->pendingMoves.size();
		HIP_TRY(hipMemcpyAsync(&w->d_state.p->c.nMoves, &total, sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		w->pendingMoves.clear();
	}
	// scan lengths that live on the device: uploaded when they change, not every step
	const int consts[3] = { (int)w->bodies.size(), (int)(w->dw.gridMask + 1), (int)w->bodies.size() + 1 };
	if (consts[0] != w->constsUploaded[0] || consts[1] != w->constsUploaded[1] || w->consts.p != w->constsUploadedAt)
	{
		HIP_TRY(hipMemcpyAsync(w->consts.p, consts, sizeof(int) * 2, hipMemcpyHostToDevice, s));
		// [4]: scan length of the TOI adjacency (nBodies + 1 so that adjStart[nBodies] is the total)
		HIP_TRY(hipMemcpyAsync(w->consts.p + 4, &consts[2], sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		w->constsUploaded[0] = consts[0];
		w->constsUploaded[1] = consts[1];
		w->constsUploadedAt = w->consts.p;
	}
	return 0;
}

// Flags the contacts between the bodies of every joint created or destroyed since the last step for re-filtering
// (b2World.cpp:716-732, 833-845): one upload of the sorted pair keys, one launch. Called by the step and by the snapshot
// (so that a snapshot taken right after CreateJoint / DestroyJoint carries the flags).
static int applyPendingFilters(b2hip_world* w)
{
	if (w->pendingFilter.empty()) return 0;
	std::vector<unsigned long long> keys(w->pendingFilter.size());
	for (size_t k = 0; k < keys.size(); ++k)
	{
		const unsigned a = (unsigned)std::min(w->pendingFilter[k].first, w->pendingFilter[k].second);
		const unsigned b = (unsigned)std::max(w->pendingFilter[k].first, w->pendingFilter[k].second);
		keys[k] = ((unsigned long long)a << 32) | b;
	}
	std::sort(keys.begin(), keys.end());
	keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
	int rc = w->filterPairs.ensure(keys.size(), w->stream, false, false);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(w->filterPairs.p, keys.data(), keys.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, w->stream));
	LAUNCH(w, k_flag_filter, gridFor(w->dw.capContacts), 256, w->dw, w->filterPairs.p, (int)keys.size());
	HIP_TRY(hipStreamSynchronize(w->stream)); // `keys` is pageable host memory
	w->pendingFilter.clear();
	w->refilterPending = true;
	return 0;
}

// Applies the queued contact-array ops (b2d_kernels_edit.h) in call order, then compacts the contact array if contacts
// were destroyed. Called by the step right after its counters are zeroed, and by whoever reads the contacts between steps.
static int downloadState(b2hip_world* w, int clearForces, bool skipRowsIfRedo = false);
static int startEarlyRows(b2hip_world* w);

static int applyEditOps(b2hip_world* w, bool betweenSteps)
{
	if (w->editOps.empty()) return 0;
	bool destroys = false;
	for (size_t k = 0; k < w->editOps.size(); ++k) destroys = destroys || w->editOps[k].x == EDIT_DESTROY_BODY || w->editOps[k].x == EDIT_DESTROY_FIXTURE;
	int rc = w->d_editOps.ensure(w->editOps.size(), w->stream, false, false);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(w->d_editOps.p, w->editOps.data(), w->editOps.size() * sizeof(int2), hipMemcpyHostToDevice, w->stream));
	DW& d = w->dw;
	LAUNCH(w, k_apply_edits, 1, 1024, d, (const int2*)w->d_editOps.p, (int)w->editOps.size());
	if (destroys)
	{
		LAUNCH(w, k_edit_keepflags, gridFor(d.capContacts), 256, d);
		deviceExclusiveScan<int>(w->stream, d.keepFlag, d.keepScan, d.scanTmp, w->scanCtx, &d.st->c.nContacts, d.capContacts);
		LAUNCH(w, k_compact_contacts, gridFor(d.capContacts), 256, d); // (its last workgroup switches the buffers)
		LAUNCH(w, k_edit_finish, 1, 1, d);
		if (betweenSteps)
		{
			// destroying a touching contact wakes its bodies (b2Contact::Destroy, b2Contact.cpp:105-111): the host rows are
			// read again from the device (every edit made so far has been uploaded by the caller)
			rc = downloadState(w, 0);
			if (rc) return rc;
			w->stateCount = w->bodies.size();
			++w->mirrorEpoch;
		}
	}
	rc = readState(w); // (also makes the staging vector reusable, and the state rows above readable)
	if (rc) return rc;
	w->editOps.clear();
	w->lastContacts = w->h_dstate->c.nContacts;
	w->last.nContacts = w->lastContacts;
	return 0;
}

// ------------------------------------------------------------------------------------------------
// Phases
// ------------------------------------------------------------------------------------------------
// spatially sharded worlds (defined behind the RCCL section; b2d_kernels_spatial.h)
static int spExchangeState(b2hip_world* w, int mode);
static int spExchangePairs(b2hip_world* w, long long* totalPairs, int* straddling);
static int spResolve(b2hip_world* w, int nVirt = 0);
static int spAfterToi(b2hip_world* w);
static int spBeginStep(b2hip_world* w);

static int radixBits(int maxKey)
{
	int bits = 1;
	while ((1 << bits) <= maxKey && bits < 31) ++bits;
	return bits;
}

// Uploads `list` and runs `kernel(d, list, count)` (contacts to disable / reject, candidate pairs to drop)
template <typename K>
static int applyHostList(b2hip_world* w, K kernel, const std::vector<int>& list)
{
	if (list.empty()) return 0;
	HIP_TRY(hipMemcpyAsync(w->hostList.p, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, w->stream));
	LAUNCH(w, kernel, gridFor(list.size()), 256, w->dw, (const int*)w->hostList.p, (int)list.size());
	HIP_TRY(hipStreamSynchronize(w->stream));
	return 0;
}

// b2ContactManager::AddPair's user filter (b2ContactManager.cpp:283-287): the first occurrence of every candidate pair is
// shown to the user's b2hip_should_collide_fn (lower proxy id first, as AddPair passes them); refused pairs stop being
// first occurrences, so nothing is created for them. Between the "first" flags and the ranks of either ordering path.
// the user's filter on a list of fixture pairs: one call with all of them (batch form) or one call per pair
static void askFilter(b2hip_world* w, const std::vector<int>& pairs2, std::vector<int>& verdict)
{
	const int n = (int)pairs2.size() / 2;
	verdict.assign((size_t)n, 1);
	if (n == 0) return;
	if (w->filterBatchFn) w->filterBatchFn(w->filterUser, n, pairs2.data(), verdict.data());
	else for (int i = 0; i < n; ++i) verdict[i] = w->filterFn(w->filterUser, pairs2[2 * i], pairs2[2 * i + 1]) ? 1 : 0;
}

static int userFilterPairs(b2hip_world* w, const int2* proxies)
{
	int rc = readState(w);
	if (rc) return rc;
	const int n = std::min(w->h_dstate->c.nPairs, w->dw.capPairs);
	if (n <= 0 || (w->h_dstate->c.overflow & 3)) return 0;
	std::vector<int> first(n);
	std::vector<int2> pr(n);
	HIP_TRY(hipMemcpy(first.data(), w->pairFirst.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(pr.data(), proxies, (size_t)n * sizeof(int2), hipMemcpyDeviceToHost));
	std::vector<int> refused, asked, which, verdict;
	for (int i = 0; i < n; ++i)
	{
		if (!first[i]) continue;
		asked.push_back(pr[i].x);
		asked.push_back(pr[i].y);
		which.push_back(i);
	}
	askFilter(w, asked, verdict);
	for (size_t k = 0; k < which.size(); ++k) if (!verdict[k]) refused.push_back(which[k]);
	return applyHostList(w, k_pairs_reject, refused);
}

// b2World::FindNewContacts. `sync` = the host may block on the pair count to pick the sort path
// (top-of-step call after fixtures were added); otherwise the small path runs optimistically and
// the caller checks Counters::nPairs at the end-of-step read-back.
// knownPairs: the candidate pairs in the buffer where the host has just read the count (-1: unknown) - the radix passes then
// launch and scan for that many tiles, not for the buffer's capacity (a rank of a sharded world keeps buffers of the whole
// world's size: eleven scans per step took the three-kernel form for 25 000 pairs)
static int runSortAndCreate(b2hip_world* w, bool largePath, long long knownPairs = -1)
{
	DW& d = w->dw;
	const uint64_t* sortedKeys = d.pairKey;
	const int2* sortedProxies = d.pairProxy;
	if (largePath)
	{
		// LSD radix sort on the two key halves
		int bits = radixBits(w->nextNode + 1);
		std::vector<int> shifts;
		for (int sft = 0; sft < bits; sft += 8) shifts.push_back(sft);
		for (int sft = 0; sft < bits; sft += 8) shifts.push_back(32 + sft);
		uint64_t* kin = d.pairKey;
		uint64_t* kout = d.pairKey2;
		int2* vin = d.pairProxy;
		int2* vout = d.pairProxy2;
		int tilesCap = d.capPairs / RADIX_TILE + 1;
		if (knownPairs >= 0) tilesCap = (int)std::min<long long>(tilesCap, knownPairs / RADIX_TILE + 2);
		// (the length of the histogram matrix depends on the pair count only: once per sort, not once per pass)
		LAUNCH(w, k_radix_count, 1, 1, &d.st->c.nPairs, 0, w->consts.p + 2);
		for (size_t p = 0; p < shifts.size(); ++p)
		{
			LAUNCH(w, k_radix_hist, tilesCap, RADIX_THREADS, kin, d.radixHist, &d.st->c.nPairs, 0, shifts[p], tilesCap);
			deviceExclusiveScan<int>(w->stream, d.radixHist, w->radixHistScan.p, d.scanTmp, w->scanCtx, w->consts.p + 2, 256 * tilesCap);
			LAUNCH(w, k_radix_scatter, tilesCap, RADIX_THREADS, kin, vin, kout, vout, w->radixHistScan.p, &d.st->c.nPairs, 0, shifts[p]);
			std::swap(kin, kout);
			std::swap(vin, vout);
		}
		sortedKeys = kin;
		sortedProxies = vin;
		LAUNCH(w, k_pairs_sorted_first, gridFor(d.capPairs), 256, d, sortedKeys, w->consts.p + 3);
		if (hasFilter(w)) { int rcf = userFilterPairs(w, sortedProxies); if (rcf) return rcf; }
		deviceExclusiveScan<int>(w->stream, d.pairFirst, d.pairRank, d.scanTmp, w->scanCtx, w->consts.p + 3,
			knownPairs >= 0 ? (int)std::min<long long>(d.capPairs, knownPairs + 2) : d.capPairs);
		LAUNCH(w, k_pairs_sorted_total, 1, 1, d, w->consts.p + 3);
	}
	else
	{
		LAUNCH(w, k_pairs_first, 16, 256, d);
		if (hasFilter(w)) { int rcf = userFilterPairs(w, sortedProxies); if (rcf) return rcf; }
		LAUNCH(w, k_pairs_rank, 16, 256, d);
	}
	const int smallPath = largePath ? 0 : 1;
	LAUNCH(w, k_create_contacts, gridFor(largePath ? d.capPairs : COUNT_RANK_MAX), 256, d, sortedKeys, sortedProxies, smallPath);
	LAUNCH(w, k_create_finish, gridFor(d.nBodies), 256, d, smallPath);
	LAUNCH(w, k_toi_order_create, 1, 1024, d, smallPath); // (+ the commit of the update)
	return 0;
}

static int findNewContacts(b2hip_world* w, bool sync);
static int findNewContactsOnce(b2hip_world* w, bool sync);
static int findNewContactsGraph(b2hip_world* w)
{
	// (a user contact filter is asked on the host in the middle of the update: synchronous, no graph)
	if (hasFilter(w) || w->spatial) return findNewContacts(w, true);
	// A scene that creates more pairs per step than the optimistic counting path ranks (the settled 50 086-box pyramid and the
	// 100 000-box Tumbler: ~25 000 and ~150 000 new fat-AABB pairs per step) would find that out at the end of the step, sort
	// with the radix path then - and run the TOI phase and the read-back a second time, every step. While that has happened
	// lately the host looks at the pair count right after the search instead (one small read-back) and takes the right path.
	if (w->pairsLargeSticky > 0) return findNewContacts(w, true);
	return runSegment(w, w->segPairs, 3, [w]() -> int { return findNewContacts(w, false); });
}

// The pair finder met more candidate pairs than the buffer holds (a dense start: every proxy is "moved" and overlaps dozens
// of others). Counters::nPairs counted all of them: size the buffers for that and let the caller run the search again
// (nothing was consumed: the creation kernels leave an overflowed set alone and the moves stay buffered).
static int growPairBuffers(b2hip_world* w)
{
	const Counters& c = w->h_dstate->c;
	if (c.overflow & 2) w->pairCapHint = 2 * (size_t)c.nPairs + 4096;
	// (bit 0: the new contacts did not fit the contact array - creation was skipped as a whole, see createBlocked)
	int rc = ensureCapacity(w, (size_t)c.nContacts + (size_t)std::max(c.nNewContacts, 0) + 1024);
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(&w->d_state.p->c.overflow, 0, sizeof(int), w->stream));
	return 0;
}

static int findNewContacts(b2hip_world* w, bool sync)
{
	for (int attempt = 0; sync && attempt < 4; ++attempt)
	{
		int rc = findNewContactsOnce(w, true);
		if (rc != 1) return rc; // 1 = pair buffer overflow, buffers grown: search again
	}
	if (sync) return setError(B2HIP_ERR_CAPACITY, "pair buffer overflow");
	return findNewContactsOnce(w, false);
}

static int findNewContactsOnce(b2hip_world* w, bool sync)
{
	DW& d = w->dw;
	LAUNCH(w, k_bp_clear, gridFor(std::max(d.htMask, d.gridMask) + 1), 256, d);
	LAUNCH(w, k_bp_build, gridFor(std::max(d.capContacts, d.nProxies)), 256, d);
	deviceExclusiveScan<int>(w->stream, d.gridCount, d.gridStart, d.scanTmp, w->scanCtx, w->consts.p + 1, (int)(d.gridMask + 1));
	LAUNCH(w, k_grid_fill, gridFor(d.nProxies), 256, d, 0);
	if (int rk = ktBracket(w, 4, 7)) return rk;
	if (d.gridHalf) LAUNCH(w, k_find_pairs_window, gridFor((size_t)d.capMoves * 64, 256, 2048), 256, d);
	else LAUNCH(w, k_find_pairs_small, gridFor((size_t)d.capMoves * 64, 256, 2048), 256, d);
	if (int rk = ktBracket(w, 4, 7)) return rk;
	LAUNCH(w, k_find_pairs_large, 1024, 256, d);
	bool large = false;
	if (w->spatial)
	{
		// E2: every rank searched for the proxies ITS bodies moved; all ranks order and create the union. The headers of the
		// slabs tell every host what it needs to go on (one synchronisation): the size of the union -> the ordering path, and
		// whether a new pair joins bodies of different owners
		long long total = 0;
		int straddle = 0;
		int rc = spExchangePairs(w, &total, &straddle);
		if (rc) return rc;
		large = total > COUNT_RANK_MAX;
		rc = runSortAndCreate(w, large, total);
		if (rc) return rc;
		if (straddle == 0)
		{
			// CF_FOREIGN of the new contacts (nothing straddles: no resolution, nothing to read back)
			HIP_TRY(hipMemsetAsync(&w->d_state.p->c.nStraddle, 0, 4 * sizeof(int), w->stream));
			LAUNCH(w, k_sp_flag_contacts, gridFor(d.capContacts), 256, d);
			return 0;
		}
		// E3: a new contact joins components of different owners
		return spResolve(w);
	}
	if (sync)
	{
		int rc = startEarlyRows(w); // (before the host waits for the pair count)
		if (rc) return rc;
		rc = readState(w);
		if (rc) return rc;
		if (w->h_dstate->c.overflow & 3)
		{
			rc = growPairBuffers(w);
			return rc ? rc : 1;
		}
		if (w->h_dstate->c.nMoves == 0) { if (w->pairsLargeSticky > 0) w->pairsLargeSticky -= 1; return 0; }
		large = w->h_dstate->c.nPairs > COUNT_RANK_MAX;
		if (large) w->pairsLargeSticky = 16;
		else if (w->pairsLargeSticky > 0) w->pairsLargeSticky -= 1;
	}
	return runSortAndCreate(w, large, sync ? (long long)w->h_dstate->c.nPairs : -1);
}

static int phaseCollide(b2hip_world* w)
{
	return runSegment(w, w->segCollide, 1 + (w->dw.preSolveOn ? 32 : 0), [w]() -> int
	{
		DW& d = w->dw;
		if (int rk = ktBracket(w, 2, 5)) return rk;
		// (shape records staged through LDS where the world holds many distinct ones: b2d_kernels_collide.h)
		// Measured (tools/gpu_collide_variants.py, profiles/r04_collide_variants.txt): on the 1 M-body field (a record per body,
		// circles / boxes / n-gons mixed) staging + sorting a tile by shape-pair class 133 -> 124 us; on the 100 000-box Tumbler
		// (one record, one class) the sort costs 2 % - so both follow the number of distinct records unless the environment says otherwise.
		const bool many = w->shapes.size() > 4096;
		const bool stage = w->collideStage < 0 ? many : w->collideStage != 0;
		const int sort = w->collideSortEnv < 0 ? (many ? 1 : 0) : (w->collideSortEnv != 0 ? 1 : 0);
		if (stage) LAUNCH(w, k_collide<1>, gridFor(d.capContacts), 256, d, sort);
		else LAUNCH(w, k_collide<0>, gridFor(d.capContacts), 256, d, sort);
		if (int rk = ktBracket(w, 2, 5)) return rk;
		deviceExclusiveScan<int>(w->stream, d.keepFlag, d.keepScan, d.scanTmp, w->scanCtx, &d.st->c.nContacts, d.capContacts);
		if (d.preSolveOn) LAUNCH(w, k_presolve_gather, gridFor(d.capContacts), 256, d);
		LAUNCH(w, k_compact_contacts, gridFor(d.capContacts), 256, d); // (its last workgroup switches the buffers)
		return 0;
	});
}

int b2hip_debug_hash(b2hip_world* w, int which, uint64_t* out);
static void tracePoint(b2hip_world* w, const char* label)
{
	if (!w->debugTrace) return;
	uint64_t hb = 0, hi = 0;
	(void)b2hip_debug_hash(w, 0, &hb);
	(void)b2hip_debug_hash(w, 3, &hi);
	w->trace.push_back(std::make_pair(std::string(label), hb ^ (hi * 0x9E3779B97F4A7C15ull)));
}
#define TRACE(label) tracePoint(w, label)

// New block partition of the large-island bodies (b2d_kernels_solve_blocks.h): sort by Morton cell, cut the sorted sequence
// where the contact degrees add up to `targetDeg`, then look at the colours and the census again (classes have changed).
static int partitionLargeIslands(b2hip_world* w, int targetDeg)
{
	DW& d = w->dw;
	HIP_TRY(hipMemcpyAsync(&w->d_state.p->c.blkTargetDeg, &targetDeg, sizeof(int), hipMemcpyHostToDevice, w->stream));
	uint64_t* kin = d.pairKey;
	uint64_t* kout = d.pairKey2;
	int2* vin = d.pairProxy;
	int2* vout = d.pairProxy2;
	LAUNCH(w, k_part_keys, gridFor(d.nBodies), 256, d, kin, vin);
	// LSD radix sort: the body-id bits, then the 32 Morton bits (the pair buffers hold at least 8 entries per proxy)
	std::vector<int> shifts;
	const int idBits = radixBits(d.nBodies + 1);
	for (int sft = 0; sft < idBits; sft += 8) shifts.push_back(sft);
	for (int sft = 0; sft < 32; sft += 8) shifts.push_back(32 + sft);
	const int tilesCap = d.capPairs / RADIX_TILE + 1;
	const int* nPtr = &d.st->c.nLBodies;
	LAUNCH(w, k_radix_count, 1, 1, nPtr, 0, w->consts.p + 2);
	for (size_t p = 0; p < shifts.size(); ++p)
	{
		LAUNCH(w, k_radix_hist, tilesCap, RADIX_THREADS, kin, d.radixHist, nPtr, 0, shifts[p], tilesCap);
		deviceExclusiveScan<int>(w->stream, d.radixHist, w->radixHistScan.p, d.scanTmp, w->scanCtx, w->consts.p + 2, 256 * tilesCap);
		LAUNCH(w, k_radix_scatter, tilesCap, RADIX_THREADS, kin, vin, kout, vout, w->radixHistScan.p, nPtr, 0, shifts[p]);
		std::swap(kin, kout);
		std::swap(vin, vout);
	}
	LAUNCH(w, k_part_weights, gridFor(d.nBodies), 256, d, vin, d.pairFirst);
	deviceExclusiveScan<int>(w->stream, d.pairFirst, d.pairRank, d.scanTmp, w->scanCtx, nPtr, d.nBodies);
	LAUNCH(w, k_part_assign, gridFor(d.nBodies), 256, d, vin, d.pairRank);
	LAUNCH(w, k_color_recheck_begin, gridFor(d.nBodies), 256, d);
	LAUNCH(w, k_color_check, gridFor(d.capContacts), 256, d);
			LAUNCH(w, k_color_masks, gridFor(d.nBodies), 256, d);
	LAUNCH(w, k_block_census, 1, 1024, d, (DState*)nullptr);
	return 0;
}

// b2World::Solve (b2World.cpp:1166-1431): island build, census read-back, then the solver tier of each island.
static int phaseSolve(b2hip_world* w)
{
	w->trace.clear();
	w->blocksThisStep = false;
	DW& d = w->dw;
	d.serialOrphans = w->serialOrphansNext;
	const StepParams& sp = w->sp;
	if (w->kernelTiming <= 1)
	{
		// (the event pairs of the solver kernels belong to this phase; those of k_collide / k_sync_fixtures / k_find_pairs_small -
		// timing modes 2 to 4 - are taken in other phases of the step and cleared by b2hip_step_begin)
		w->ktUsed = 0;
		w->ktKind = 0;
	}
	const int forceLarge = w->forceLarge;
	// (the step parameters are kernel arguments of k_island_classify - it steps the free bodies - so a captured segment is
	// only replayed for the same ones)
	uint64_t spHash = 1469598103934665603ull;
	for (size_t k = 0; k < sizeof(StepParams); ++k) spHash = (spHash ^ ((const unsigned char*)&sp)[k]) * 1099511628211ull;
	const bool largeHint = w->largeHintSteps > 0;
	// the island build ends with the publication of its census (b2dPublishCensus): by k_block_census, by k_island_edges when
	// that is the last kernel, else by a launch of its own. B2HIP_NO_CENSUS_POLL=1: copy + stream synchronisation instead.
	const bool poll = !w->noCensusPoll;
	const bool adopt = w->adoptPasses;
	const int pubBy = !poll ? 0 : (largeHint ? 1 : ((d.nJoints == 0 && !adopt) ? 2 : 3));
	int rc = runSegment(w, w->segIslands, (2 + 16ull * (uint64_t)forceLarge + (largeHint ? 8ull : 0ull) + 64ull * (uint64_t)pubBy + (adopt ? 512ull : 0ull)) ^ (spHash << 12), [w, forceLarge, sp, largeHint, pubBy, adopt]() -> int
	{
		DW& d = w->dw;
		LAUNCH(w, k_island_init, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_island_union, gridFor(d.capContacts), 256, d);
		LAUNCH(w, k_island_flatten, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_island_count, gridFor(d.capContacts), 256, d);
		LAUNCH(w, k_island_classify, gridFor(d.nBodies), 256, d, forceLarge, sp);
		if (d.shardCount > 1 && !d.spatial) LAUNCH(w, k_shard_big, 1, 1024, d); // the big islands of a sharded world, dealt over the ranks
		{
			int blocks = (d.nBodies + SCAN_TILE - 1) / SCAN_TILE;
			if (blocks < 1) blocks = 1;
			deviceExclusiveScan<int4>(w->stream, d.rootScanIn, d.rootScanOut, w->scanTmp4.p, w->scanCtx, w->consts.p, d.nBodies);
			if (hipError_t le = hipGetLastError()) return setError(B2HIP_ERR_HIP, std::string("k_scan<int4> launch: ") + hipGetErrorString(le));
		}
		deviceExclusiveScan<int>(w->stream, d.deg, d.adjStart, d.scanTmp, w->scanCtx, w->consts.p, d.nBodies);
		if (d.nJoints > 0)
		{
			deviceExclusiveScan<int>(w->stream, d.rootJoints, d.rootJointStart, d.scanTmp, w->scanCtx, w->consts.p, d.nBodies);
		}
		LAUNCH(w, k_island_assign, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_island_edges, gridFor(d.capContacts), 256, d, pubBy == 2 ? w->d_pub : (DState*)nullptr);
		// (a growing pile: hand home blocks on to newcomers up to four contacts away instead of partitioning again)
		if (adopt)
			for (int stage = 0; stage < 3; ++stage) LAUNCH(w, k_block_adopt, gridFor(d.capContacts), 256, d, stage);
		if (d.nJoints > 0) LAUNCH(w, k_joints_fill, gridFor(d.nJoints), 256, d);
		// (colour bookkeeping and block census only matter to large islands: skipped while the world has had none lately)
		if (largeHint)
		{
			LAUNCH(w, k_color_check, gridFor(d.capContacts), 256, d);
			LAUNCH(w, k_color_masks, gridFor(d.nBodies), 256, d);
			LAUNCH(w, k_block_census, 1, 1024, d, pubBy == 1 ? w->d_pub : (DState*)nullptr);
		}
		if (pubBy == 3) LAUNCH(w, k_publish_census, 1, 256, d, w->d_pub);
		return 0;
	});
	if (rc) return rc;

	// the host needs the island census to size the solver launches
	bool colorSmallQueued = false;
	if (poll)
	{
		w->pubSeq = (w->pubSeq + 1) & 0x3fffffff; // (the device counts its publications the same way: DState::pubCount)
		// what the host would launch next in the usual case (a few new contacts on a settled pile to colour, a colour class to
		// compact) goes behind the census at once and runs while the host is busy with it; the kernel looks at the same
		// counters and returns if the case is another one
		if (largeHint && forceLarge != 2) { LAUNCH(w, k_color_small, 1, 1024, d, 1); colorSmallQueued = true; }
		rc = awaitCensus(w);
		if (rc == 0 && !b2dPartitionSettled(w->h_dstate->c)) colorSmallQueued = false; // (it saw the same and returned)
	}
	else rc = readState(w);
	if (rc) return rc;
	Counters c = w->h_dstate->c;
	if (c.nLIslands > 0)
	{
		if (!largeHint)
		{
			// the first large island after a while: run what was skipped, look again
			LAUNCH(w, k_color_check, gridFor(d.capContacts), 256, d);
			LAUNCH(w, k_color_masks, gridFor(d.nBodies), 256, d);
			LAUNCH(w, k_block_census, 1, 1024, d, (DState*)nullptr);
			rc = readState(w);
			if (rc) return rc;
			c = w->h_dstate->c;
		}
		w->largeHintSteps = 120;
	}
	else if (w->largeHintSteps > 0) w->largeHintSteps -= 1;
	// (newcomers without a home block: from the next step on k_block_adopt hands blocks further, for a while)
	if (c.nOrphanRows > 0) w->adoptSticky = 16; else if (w->adoptSticky > 0) w->adoptSticky -= 1;
	w->adoptPasses = w->adoptSticky > 0;
	const bool plainIslands = d.nJoints == 0 && c.maxDegree <= HUB_DEGREE;
	// ---- large islands that no block solver can take: more constraints than the blocks that fit the device together hold
	// (1024-lane blocks of ~750 rows: ~190 000; the settled 100 000-box Tumbler has 350 000). They run launch per colour
	// whatever happens - and a partition would only cost them: it splits the colours into two ranges (interior / cut), 27
	// colours in use where one range needs 18, and every colour is a launch of every sweep (Tumbler: 6.1 -> 5.2 ms per step).
	// So the partition is dissolved (all constraints are one class again, coloured afresh once) until the islands have shrunk.
	{
		const int cap1024 = plainIslands ? w->blocksMaxWG : w->sweepMaxWG[2];
		const bool was = w->blocksTooBig;
		if (!w->blocksTooBig && cap1024 > 0 && c.nLContacts > 800 * cap1024) w->blocksTooBig = true;
		else if (w->blocksTooBig && c.nLContacts < 650 * cap1024) w->blocksTooBig = false;
		if (w->blocksTooBig && c.nBlocks > 0 && forceLarge != 2 && !w->noBlocks)
		{
			if (w->tracePartition) fprintf(stderr, "[b2hip] partition dissolved: %d constraints in large islands, %d blocks of %d lanes (room for %d)\n", c.nLContacts, c.nBlocks, c.blkLanes, cap1024);
			const size_t nbAll = w->bodies.size();
			HIP_TRY(hipMemsetAsync(w->b_blk1.p, 0, nbAll * sizeof(int), w->stream));
			HIP_TRY(hipMemsetAsync(w->b_adopt.p, 0, nbAll * sizeof(int), w->stream));
			HIP_TRY(hipMemsetAsync(w->b_adoptStage.p, 0, 3 * nbAll * sizeof(int), w->stream));
			HIP_TRY(hipMemsetAsync(&w->d_state.p->c.nBlocks, 0, sizeof(int), w->stream));
			d.serialOrphans = 0; // (no body has a home block now: nothing is an orphan)
			// the colour census again, under the one class (as after a new partition), then every colour afresh
			LAUNCH(w, k_color_recheck_begin, gridFor(d.nBodies), 256, d);
			LAUNCH(w, k_color_check, gridFor(d.capContacts), 256, d);
			LAUNCH(w, k_color_masks, gridFor(d.nBodies), 256, d);
			LAUNCH(w, k_block_census, 1, 1024, d, (DState*)nullptr);
			rc = readState(w);
			if (rc) return rc;
			c = w->h_dstate->c;
			c.needRecolor = 1;
			colorSmallQueued = false;
		}
		(void)was;
		// Without a partition every colour is a launch of every sweep, and colours handed out one new contact at a time creep
		// up (24 in use on the settled Tumbler where a colouring from scratch needs 19 - five colours are 0.3 ms of its step):
		// every 64th step the island is coloured afresh.
		// (round 5: ... if they HAVE crept up - more than two colours above what the last colouring from scratch needed; the
		// top colours of a sweep are hops of k_large_rest now, ~2.5 us each, and a colouring from scratch is 5 ms of claim /
		// resolve rounds with read-backs: the settled Tumbler's step 64, 128, ... took 9 - 11 ms against 4)
		if (w->blocksTooBig && forceLarge != 2 && !w->noBlocks && c.nLIslands > 0)
		{
			if (w->recolorCountdown <= 0)
			{
				if (w->freshColors <= 0 || c.nColors > w->freshColors + w->recolorSlack)
				{
					c.needRecolor = 1;
					colorSmallQueued = false;
					w->freshColorsPending = true;
					w->recolorCountdown = 64;
				}
				else w->recolorCountdown = 16;
			}
			w->recolorCountdown -= 1;
		}
		else { w->recolorCountdown = 0; w->freshColors = 0; }
	}
	// (from the next step on: in islands with joints / hubs the constraints of such newcomers are swept in order instead)
	w->serialOrphansNext = (forceLarge != 2 && !w->noBlocks && !w->noSweepBlocks && !w->blocksTooBig && c.nBlocks > 0 && (d.nJoints > 0 || c.maxDegree > HUB_DEGREE)) ? 1 : 0;
	// ---- block partition of the large islands: (re)made when bodies without a home block joined, when a block outgrew a
	// workgroup, or when too many constraints cross block boundaries (the pile has moved since the partition was made)
	// (islands with joints or hub bodies are partitioned too: k_blocks_sweep does their contact sweeps block-wise, one launch
	// per sweep, between the joint walks and the hub sweeps)
	const bool blockShape = forceLarge != 2 && !w->noBlocks && !w->blocksTooBig && c.nLIslands > 0 && (plainIslands || !w->noSweepBlocks) &&
		(sp.warmStarting ? 1 : 0) + sp.velIters > 0 && w->blocksMaxWG > 0;
	if (blockShape && c.partitionCooldown == 0)
	{
		// Block size: one 1024-lane block while the large islands fit it (nothing ever goes through memory then), else
		// 256-lane blocks (measured on the 10 011-box pyramid: 215 us against 235 / 245 us with 512 / 1024 lanes - the
		// workgroup barriers of the interior colours are cheaper and the position solves spread over more CUs)
		// (512 lanes once 256-lane blocks would be more than fit the device together)
		// (... and 1024 again once 512-lane blocks would not)
		auto lanesFor = [w](const Counters& k) { return w->blockLanes ? w->blockLanes : (k.nLContacts <= 900 ? 1024 : (k.nLContacts > 400 * w->blocksMaxWG ? 1024 : (k.nLContacts > 200 * w->blocksMaxWG ? 512 : 256))); };
		auto misfit = [](const Counters& k) { return k.nBlocks == 0 || k.nOrphanRows > 0 || k.blkMaxRows > k.blkLanes || k.blkMaxBodies > k.blkLanes || k.nSerialOrphans > 2048; };
		bool need = misfit(c) || (c.partitionAge > 240 && (4 * c.nCutRows > c.nLContacts || (c.blkLanes != lanesFor(c) && 2 * c.nLContacts < 900)));
		int lanes = lanesFor(c);
		int target = BLOCK_TARGET_DEG * lanes / BLOCK_LANES;
		// (the last partition did not last - a growing pile: leave the blocks room for the bodies they will adopt, if half as
		// many blocks again still fit the device together)
		{
			const int cap = plainIslands ? w->blocksMaxWG : (lanes == 512 ? w->sweepMaxWG[1] : (lanes == BLOCK_LANES ? w->sweepMaxWG[2] : w->sweepMaxWG[0]));
			if (c.nBlocks > 0 && c.partitionAge < 16 && c.nBlocks * 3 / 2 + 8 <= cap) target = target * 2 / 3;
			if (w->tracePartition) fprintf(stderr, "[b2hip] capacity for %d-lane blocks: %d\n", lanes, cap);
		}
		for (int attempt = 0; need && attempt < 3; ++attempt)
		{
			if (w->tracePartition)
				fprintf(stderr, "[b2hip] partition (attempt %d, target %d, lanes %d): blocks %d orphan rows %d max rows %d max bodies %d lanes %d age %d cut %d of %d\n",
					attempt, target, lanes, c.nBlocks, c.nOrphanRows, c.blkMaxRows, c.blkMaxBodies, c.blkLanes, c.partitionAge, c.nCutRows, c.nLContacts);
			HIP_TRY(hipMemcpyAsync(&w->d_state.p->c.blkLanes, &lanes, sizeof(int), hipMemcpyHostToDevice, w->stream));
			rc = partitionLargeIslands(w, target);
			if (rc) return rc;
			colorSmallQueued = false; // (the colours are checked against the new partition: what is open after that is new work)
			rc = readState(w);
			if (rc) return rc;
			c = w->h_dstate->c;
			need = misfit(c);
			target = target * 2 / 3;
		}
		if (need)
		{
			// does not fit (e.g. more blocks than workgroups can be resident): back to the other solvers for a while
			const int cooldown = 120;
			HIP_TRY(hipMemcpyAsync(&w->d_state.p->c.partitionCooldown, &cooldown, sizeof(int), hipMemcpyHostToDevice, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
		}
	}

	stampPhase(w, 4);
	const bool exactLarge = forceLarge == 2;
	const bool hasHubs = !exactLarge && (c.maxDegree > HUB_DEGREE || c.nSerialOrphans > 0); // (anything for k_large_hub)
	// Small and large islands share nothing (different bodies, contacts, island tables): when both tiers are present the
	// small-island chain (DFS order, chunking, k_solve_small) runs on a side stream beside the large-island solver and
	// joins before SynchronizeFixtures. With a handful of small islands that chain is one or two workgroups of big kernels
	// whose cost is instruction fetch from a cold cache (~2 us per KB of code executed: 83 us for a dozen free bodies next
	// to the 10k-body pyramid) - time the large solver's resident grid leaves plenty of idle CUs for.
	bool sideStream = false;
	bool smallDeferred = false;
	// (the launches of the small-island chain; on the side stream they are issued AFTER the large-island solver's own
	// launches - the solver is what the step waits for, and every launch the host makes first delays it by ~3 us)
	auto launchSmallIslands = [&](hipStream_t ss) -> int
	{
		if (sideStream && !poll) HIP_TRY(hipStreamWaitEvent(ss, w->evFork, 0));
		LAUNCH_ON(w, ss, k_island_dfs, gridFor(c.nSIslands, 64, 1 << 20), 64, d);
		if (!sideStream) stampPhase(w, 5);
		if (!exactLarge)
		{
			const bool timeIt = w->kernelTiming == 1 && c.nLIslands == 0;
			if (timeIt) { int rck = ktRecord(w); if (rck) return rck; w->ktKind = 2; }
			if (c.nSmallJointed > 0)
			{
				if (c.chunkLanes == TINY_CHUNK_LANES) LAUNCH_ON(w, ss, (k_solve_small<TINY_CHUNK_LANES, true>), c.nChunks, TINY_CHUNK_LANES, d, sp);
				else LAUNCH_ON(w, ss, (k_solve_small<SMALL_CHUNK_LANES, true>), c.nChunks, SMALL_CHUNK_LANES, d, sp);
			}
			else if (c.chunkLanes == TINY_CHUNK_LANES) LAUNCH_ON(w, ss, (k_solve_small<TINY_CHUNK_LANES, false>), c.nChunks, TINY_CHUNK_LANES, d, sp);
			else LAUNCH_ON(w, ss, (k_solve_small<SMALL_CHUNK_LANES, false>), c.nChunks, SMALL_CHUNK_LANES, d, sp);
			if (timeIt) { int rck = ktRecord(w); if (rck) return rck; }
		}
		if (sideStream) HIP_TRY(hipEventRecord(w->evJoin, ss));
		else stampPhase(w, 6);
		return 0;
	};
	if (c.nSIslands > 0)
	{
		sideStream = !exactLarge && c.nLIslands > 0 && !w->debugTrace && !w->kernelTimingLaunches && !w->noSideStream;
		if (sideStream)
		{
			// fork here (the side stream needs the island build, nothing of the large-island solver); launches later.
			// With the census polled the host has SEEN the island build finish (the publication is its last act): the side
			// stream needs no event to wait for - an event record is a packet of its own on the main stream, ~6 us between
			// the colouring and the solver
			if (!poll) HIP_TRY(hipEventRecord(w->evFork, w->stream));
			smallDeferred = true;
			stampPhase(w, 5);
			stampPhase(w, 6);
		}
		else
		{
			rc = launchSmallIslands(w->stream);
			if (rc) return rc;
		}
	}
	else
	{
		stampPhase(w, 5);
		stampPhase(w, 6);
	}
	int nColors = 0;
	int nLIslands = c.nLIslands, nLBodies = c.nLBodies, nLContacts = c.nLContacts;
	if (exactLarge && c.nSIslands > 0)
	{
		// exact-order mode: colours := dependency levels of the reference's own constraint order
		LAUNCH(w, k_exact_begin, gridFor(c.nSContacts + 1), 256, d);
		LAUNCH(w, k_exact_convert, gridFor(std::max(c.nSContacts, c.nSBodies)), 256, d);
		rc = readState(w);
		if (rc) return rc;
		nColors = w->h_dstate->c.nColors;
		nLIslands = c.nSIslands;
		nLBodies = c.nSBodies;
		nLContacts = c.nSContacts;
	}
	if (nLIslands > 0)
	{
		const int gB = gridFor(nLBodies), gC = gridFor(std::max(nLContacts, 1));
		nColors = exactLarge ? nColors : c.nColors;
		const bool hasJoints = d.nJoints > 0;
		// k_solve_dataflow: two waves per workgroup while that still leaves at most ~2 workgroups per CU (a hand-off is
		// priced by the consumer CU's memory queue: 256 -> 128 lanes took the 10k-body pyramid from 464 to 409 us; 64 lanes
		// lost it again to the grid barriers), four waves for bigger islands
		const int dfLanes = w->dfLanesForced ? w->dfLanesForced : (nLContacts <= 128 * 2 * w->nCU ? 128 : PERSIST_LANES);
		const int persistLanes = w->solverBarriers ? PERSIST_LANES : dfLanes;
		const int persistWG = (nLContacts + persistLanes - 1) / persistLanes;
		const int persistMaxWG = w->persistMaxWG * (PERSIST_LANES / persistLanes);
		const bool usePersistent = !exactLarge && !hasJoints && !hasHubs && !w->debugTrace && !w->kernelTimingLaunches &&
			persistMaxWG > 0 && persistWG <= persistMaxWG;
		// the block solver (bodies in LDS, one workgroup per block of the partition): whenever the partition fits
		const bool partitionFits = blockShape && c.nBlocks > 0 && c.nOrphanRows == 0 && c.blkMaxRows <= c.blkLanes && c.blkMaxBodies <= c.blkLanes;
		const bool useBlocks = usePersistent && partitionFits && plainIslands && !w->solverBarriers && !w->solverRows && !w->solverMailbox &&
			c.nBlocks <= w->blocksMaxWG &&
			(sp.velIters + 2) * (MAX_COLORS + 1) < 65536 && (sp.posIters + 1) * (MAX_COLORS + 1) < 65536;
		// one launch per sweep over the same blocks for islands that need joint walks / hub sweeps in between
		const int sweepMaxWG = c.blkLanes == 512 ? w->sweepMaxWG[1] : (c.blkLanes == BLOCK_LANES ? w->sweepMaxWG[2] : w->sweepMaxWG[0]);
		const bool useSweep = !useBlocks && !exactLarge && partitionFits && !plainIslands && !w->debugTrace && !w->kernelTimingLaunches &&
			c.nBlocks <= sweepMaxWG;
		d.blockSort = (useBlocks || useSweep) ? 1 : 0;
		const bool useResident = usePersistent && (useBlocks || B2HIP_HAVE_VALIDATION_SOLVERS);
		bool colorsOnDevice = false;
		bool censusVoid = false;
		if (!exactLarge && (c.needRecolor || c.nUncolored > 0 || c.nCompact > 0))
		{
			if (!c.needRecolor && c.nUncolored <= COLOR_SMALL_MAX)
			{
				// the usual case (a few new contacts on a settled island, a colour class to compact): one workgroup colours
				// them; the resident solver reads the colour count from the device, the launch-per-colour path reads it back
				if (!colorSmallQueued) LAUNCH(w, k_color_small, 1, 1024, d, 0); // (else: it went out behind the census)
				if (useResident)
				{
					colorsOnDevice = true;
					w->colorSmallPending = true;
				}
				else
				{
					rc = readState(w);
					if (rc) return rc;
					nColors = w->h_dstate->c.nColors;
					if (w->h_dstate->c.overflow & 4) return setError(B2HIP_ERR_CAPACITY, "more than 64 constraint colours on one body");
					if (w->h_dstate->c.nUncolored != 0) return setError(B2HIP_ERR_CAPACITY, "incremental colouring did not converge");
				}
			}
			else
			{
			// a colour clash on some body -> colour the large islands from scratch; otherwise only the
			// constraints that have no colour yet join the Jones-Plassmann rounds (existing masks stay)
			censusVoid = true; // (thousands of constraints get their colours now: the colour census of this step is history)
			int uncolored = c.nUncolored;
			if (c.needRecolor)
			{
				LAUNCH(w, k_color_begin, gridFor(d.capContacts), 256, d);
				uncolored = nLContacts; // upper bound (hub constraints excluded on the device); refreshed by the read-back below
			}
			int batch = c.needRecolor ? 8 : 2;
			while (uncolored > 0)
			{
				for (int r = 0; r < batch; ++r)
				{
					LAUNCH(w, k_color_claim, gC, 256, d);
					LAUNCH(w, k_color_resolve, gC, 256, d);
				}
				rc = readState(w);
				if (rc) return rc;
				uncolored = w->h_dstate->c.nUncolored;
				nColors = w->h_dstate->c.nColors;
				if (w->h_dstate->c.overflow & 4) return setError(B2HIP_ERR_CAPACITY, "more than 64 constraint colours on one body");
				batch = 8;
			}
			if (w->freshColorsPending) { w->freshColors = nColors < 63 ? nColors : 63; w->freshColorsPending = false; }
			}
		}
		if (!d.blockSort) LAUNCH(w, k_color_scan, 1, 1, d); // (segments by colour: the block solvers sort their rows themselves)
		// ---- the end of every sweep without a launch per colour (b2d_kernels_sweep_end.h). The REST colours - from the highest
		// colour down while this step's colour census (k_color_check, published with the island census) keeps them below
		// restRowsMax rows together - are swept by ONE launch of k_large_rest, as data flow per body; k_color_fill notes them
		// on their bodies. Which colours are "rest" changes nothing in the result (the order on every body is the launches'):
		// a launch-count matter. Not on a step that colours afresh: its census is void.
		const bool useSweepEnd = w->sweepEnd && !exactLarge && !w->debugTrace;
		int restFirst = 0x7fffffff; // (none)
		if (useSweepEnd && w->restFlow && !d.blockSort && !c.needRecolor && !censusVoid && nColors > 0 && nColors < MAX_COLORS)
		{
			long long sum = 0;
			int t = nColors;
			while (t > 0 && sum + c.colorRows[t - 1] <= (long long)w->restRowsMax)
			{
				sum += c.colorRows[t - 1];
				t -= 1;
			}
			if (nColors - t >= 2) restFirst = t;
		}
		w->lastRestFirst = restFirst < nColors ? restFirst : nColors;
		LAUNCH(w, k_color_fill, gC, 256, d, restFirst < MAX_COLORS ? restFirst : MAX_COLORS);
		if (hasHubs)
		{
			// the hub constraints in contact-index order (deterministic whatever the atomics of k_color_fill did)
			LAUNCH(w, k_hub_flag, gridFor(d.capContacts), 256, d);
			deviceExclusiveScan<int>(w->stream, d.keepFlag, d.keepScan, d.scanTmp, w->scanCtx, &d.st->c.nContacts, d.capContacts);
			LAUNCH(w, k_hub_fill, gridFor(d.capContacts), 256, d);
			w->hubSteps += 1;
		}
		stampPhase(w, 7);
		const int gK = gridFor(std::max(nLContacts / std::max(nColors, 1), 1) * 2);
		const int gJ = gridFor(std::max(nLIslands, 1), 64, 1 << 16);
		if (useResident)
		{
			// one resident grid for the whole sweep structure; colour boundaries are grid barriers (validation_src/b2d_validation_solvers.h)
			// (the barrier words were zeroed by k_step_begin: one resident launch per step)
			if (w->kernelTiming == 1) { rc = ktRecord(w); if (rc) return rc; w->ktKind = useBlocks ? 4 : 3; }
			const int nColorsArg = colorsOnDevice ? -1 : nColors; // -1: read Counters::nColors on the device
			if (useBlocks)
			{
				// (tags carry a 15-bit epoch: wipe the exchange rows when it comes round, like the mailbox slots below)
				if ((w->dfEpoch >> 14) != w->dfWipedAt)
				{
					HIP_TRY(hipMemsetAsync(w->b_cutv.p, 0, w->b_cutv.cap * sizeof(float4), w->stream));
					HIP_TRY(hipMemsetAsync(w->b_posv.p, 0, w->b_posv.cap * sizeof(float4), w->stream));
					HIP_TRY(hipMemsetAsync(w->dfInbox.p, 0, w->dfInbox.cap * sizeof(float4), w->stream));
					w->dfWipedAt = w->dfEpoch >> 14;
				}
				if (c.blkLanes == 512) LAUNCH(w, k_solve_blocks<512>, c.nBlocks, 512, d, sp, w->gridBar.p, w->dfEpoch);
				else if (c.blkLanes == 256) LAUNCH(w, k_solve_blocks<256>, c.nBlocks, 256, d, sp, w->gridBar.p, w->dfEpoch);
				else if (c.blkLanes == BLOCK_LANES) LAUNCH(w, k_solve_blocks<BLOCK_LANES>, c.nBlocks, BLOCK_LANES, d, sp, w->gridBar.p, w->dfEpoch);
				else return setError(B2HIP_ERR_INVALID, "block partition made for an unknown workgroup size");
				w->dfEpoch += 1;
				w->blockSteps += 1;
				w->blocksThisStep = true;
			}
#if B2HIP_HAVE_VALIDATION_SOLVERS
			else if (w->solverBarriers) LAUNCH(w, k_solve_persistent, persistWG, PERSIST_LANES, d, sp, nColorsArg, w->gridBar.p);
			else if (w->solverRows || (sp.velIters + 2) * DF_RANKS >= 65536 || (sp.posIters + 1) * DF_RANKS >= 65536)
				LAUNCH(w, k_solve_dataflow, persistWG, persistLanes, d, sp, nColorsArg, w->gridBar.p, w->dfSleep);
			else
			{
				// optional single-XCD attempt first when the island fits one XCD's CUs (k_solve_mailbox<true>), then the ordinary
				// launch, which returns at once if the attempt took the step
				const int xcdWG = persistMaxWG / 8;
				const bool tryLocal = w->solverLocal && persistWG <= xcdWG;
				// the 15-bit epoch of the mailbox tags comes round every 16 384 steps: wipe the slots then, so that a slot
				// nobody has written since cannot carry a matching tag
				if (w->dfEpoch != 0 && (w->dfEpoch & 0x3fff) == 0)
				{
					HIP_TRY(hipMemsetAsync(w->dfInbox.p, 0, w->dfInbox.cap * sizeof(float4), w->stream));
					HIP_TRY(hipMemsetAsync(w->b_cutv.p, 0, w->b_cutv.cap * sizeof(float4), w->stream));
				}
				if (tryLocal) LAUNCH(w, k_solve_mailbox<true>, 8 * persistWG, persistLanes, d, sp, nColorsArg, w->gridBar.p, w->dfEpoch, persistWG, 0);
				LAUNCH(w, k_solve_mailbox<false>, persistWG, persistLanes, d, sp, nColorsArg, w->gridBar.p, w->dfEpoch, persistWG, tryLocal ? 1 : 0);
				w->dfEpoch += 1;
			}
#endif
			if (w->kernelTiming == 1) { rc = ktRecord(w); if (rc) return rc; }
			w->persistSteps += 1;
			if (smallDeferred) { smallDeferred = false; rc = launchSmallIslands(w->stream2); if (rc) return rc; }
		}
		else
		{
		if (smallDeferred) { smallDeferred = false; rc = launchSmallIslands(w->stream2); if (rc) return rc; }
		TRACE("before_integrate");
		if (w->debugTrace)
		{
			const size_t nb = w->bodies.size();
			(void)w->dbgPreVel.ensure(nb, w->stream); (void)w->dbgVel.ensure(nb, w->stream); (void)w->dbgLi.ensure(nb + 64, w->stream);
			HIP_TRY(hipMemcpyAsync(w->dbgPreVel.p, w->b_vel.p, nb * 16, hipMemcpyDeviceToDevice, w->stream));
			HIP_TRY(hipMemcpyAsync(w->dbgLi.p, w->li_bodies.p, nb * 4, hipMemcpyDeviceToDevice, w->stream));
			HIP_TRY(hipMemcpyAsync(w->dbgLi.p + nb, &w->d_state.p->c, 64 * 4 > sizeof(Counters) ? sizeof(Counters) : 64 * 4, hipMemcpyDeviceToDevice, w->stream));
		}
		// (timing mode 5: ONE event pair around the whole large-island solver family of the launch-per-colour path - integrate,
		// constraint set-up, every sweep, impulses stored, positions, write-back and sleep)
		if (w->kernelTiming == 5) { rc = ktRecord(w); if (rc) return rc; w->ktKind = 8; w->familyLaunchesAtStart = w->launchCount; }
		LAUNCH(w, k_large_integrate, gB, 256, d, sp);
		if (w->debugTrace) HIP_TRY(hipMemcpyAsync(w->dbgVel.p, w->b_vel.p, w->bodies.size() * 16, hipMemcpyDeviceToDevice, w->stream));
		TRACE("integrate");
		if (hasJoints && !exactLarge) LAUNCH(w, k_joints_sort, gJ, 64, d);
		LAUNCH(w, k_large_init, gC, 256, d, sp);
		TRACE("init");
		// colours that own no constraint (the partition keeps two colour ranges apart) are not launched
		const uint64_t colorMask = exactLarge ? ~0ull : ((uint64_t)w->h_dstate->c.colorMaskLo | ((uint64_t)w->h_dstate->c.colorMaskHi << 32));
		auto colorUsed = [&](int col) { return col >= 64 || ((colorMask >> col) & 1ull) != 0; };
		// every launch of k_blocks_sweep tags its hand-over rows with an epoch of its own (15 bits: the rows are wiped twice per round)
		auto sweep = [&](int mode) -> int
		{
			if ((w->dfEpoch >> 14) != w->dfWipedAt)
			{
				HIP_TRY(hipMemsetAsync(w->b_cutv.p, 0, w->b_cutv.cap * sizeof(float4), w->stream));
				HIP_TRY(hipMemsetAsync(w->b_posv.p, 0, w->b_posv.cap * sizeof(float4), w->stream));
				w->dfWipedAt = w->dfEpoch >> 14;
			}
			if (c.blkLanes == 512) LAUNCH(w, k_blocks_sweep<512>, c.nBlocks, 512, d, sp, mode, w->gridBar.p, w->dfEpoch);
			else if (c.blkLanes == 256) LAUNCH(w, k_blocks_sweep<256>, c.nBlocks, 256, d, sp, mode, w->gridBar.p, w->dfEpoch);
			else if (c.blkLanes == BLOCK_LANES) LAUNCH(w, k_blocks_sweep<BLOCK_LANES>, c.nBlocks, BLOCK_LANES, d, sp, mode, w->gridBar.p, w->dfEpoch);
			else return setError(B2HIP_ERR_INVALID, "block partition made for an unknown workgroup size");
			w->dfEpoch += 1;
			return 0;
		};
		// the hub sweeps: eight waves that fetch their chunks of hub constraints ahead of their turn (one wave on request)
		auto hubSweepLaunch = [&](int mode, int useGuess, int behindWide = 0) -> int
		{
			if (w->hubWaves == 1) LAUNCH(w, k_large_hub<1>, 1, 64, d, mode, useGuess, behindWide);
			else LAUNCH(w, k_large_hub<8>, 1, 512, d, mode, useGuess, behindWide);
			return 0;
		};
		if (useSweep) w->sweepSteps += 1;
		// ---- the end of every sweep in ONE single-workgroup launch (b2d_kernels_sweep_end.h): the tail colours - those this
		// step's colour census (k_color_check, published with the island census) found small, from the highest colour down -
		// the hub rows, the joint walk, the verdict of a position iteration. Which colours are "tail" changes nothing in the
		// result (k_sweep_end does k_large_velocity's / k_large_position's arithmetic row for row): a pure launch-count matter.
		const bool useRest = useSweepEnd && restFirst < nColors;
		int tailFirst = useRest ? restFirst : nColors;
		if (useSweepEnd && w->sweepTail && !useRest && !useSweep && !c.needRecolor && !censusVoid && nColors <= MAX_COLORS)
		{
			long long sum = 0;
			while (tailFirst > 0 && c.colorRows[tailFirst - 1] <= w->tailRowsMax && sum + c.colorRows[tailFirst - 1] <= 8ll * w->tailRowsMax)
			{
				sum += c.colorRows[tailFirst - 1];
				tailFirst -= 1;
			}
		}
		bool tailAny = false;
		for (int col = tailFirst; col < nColors && !useRest; ++col) tailAny = tailAny || colorUsed(col);
		// (every launch of k_large_rest tags its hand-over rows with an epoch of its own, like k_blocks_sweep)
		auto restLaunch = [&](int mode) -> int
		{
			if (!useRest) return 0;
			if ((w->dfEpoch >> 14) != w->dfWipedAt)
			{
				HIP_TRY(hipMemsetAsync(w->b_cutv.p, 0, w->b_cutv.cap * sizeof(float4), w->stream));
				HIP_TRY(hipMemsetAsync(w->b_posv.p, 0, w->b_posv.cap * sizeof(float4), w->stream));
				w->dfWipedAt = w->dfEpoch >> 14;
			}
			long long rows = 0;
			for (int col = restFirst; col < nColors && col < MAX_COLORS; ++col) rows += c.colorRows[col];
			// (the census is this step's before k_color_small handed out its colours - at most COLOR_SMALL_MAX rows more)
			const int gR = (int)((rows + COLOR_SMALL_MAX + 255) / 256);
			if (mode == 0) LAUNCH(w, k_large_rest<0>, gR, 256, d, restFirst, nColors, w->gridBar.p, w->dfEpoch);
			else if (mode == 1) LAUNCH(w, k_large_rest<1>, gR, 256, d, restFirst, nColors, w->gridBar.p, w->dfEpoch);
			else LAUNCH(w, k_large_rest<2>, gR, 256, d, restFirst, nColors, w->gridBar.p, w->dfEpoch);
			w->dfEpoch += 1;
			return 0;
		};
		w->lastTailFirst = tailFirst;
		w->lastSweepLaunches = 0;
		const int bigEnd = useSweep ? 0 : tailFirst; // colours [0, bigEnd) are launches of their own
		// The hub rows the one fixed point cannot take are swept lane after lane inside k_sweep_end - a handful on the Tumbler
		// (boxes in the corners). An island that kept many of them in the LAST step (several hubs, constraints swept in order for
		// lack of a home block; the census still carries that step's counts) gets k_large_hub's eight prefetching waves for
		// them: k_sweep_end up to the fixed point, k_large_hub, k_sweep_end for what follows the hub rows. (Either way a valid
		// sweep; which one is decided from counters a snapshot carries, so a loaded world decides alike.)
		const bool leftoverApart = useSweepEnd && hasHubs && d.hubWide && c.nHubRows - c.nHubWide > SE_LEFT_INLINE_MAX;
		auto sweepEndOne = [&](int mode, int tf, int te, int what) -> int
		{
			if (mode == 0) LAUNCH(w, k_sweep_end<0>, 1, SWEEP_END_LANES, d, sp, tf, te, what);
			else if (mode == 1) LAUNCH(w, k_sweep_end<1>, 1, SWEEP_END_LANES, d, sp, tf, te, what);
			else LAUNCH(w, k_sweep_end<2>, 1, SWEEP_END_LANES, d, sp, tf, te, what);
			w->lastSweepLaunches += 1;
			return 0;
		};
		auto sweepEndLaunch = [&](int mode, int what) -> int
		{
			if (!what && !tailAny) return 0;
			const int tf = (useSweep || useRest) ? 0 : tailFirst, te = (useSweep || useRest) ? 0 : nColors;
			if (leftoverApart && (what & SE_HUB))
			{
				int rcl = sweepEndOne(mode, tf, te, (what & (SE_HUB | SE_GUESS)) | SE_HUB_WIDE_ONLY);
				if (rcl) return rcl;
				rcl = hubSweepLaunch(mode, (what & SE_GUESS) ? 1 : 0, 1);
				if (rcl) return rcl;
				const int rest = what & ~(SE_HUB | SE_GUESS);
				return rest ? sweepEndOne(mode, 0, 0, rest) : 0;
			}
			return sweepEndOne(mode, tf, te, what);
		};
		if (sp.warmStarting)
		{
			if (useSweep) { rc = sweep(0); if (rc) return rc; }
			else
			{
				for (int col = 0; col < bigEnd; ++col)
					if (colorUsed(col)) LAUNCH(w, k_large_velocity, gK, 256, d, col, 0);
				rc = restLaunch(0);
				if (rc) return rc;
			}
			if (useSweepEnd)
			{
				// (b2Island.cpp:256-268: the joints' InitVelocityConstraints follows the contacts' warm start; the first velocity
				// iteration then begins with the joints)
				rc = sweepEndLaunch(0, (hasHubs ? SE_HUB : 0) | (hasJoints ? SE_JOINTS_INIT | (sp.velIters > 0 ? SE_JOINTS_VEL : 0) : 0));
				if (rc) return rc;
			}
			else if (hasHubs) { rc = hubSweepLaunch(0, 0); if (rc) return rc; }
		}
		TRACE("warmstart");
		if (hasJoints && !(useSweepEnd && sp.warmStarting)) LAUNCH(w, k_large_joints, gJ, 64, d, sp, 0);
		for (int it = 0; it < sp.velIters; ++it)
		{
			// (with k_sweep_end the joint walk of iteration it is the last act of the sweep before it)
			if (hasJoints && !(useSweepEnd && (it > 0 || sp.warmStarting))) LAUNCH(w, k_large_joints, gJ, 64, d, sp, 1);
			if (useSweep) { rc = sweep(1); if (rc) return rc; }
			else
			for (int col = 0; col < bigEnd; ++col)
			{
				if (!colorUsed(col)) continue;
				if (w->kernelTiming == 1) { rc = ktRecord(w); if (rc) return rc; w->ktKind = 1; }
				LAUNCH(w, k_large_velocity, gK, 256, d, col, 1);
				if (w->kernelTiming == 1) { rc = ktRecord(w); if (rc) return rc; }
				if (w->debugTrace) TRACE(("vel" + std::to_string(it) + "_c" + std::to_string(col)).c_str());
			}
			if (!useSweep) { rc = restLaunch(1); if (rc) return rc; }
			if (useSweepEnd)
			{
				rc = sweepEndLaunch(1, (hasHubs ? SE_HUB | (it > 0 ? SE_GUESS : 0) : 0) | (hasJoints && it + 1 < sp.velIters ? SE_JOINTS_VEL : 0));
				if (rc) return rc;
			}
			else if (hasHubs) { rc = hubSweepLaunch(1, it > 0 ? 1 : 0); if (rc) return rc; }
		}
		LAUNCH(w, k_large_store_impulses, gC, 256, d);
		TRACE("store_impulses");
		LAUNCH(w, k_large_integrate_positions, gB, 256, d, sp);
		TRACE("integrate_positions");
		for (int it = 0; it < sp.posIters; ++it)
		{
			if (!useSweepEnd || it == 0) LAUNCH(w, k_large_pos_begin, gridFor(nLIslands), 256, d);
			if (useSweep) { rc = sweep(2); if (rc) return rc; }
			else
			for (int col = 0; col < bigEnd; ++col)
			{
				if (!colorUsed(col)) continue;
				LAUNCH(w, k_large_position, gK, 256, d, col);
				if (w->debugTrace) TRACE(("pos" + std::to_string(it) + "_c" + std::to_string(col)).c_str());
			}
			if (!useSweep) { rc = restLaunch(2); if (rc) return rc; }
			if (useSweepEnd)
			{
				rc = sweepEndLaunch(2, (hasHubs ? SE_HUB | (it > 0 ? SE_GUESS : 0) : 0) | (hasJoints ? SE_JOINTS_POS : 0) | SE_POS_END | (it + 1 < sp.posIters ? SE_POS_BEGIN : 0));
				if (rc) return rc;
			}
			else
			{
				if (hasHubs) { rc = hubSweepLaunch(2, it > 0 ? 1 : 0); if (rc) return rc; }
				if (hasJoints) LAUNCH(w, k_large_joints, gJ, 64, d, sp, 2);
				LAUNCH(w, k_large_pos_end, 1, 256, d);
			}
		}
		}
		LAUNCH(w, k_large_finalize, gB, 256, d, sp);
		TRACE("finalize");
		LAUNCH(w, k_large_sleep, gB, 256, d, sp);
		if (w->kernelTiming == 5 && w->ktKind == 8) { w->familyLaunches = (int)(w->launchCount - w->familyLaunchesAtStart); rc = ktRecord(w); if (rc) return rc; }
		TRACE("sleep");
		if (sideStream) HIP_TRY(hipStreamWaitEvent(w->stream, w->evJoin, 0));
		stampPhase(w, 8);
	}
	else
	{
		stampPhase(w, 7);
		stampPhase(w, 8);
	}
	w->last.nSIslands = c.nSIslands;
	w->last.nFreeIslands = c.nFreeIslands;
	w->last.nSBodies = c.nSBodies;
	w->last.nSContacts = c.nSContacts;
	w->last.nChunks = c.nChunks;
	w->last.nLIslands = nLIslands;
	w->last.nLBodies = nLBodies;
	w->last.nLContacts = nLContacts;
	w->last.nIslands = c.nIslands;
	w->last.nColors = nColors;
	w->last.nTouching = c.nTouching;
	w->last.nDestroy = c.nDestroy;
	w->last.nBlocks = c.nBlocks;
	w->last.nCutRows = c.nCutRows;
	w->last.blkMaxRows = c.blkMaxRows;
	w->last.partitions = c.partitions;
	return 0;
}

static int phaseSyncFixtures(b2hip_world* w)
{
	DW& d = w->dw;
	if (int rk = ktBracket(w, 3, 6)) return rk;
	LAUNCH(w, k_sync_fixtures, gridFor(d.nProxies >= 262144 ? ((size_t)d.nProxies + 3) / 4 : (size_t)d.nProxies), 256, d); // (SYNC_TILE proxies per workgroup and round in a large world)
	if (int rk = ktBracket(w, 3, 6)) return rk;
	return 0;
}

// The serial event loop walks contacts by body (CSR) and searches new pairs through the hash grid.
static int toiBuildIndexes(b2hip_world* w, bool csr)
{
	DW& d = w->dw;
	if (csr)
	{
		LAUNCH(w, k_toi_adj_clear, gridFor(d.nBodies + 1), 256, d);
		LAUNCH(w, k_toi_adj_count, gridFor(d.capContacts), 256, d);
		deviceExclusiveScan<int>(w->stream, d.deg, d.adjStart, d.scanTmp, w->scanCtx, w->consts.p + 4, d.nBodies + 1);
		LAUNCH(w, k_toi_adj_fill, gridFor(d.capContacts), 256, d);
	}
	// make the grid reflect every fat AABB as of now (the end-of-step pair update skips the rebuild when nothing
	// moved, and TOI moves of earlier steps never enter the move buffer)
	LAUNCH(w, k_grid_clear, gridFor(d.gridMask + 1), 256, d, 1);
	LAUNCH(w, k_grid_count, gridFor(d.nProxies), 256, d, 1);
	deviceExclusiveScan<int>(w->stream, d.gridCount, d.gridStart, d.scanTmp, w->scanCtx, w->consts.p + 1, (int)(d.gridMask + 1));
	LAUNCH(w, k_grid_fill, gridFor(d.nProxies), 256, d, 1);
	return 0;
}

static int toiSerial(b2hip_world* w)
{
	DW& d = w->dw;
	int rc = toiBuildIndexes(w, true);
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(&w->d_state.p->c.toiUnsafe, 0, sizeof(int) * 3, w->stream));
	LAUNCH(w, k_toi_loop, 1, TOI_LANES, d, w->sp);
	if (!w->spatial) LAUNCH(w, k_toi_clear, gridFor(d.nBodies), 256, d); // (spatial worlds: k_end_step does it, behind the exchange)
	w->toiChains = false;
	return 0;
}

// b2World::SolveTOI (b2World.cpp:1026-1093). The first arg-min pass runs over the whole contact array; the
// event loop only runs (one persistent workgroup) when some impact lies inside the step.
static int phaseToiSync(b2hip_world* w);

// The phase without a host round trip: k_toi_first, then the chain kernels at once. Each of them leaves immediately if no
// impact is pending (or if k_toi_first saw a bullet / kinematic partner: toiUnsafe), so the host learns the outcome from
// the read-back b2hip_step_end makes anyway, and falls back there (snapshot restore + serial loop, or - when the pair
// update had overflowed its optimistic small-sort path, so this phase did not see every contact - restore, finish the
// contacts, and the synchronous phase). One read-back and ~45 us less per step with continuous physics on.
static int phaseToi(b2hip_world* w)
{
	if (w->toiSerialOnly || w->toiSyncOnly || listenerOn(w) || w->dw.toiEventCap > 0 || w->dw.toiContinue || w->spatial) return phaseToiSync(w);
	if (w->toiSyncSticky > 0)
	{
		// the serial loop was needed recently (bullets, kinematic partners, contact-creating events): decide from the
		// read-back again instead of paying a wasted snapshot + state download per step
		int rc = phaseToiSync(w);
		if (rc) return rc;
		if (w->h_dstate->c.nToiList == 0 || (w->toiChains && w->h_dstate->c.toiUnsafe == 0)) w->toiSyncSticky -= 1;
		else w->toiSyncSticky = 16;
		return 0;
	}
	DW& d = w->dw;
	if (!w->toiCountersFresh)
	{
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.nToiList, 0, sizeof(int) * 5, w->stream));
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.toiUnsafe, 0, sizeof(int) * 3, w->stream));
	}
	w->toiCountersFresh = false;
	LAUNCH(w, k_toi_first, gridFor(d.capContacts), 256, d);
	const int haveGrid = w->toiGridSticky > 0 ? 1 : 0;
	LAUNCH(w, k_toi_groups_begin, gridFor(std::min(d.capContacts, 1 << 16)), 256, d);
	LAUNCH(w, k_toi_group_contacts, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 1); // (+ the snapshot)
	w->toiSnapshotTaken = true;
	if (haveGrid)
	{
		int rc = toiBuildIndexes(w, false);
		if (rc) return rc;
	}
	LAUNCH(w, k_toi_chains, 1024, CHAIN_LANES, d, w->sp, haveGrid);
	// (k_toi_clear's work is done by k_end_step, which follows)
	w->toiChainsHadGrid = haveGrid != 0;
	w->toiChains = true;
	w->toiSpeculative = true;
	return 0;
}

static int phaseToiSync(b2hip_world* w)
{
	DW& d = w->dw;
	// one read-back serves both questions: did the optimistic small-sort path of the end-of-step pair update
	// apply (else finish it first: the TOI phase must see every contact), and is any impact pending
	int rc = 0;
	for (int pass = 0; pass < 2; ++pass)
	{
		if (pass == 1 || !w->toiCountersFresh)
		{
			// (the first pass of a step starts from the zeros of k_step_begin)
			HIP_TRY(hipMemsetAsync(&w->d_state.p->c.nToiList, 0, sizeof(int) * 5, w->stream));
			HIP_TRY(hipMemsetAsync(&w->d_state.p->c.toiUnsafe, 0, sizeof(int) * 3, w->stream));
		}
		w->toiCountersFresh = false;
		LAUNCH(w, k_toi_first, gridFor(d.capContacts), 256, d);
		rc = readState(w);
		if (rc) return rc;
		if (w->h_dstate->c.overflow & 3)
		{
			// the end-of-step pair update overflowed its buffer (or the contact array): grow, run the whole update again, look again
			if (pass == 1) return setError(B2HIP_ERR_CAPACITY, "pair buffer overflow");
			rc = growPairBuffers(w);
			if (rc) return rc;
			rc = findNewContacts(w, true);
			if (rc) return rc;
			continue;
		}
		if (pass == 1 || w->h_dstate->c.nMoves == 0 || w->h_dstate->c.nPairs <= COUNT_RANK_MAX) break;
		rc = runSortAndCreate(w, true, w->h_dstate->c.nPairs);
		if (rc) return rc;
	}
	w->last.nToiList = w->h_dstate->c.nToiList;
	w->last.nToiCalls = w->h_dstate->c.nToiCalls;
	w->last.nToiEvents = 0;
	w->spContactsBeforeToi = w->h_dstate->c.nContacts;
	w->spToiOrderBefore = w->h_dstate->c.nToiOrder;
	// (sub-stepping: one event per call in the reference's serial order; a call that continues a step has impacts to compute
	// even when nothing is pending yet - the event loop's first batch)
	const bool subStepped = d.toiEventCap > 0 || d.toiContinue != 0;
	if (w->h_dstate->c.nToiList == 0 && !d.toiContinue) return 0;
	if (w->h_dstate->c.nToiList > d.capContacts) return setError(B2HIP_ERR_CAPACITY, "TOI list overflow");
	w->toiRan = true;
	if (w->h_dstate->c.toiUnsafe == 0 && !w->toiSerialOnly && !listenerOn(w) && !subStepped)
	{
		// every pending impact pairs a dynamic body with a static one: one wave per dynamic body, verified afterwards
		// (b2hip_step_end falls back to the serial loop from the snapshot if a chain met a case that is order dependent)
		// The hash grid is only needed when a chain moves a proxy out of its fat AABB: it is rebuilt while that has
		// happened recently, otherwise such a move sends the phase to the serial loop (which rebuilds it).
		const int groups = std::min(std::min(w->h_dstate->c.nToiList, d.nBodies), (int)TOI_GROUPS_MAX);
		const int haveGrid = w->toiGridSticky > 0 ? 1 : 0;
		LAUNCH(w, k_toi_groups_begin, gridFor(w->h_dstate->c.nToiList), 256, d);
		LAUNCH(w, k_toi_group_contacts, gridFor(d.capContacts), 256, d, 0);
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 0);
		w->toiSnapshotTaken = true;
		if (haveGrid)
		{
			rc = toiBuildIndexes(w, false);
			if (rc) return rc;
		}
		LAUNCH(w, k_toi_chains, std::min(groups, 1024), CHAIN_LANES, d, w->sp, haveGrid);
		if (!w->spatial) LAUNCH(w, k_toi_clear, gridFor(d.nBodies), 256, d);
		w->toiChainsHadGrid = haveGrid != 0;
		w->toiChains = true;
		return 0;
	}
	if (!w->toiSerialOnly && !w->toiNoDomains && !listenerOn(w) && !subStepped)
	{
		// bullets / kinematic partners: the event loop runs per connected component of the contact graph, side by side
		// (b2d_kernels_toi_domains.h); b2hip_step_end falls back to the serial loop from the snapshot if a component met
		// something that couples it to another one
		rc = toiBuildIndexes(w, true);
		if (rc) return rc;
		LAUNCH(w, k_toi_dom_init, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_toi_dom_union, gridFor(d.capContacts), 256, d);
		LAUNCH(w, k_toi_dom_flatten, gridFor(d.nBodies), 256, d);
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.toiUnsafe, 0, sizeof(int), w->stream)); // (k_toi_first's "not chains" bit)
		LAUNCH(w, k_toi_dom_mark, gridFor(w->h_dstate->c.nToiList), 256, d);
		LAUNCH(w, k_toi_dom_count, gridFor(d.capContacts), 256, d);
		LAUNCH(w, k_toi_dom_scan, 1, 1024, d);
		LAUNCH(w, k_toi_dom_fill, gridFor(w->h_dstate->c.nToiList), 256, d);
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 0);
		w->toiSnapshotTaken = true;
		LAUNCH(w, k_toi_domains, std::min(w->h_dstate->c.nToiList, 2048), TOI_LANES, d, w->sp);
		LAUNCH(w, k_toi_domains_end, std::min(std::max(w->h_dstate->c.nToiList, 1), 1024), 256, d);
		// components tied together by a new contact: back to the snapshot, then the serial loop over just those
		LAUNCH(w, k_toi_dom_rollback, gridFor(std::max(std::max(d.nBodies, d.nProxies), d.capContacts)), 256, d);
		LAUNCH(w, k_toi_loop_partial, 1, TOI_LANES, d, w->sp);
		if (!w->spatial) LAUNCH(w, k_toi_clear, gridFor(d.nBodies), 256, d);
		w->toiChainsHadGrid = true; // (the components always have it)
		w->toiChains = true;
		return 0;
	}
	if ((hasPreSolve(w) || w->spatial) && !w->toiSnapshotTaken)
	{
		// a PreSolve called from a sub-step may change that sub-step (toiPreSolveRounds): the phase must be able to start over
		// (a spatially sharded world sends the other ranks what differs from this snapshot: spAfterToi)
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 0);
		w->toiSnapshotTaken = true;
	}
	return toiSerial(w);
}

// The read-back of a step (and of a between-step destroy): k_end_step writes the state rows and, behind them, the counters
// straight into the pinned host buffer, its last workgroup the sequence number - which the host polls. No copy, no stream
// synchronisation (the kernel is the last thing on the stream).
static int awaitState(b2hip_world* w, size_t nb)
{
	const DState* tail = (const DState*)(w->h_state + B2D_STATE_TAIL(nb));
	if (int rc = pollPublished(w, (volatile const int*)&tail->pubSeq, w->stateSeq, "state read-back")) return rc;
	memcpy(w->h_dstate, (const void*)tail, offsetof(DState, pubSeq));
	return 0;
}

static inline bool shadowValid(const b2hip_world* w)
{
	return w->shadowDev != nullptr && w->shadowDev == w->stateOut.p && w->shadowHost == w->h_state && w->shadowRows == (size_t)w->dw.nBodies;
}
static inline void shadowWritten(b2hip_world* w)
{
	w->shadowDev = w->stateOut.p;
	w->shadowHost = w->h_state;
	w->shadowRows = (size_t)w->dw.nBodies;
}

// Behind SynchronizeFixtures the rows of all bodies but those the TOI phase will still move are final: a large world sends
// them now, on a second stream, under the pair update and the TOI phase (40 bytes per body over PCIe: 0.75 ms for a million
// bodies, the longest single item of that step), and k_end_step sends the rows that changed since (its shadow comparison).
// (two halves: the point on the main stream from which the rows may be read is marked right behind SynchronizeFixtures; the
// launches on the second stream are issued once the host has queued the pair search - or the main stream would wait for
// the host to get through these calls)
static int forkEarlyRows(b2hip_world* w)
{
	DW& d = w->dw;
	const bool lazy = w->lazyReadback || (w->spatial && !w->spFullRows);
	if (w->earlyRowsMin <= 0 || d.nBodies < w->earlyRowsMin || w->noStatePoll || lazy || w->rowsEarlyPending || w->rowsForked || w->debugSync || w->debugTrace) return 0;
	// (a user contact filter is called by the pair update while the copy would be running: a ShouldCollide that reads a body
	// - pullBody reads h_state - could see a row half old, half new. No early rows then: ADVICE round 4.)
	if (hasFilter(w)) return 0;
	if (!w->rowStream)
	{
		HIP_TRY(hipStreamCreateWithFlags(&w->rowStream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&w->rowFork, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&w->rowJoin, hipEventDisableTiming));
	}
	HIP_TRY(hipEventRecord(w->rowFork, w->stream));
	w->rowsForked = true;
	return 0;
}

static int startEarlyRows(b2hip_world* w)
{
	DW& d = w->dw;
	if (!w->rowsForked) return 0;
	w->rowsForked = false;
	HIP_TRY(hipStreamWaitEvent(w->rowStream, w->rowFork, 0));
	// The rows are gathered into the device's copy (stateOut: ~30 us for a million bodies) and leave from there by a copy - the
	// DMA engine's, which does not stand in the way of the kernels running meanwhile. (Stores from a kernel straight into host
	// memory, as k_end_step's are, do: with enough of them in flight to fill the link, the pair update beside them ran 1.5 x
	// slower - measured, 1 M bodies: 3.77 ms per step without the early launch, 3.29 at best with such a kernel, 3.11 with the copy.)
	DW dEarly = d;
	dEarly.stampMask = 0u; // (the phase stamps belong to the main stream's next kernel)
	hipLaunchKernelGGL(k_end_step, dim3(gridFor(d.nBodies)), dim3(256), 0, w->rowStream, dEarly, 0, (const int*)nullptr, w->stateOut.p, 0, END_STEP_EARLY, (float*)nullptr, 0);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(w->h_state, w->stateOut.p, (size_t)d.nBodies * 10 * sizeof(float), hipMemcpyDeviceToHost, w->rowStream));
	HIP_TRY(hipEventRecord(w->rowJoin, w->rowStream));
	shadowWritten(w);
	w->rowsEarlyPending = true;
	return 0;
}

static int downloadState(b2hip_world* w, int clearForces, bool skipRowsIfRedo)
{
	DW& d = w->dw;
	const size_t nb = w->bodies.size();
	w->rowsForked = false; // (marked, never launched: a step without a pair update)
	if (w->rowsEarlyPending)
	{
		HIP_TRY(hipStreamWaitEvent(w->stream, w->rowJoin, 0));
		w->rowsEarlyPending = false;
	}
	// (lazy: every read-back of a step end leaves the rows where they are; a read-back outside a step is a full one)
	// (... and so does a spatially sharded world with the lean exchange: the rows of the bodies THIS rank owns go to the host
	// packed - k_end_step, DW::spOwnOut - the table of all rows on demand)
	const bool lazy = (w->lazyReadback || (w->spatial && !w->spFullRows)) && w->stepActive && !w->noStatePoll;
	w->rowsPending.store(lazy, std::memory_order_release);
	w->stateSeq = (w->stateSeq + 1) & 0x3fffffff;
	if (w->stateSeq == 0) w->stateSeq = 1;
	((DState*)(w->h_state + B2D_STATE_TAIL(nb)))->pubSeq = 0; // (whatever was there: not this number)
	const int clear = clearForces < 0 ? w->def.auto_clear_forces : clearForces;
	if (w->noStatePoll)
	{
		// B2HIP_NO_STATE_POLL=1, for comparison: into the device staging array, one copy, stream synchronisation
		LAUNCH(w, k_end_step, gridFor(d.nBodies), 256, d, clear, (const int*)w->gridBar.p, w->stateOut.p, w->stateSeq, 0, (float*)nullptr, 0);
		HIP_TRY(hipMemcpyAsync(w->h_state, w->stateOut.p, B2D_STATE_TAIL(nb) * sizeof(float) + sizeof(DState), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		memcpy(w->h_dstate, w->h_state + B2D_STATE_TAIL(nb), offsetof(DState, pubSeq));
		w->shadowDev = nullptr; // (the staging array is the shadow's memory)
		return 0;
	}
	const int rowMode = shadowValid(w) ? 2 : 1;
	LAUNCH(w, k_end_step, gridFor(d.nBodies), 256, d, clear, (const int*)w->gridBar.p, w->d_hstate, w->stateSeq,
		lazy ? END_STEP_LAZY : skipRowsIfRedo ? END_STEP_SKIP_IF_REDO : END_STEP_FULL, w->stateOut.p, rowMode);
	const int rc = awaitState(w, nb);
	// (rowsSkipped: 0 - the rows were stored; the shadow of a full write is valid from here on)
	if (rc == 0 && rowMode == 1 && w->h_dstate->c.rowsSkipped == 0) shadowWritten(w);
	return rc;
}

// The rows a lazy step end left on the device (b2hip_set_lazy_readback), fetched when the first caller asks for a body's
// state: the row half of k_end_step on its own. Several user threads may ask at once (b2Body getters from range tasks).
static int fetchRows(b2hip_world* w)
{
	DEVICE_GUARD(w);
	DW& d = w->dw;
	const size_t nb = (size_t)d.nBodies; // (bodies created since the step are not on the device yet: the step's count)
	w->stateSeq = (w->stateSeq + 1) & 0x3fffffff;
	if (w->stateSeq == 0) w->stateSeq = 1;
	((DState*)(w->h_state + B2D_STATE_TAIL(nb)))->pubSeq = 0;
	const int rowMode = shadowValid(w) ? 2 : 1;
	LAUNCH(w, k_end_step, gridFor(d.nBodies), 256, d, 0, (const int*)nullptr, w->d_hstate, w->stateSeq, END_STEP_ROWS, w->stateOut.p, rowMode);
	const int rc = pollPublished(w, (volatile const int*)&((const DState*)(w->h_state + B2D_STATE_TAIL(nb)))->pubSeq, w->stateSeq, "lazy state read-back");
	if (rc == 0 && rowMode == 1) shadowWritten(w);
	return rc;
}

static void ensureRows(b2hip_world* w)
{
	if (!w->rowsPending.load(std::memory_order_acquire)) return;
	std::lock_guard<std::mutex> lock(w->rowsMutex);
	if (!w->rowsPending.load(std::memory_order_relaxed)) return;
	if (fetchRows(w) != 0)
	{
		// (the rows cannot be had: the world is as good as lost - every later call says why)
		w->failed = true;
		w->failedWhy = g_lastError;
	}
	w->rowsPending.store(false, std::memory_order_release);
}

static void refreshMirror(b2hip_world* w)
{
	// h_state now holds the state of every body; HostBody rows are pulled from it on demand (pullBody)
	w->stateCount = w->bodies.size();
	++w->mirrorEpoch;
	++w->stepEpoch;
}

// A phase that fails leaves the device state half-stepped: the world unlocks (so that it can still be inspected and
// destroyed) and every later call reports the failure instead of stepping on.
static int stepFailed(b2hip_world* w, int rc)
{
	if (rc)
	{
		w->stepActive = false;
		w->failed = true;
		w->failedWhy = g_lastError;
	}
	return rc;
}

static int checkUsable(b2hip_world* w, const char* what, bool mutator)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	if (w->failed) return setError(B2HIP_ERR_INVALID, std::string(what) + ": the world is in a failed state (" + w->failedWhy + ")");
	if (mutator && w->stepActive && !w->callbackWindow) return setError(B2HIP_ERR_INVALID, std::string(what) + " inside a step");
	return 0;
}

static int addJoint(b2hip_world* w, const JointRec& j)
{
	if (int rcu = checkUsable(w, "b2hip_create_joint", true)) return rcu;
	w->joints.push_back(j);
	if (w->spatial) w->spOwnersDirty = true; // (a joint may join bodies of different owners: resolved at the next step)
	// b2World::CreateJoint (b2World.cpp:716-732): contacts between the two bodies are re-filtered
	if (j.collideConnected == 0) w->pendingFilter.push_back(std::make_pair(j.bodyA, j.bodyB));
	return (int)w->joints.size() - 1;
}

extern "C"
{

const char* b2hip_last_error(void)
{
	return g_lastError.c_str();
}

const char* b2hip_version(void)
{
	return "b2hip 0.1 (gfx950)";
}

int b2hip_world_create(const b2hip_world_def* def, b2hip_world** out)
{
	if (!def || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
	{
		return setError(B2HIP_ERR_NO_DEVICE, "no HIP device available: the b2hip Step() path has no CPU fallback");
	}
	int current = 0;
	if (hipGetDevice(&current) != hipSuccess) current = 0;
	const int device = def->device >= 0 ? def->device : current;
	if (device >= count) return setError(B2HIP_ERR_NO_DEVICE, "no such HIP device");
	b2hip_world* w = new b2hip_world();
	w->def = *def;
	w->device = device; // the ordinal itself, also when the caller asked for "current": later calls select it again
	DEVICE_GUARD(w);
	{
		int now = -1;
		if (hipGetDevice(&now) != hipSuccess || now != device)
		{
			delete w;
			return setError(B2HIP_ERR_NO_DEVICE, "hipSetDevice failed");
		}
	}
	e = hipStreamCreateWithFlags(&w->stream2, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&w->evFork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&w->evJoin, hipEventDisableTiming);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
	if (e != hipSuccess)
	{
		delete w;
		return setError(B2HIP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
	}
	w->debugSync = getenv("B2HIP_DEBUG") != nullptr;
	w->forceLarge = getenv("B2HIP_FORCE_LARGE") ? atoi(getenv("B2HIP_FORCE_LARGE")) : 0;
	w->nextNode = 0;
	w->leafCount = 0;
	w->upBodies = w->upFixtures = w->upShapes = w->upJoints = 0;
	w->stateCount = 0;
	w->mirrorEpoch = 1;
	w->stepEpoch = 1;
	w->newFixture = false;
	w->inv_dt0 = 0.0f;
	w->stepActive = false;
	w->callbackWindow = false;
	w->h_state = nullptr;
	w->h_stateCap = 0;
	w->lastContacts = 0;
	memset(&w->last, 0, sizeof(w->last));
	memset(&w->dw, 0, sizeof(w->dw));
	memset(w->profile, 0, sizeof(w->profile));
	w->solverMs = 0.0f;
	w->solverBytes = 0.0;
	w->solverConstraints = w->solverBodies = 0;
	w->kernelTiming = 0;
	w->ktUnitsA = w->ktUnitsB = 0;
	w->ktUsed = 0;
	w->ktKind = 0;
	w->ktMs = 0.0f;
	w->ktLaunches = 0;
	w->ktBytes = 0.0;
	w->dw.cellSize = 1.0f;
	w->dw.invCellSize = 1.0f;
	w->toiRan = false;
	w->toiEventValid = false;
	w->debugTrace = getenv("B2HIP_TRACE") != nullptr;
	w->kernelTimingLaunches = getenv("B2HIP_SOLVER_LAUNCHES") != nullptr; // force the launch-per-colour solver
	w->solverBarriers = getenv("B2HIP_SOLVER_BARRIERS") != nullptr;       // persistent kernel with a grid barrier per colour instead of body-level dataflow
	w->persistSteps = 0;
	w->colorSmallPending = false;
	w->hubSteps = 0;
	w->useGraphs = getenv("B2HIP_GRAPHS") != nullptr; // opt-in: measured no gain on MI355X (the step is not host-bound), see DESIGN.md
	w->graphCaptures = 0;
	w->persistMaxWG = 0;
	w->nCU = 256;
	{
		hipDeviceProp_t prop;
		int devId = 0;
		if (hipGetDevice(&devId) == hipSuccess && hipGetDeviceProperties(&prop, devId) == hipSuccess)
		{
			w->nCU = prop.multiProcessorCount;
#if B2HIP_HAVE_VALIDATION_SOLVERS
			int perCU = 0, perCU2 = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_solve_dataflow, PERSIST_LANES, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU2, k_solve_mailbox<true>, PERSIST_LANES, 0) == hipSuccess)
			{
				perCU = std::min(perCU, perCU2);
				// the occupancy query can be one block per CU high (sgpr_count 81-112, MI355X_MICROARCH.md): keep a margin
				w->persistMaxWG = std::max(0, std::min(perCU - 1, 4)) * prop.multiProcessorCount;
			}
#else
			w->persistMaxWG = 1 << 24; // (a limit of the cross-check solvers; k_solve_blocks has its own: blocksMaxWG)
#endif
		}
	}
	w->dfLanesForced = 0;
	w->noSideStream = getenv("B2HIP_NO_SIDE_STREAM") != nullptr;
	w->profileDetail = !(getenv("B2HIP_PROFILE_DETAIL") && atoi(getenv("B2HIP_PROFILE_DETAIL")) == 0);
	w->collideSortEnv = getenv("B2HIP_COLLIDE_SORT") ? atoi(getenv("B2HIP_COLLIDE_SORT")) : -1;
	w->collideStage = getenv("B2HIP_COLLIDE_STAGE") ? atoi(getenv("B2HIP_COLLIDE_STAGE")) : -1;
	w->dfEpoch = 0;
	// single-XCD attempt of k_solve_mailbox: opt-in. Measured on the 10k-body pyramid it LOSES (launch 500 us against 368):
	// 334 waves polling on 32 CUs load the consumer CUs' memory queues, which is where a hand-off is priced; L2 locality
	// buys only 0.1-0.3 us of it (MI355X_MICROARCH.md, handoff-1to1)
	w->solverLocal = getenv("B2HIP_SOLVER_SINGLE_XCD") != nullptr;
	w->solverRows = getenv("B2HIP_SOLVER_ROWS") != nullptr; // polled body rows (k_solve_dataflow) instead of pushed mailboxes
	w->solverMailbox = getenv("B2HIP_SOLVER_MAILBOX") != nullptr; // pushed hand-offs for every constraint (k_solve_mailbox) instead of k_solve_blocks
	w->noSweepBlocks = getenv("B2HIP_NO_SWEEP_BLOCKS") != nullptr;
	w->tracePartition = getenv("B2HIP_TRACE_PARTITION") != nullptr;
	if (const char* e = getenv("B2HIP_GRID_HALF")) { w->gridForced = true; w->gridHalf = atoi(e) != 0; w->dw.gridHalf = w->gridHalf ? 1 : 0; }
	w->dw.noChainCreate = getenv("B2HIP_TOI_NO_CHAIN_CREATE") != nullptr ? 1 : 0;
	w->traceLaunches = getenv("B2HIP_TRACE_LAUNCHES") != nullptr;
	w->noBlocks = getenv("B2HIP_NO_BLOCKS") != nullptr;           // no block partition at all (colours as before it existed)
	w->blockLanes = 0; // chosen per partition (see phaseSolve); B2HIP_BLOCK_LANES = 256 | 512 | 1024 forces one size
	if (const char* e = getenv("B2HIP_BLOCK_LANES")) w->blockLanes = atoi(e) == 512 ? 512 : (atoi(e) == 256 ? 256 : (atoi(e) == 1024 ? 1024 : 0));
	{
		int perCU = 0, perCU2 = 0, perCU3 = 0;
		hipDeviceProp_t prop;
		int devId = 0;
		if (hipGetDevice(&devId) == hipSuccess && hipGetDeviceProperties(&prop, devId) == hipSuccess &&
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_solve_blocks<256>, 256, 0) == hipSuccess &&
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU2, k_solve_blocks<512>, 512, 0) == hipSuccess &&
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU3, k_solve_blocks<BLOCK_LANES>, BLOCK_LANES, 0) == hipSuccess &&
			perCU > 0 && perCU2 > 0 && perCU3 > 0)
		{
			// one block per CU is all this sizing relies on (the occupancy query can be one too high, MI355X_MICROARCH.md)
			w->blocksMaxWG = prop.multiProcessorCount - 8;
			int s0 = 0, s1 = 0, s2 = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&s0, k_blocks_sweep<256>, 256, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&s1, k_blocks_sweep<512>, 512, 0) == hipSuccess &&
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&s2, k_blocks_sweep<BLOCK_LANES>, BLOCK_LANES, 0) == hipSuccess)
			{
				// (the blocks of a launch wait for one another: all of them must be resident together)
				w->sweepMaxWG[0] = std::max(1, s0 - 1) * w->blocksMaxWG;
				w->sweepMaxWG[1] = std::max(1, s1 - 1) * w->blocksMaxWG;
				w->sweepMaxWG[2] = std::max(1, s2 - 1) * w->blocksMaxWG;
			}
		}
	}
	w->dfSleep = 1;
	if (const char* e = getenv("B2HIP_DF_LANES")) w->dfLanesForced = std::max(64, std::min(256, atoi(e) / 64 * 64));
	if (const char* e = getenv("B2HIP_DF_SLEEP")) w->dfSleep = atoi(e);
	w->toiChains = false;
	w->toiSerialOnly = getenv("B2HIP_TOI_SERIAL") != nullptr;
	w->toiSyncOnly = getenv("B2HIP_TOI_SYNC") != nullptr;
	w->toiNoDomains = getenv("B2HIP_TOI_NO_DOMAINS") != nullptr; // bullets / kinematic partners through the serial loop only // decide chains / serial loop from a read-back after k_toi_first (the older flow)
	w->toiFallbacks = 0;
	for (int i = 0; i < 13; ++i) w->ev[i] = nullptr;
	w->h_dstate = nullptr;
	for (int i = 0; i < 13; ++i)
	{
		if (hipEventCreate(&w->ev[i]) != hipSuccess)
		{
			b2hip_world_destroy(w);
			return setError(B2HIP_ERR_HIP, "hipEventCreate failed");
		}
	}
	if (hipHostMalloc((void**)&w->h_dstate, sizeof(DState), hipHostMallocDefault) != hipSuccess)
	{
		b2hip_world_destroy(w);
		return setError(B2HIP_ERR_HIP, "hipHostMalloc failed");
	}
	// (written by a kernel, polled by the host: mapped and coherent)
	if (hipHostMalloc((void**)&w->h_pub, sizeof(DState), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
		hipHostGetDevicePointer((void**)&w->d_pub, w->h_pub, 0) != hipSuccess)
	{
		b2hip_world_destroy(w);
		return setError(B2HIP_ERR_HIP, "hipHostMalloc (coherent) failed");
	}
	memset(w->h_pub, 0, sizeof(DState));
	w->noCensusPoll = getenv("B2HIP_NO_CENSUS_POLL") && atoi(getenv("B2HIP_NO_CENSUS_POLL"));
	w->noStatePoll = getenv("B2HIP_NO_STATE_POLL") && atoi(getenv("B2HIP_NO_STATE_POLL"));
	w->lazyReadback = getenv("B2HIP_LAZY_READBACK") && atoi(getenv("B2HIP_LAZY_READBACK"));
	if (getenv("B2HIP_EARLY_ROWS_MIN")) w->earlyRowsMin = atoi(getenv("B2HIP_EARLY_ROWS_MIN"));
	int rc = ensureCapacity(w, 0);
	if (rc == 0 && hipStreamSynchronize(w->stream) != hipSuccess) rc = setError(B2HIP_ERR_HIP, "stream sync failed");
	if (rc)
	{
		const std::string why = g_lastError;
		b2hip_world_destroy(w);
		return setError(rc, why);
	}
	*out = w;
	return B2HIP_OK;
}

void b2hip_world_destroy(b2hip_world* w)
{
	if (!w) return;
	DEVICE_GUARD(w);
	if (w->stream) (void)hipStreamSynchronize(w->stream);
	if (w->rowStream)
	{
		(void)hipStreamSynchronize(w->rowStream);
		(void)hipStreamDestroy(w->rowStream);
		(void)hipEventDestroy(w->rowFork);
		(void)hipEventDestroy(w->rowJoin);
		w->rowStream = nullptr;
	}
	if (w->shardComm != nullptr && g_rcclDestroy != nullptr) g_rcclDestroy(w->shardComm);
	w->shardComm = nullptr;
	w->shardSend.release(); w->shardRecv.release();
	w->b_owner.release(); w->spNewOwner.release(); w->spAwake.release(); w->spStraddle.release(); w->spCount.release(); w->spTarget.release();
	w->spSend.release(); w->spRecv.release(); w->spTailKey.release(); w->spVirt.release();
	if (w->spHost) (void)hipHostFree(w->spHost);
	w->spHost = nullptr;
	if (w->spOwnHost) (void)hipHostFree(w->spOwnHost);
	w->spOwnHost = nullptr;
	if (w->spHdrHost) (void)hipHostFree(w->spHdrHost);
	w->spHdrHost = nullptr;
	for (size_t k = 0; k < w->spTape.size(); ++k) (void)hipFree(w->spTape[k].first);
	w->spTape.clear();
	w->d_state.release();
	w->b_pos.release(); w->b_pos0.release(); w->b_vel.release(); w->b_xf.release(); w->b_mass.release(); w->b_damp.release();
	w->b_force.release(); w->b_flags.release(); w->b_wake.release();
	w->p_fat.release(); w->p_body.release(); w->p_shape.release(); w->p_key.release(); w->p_filter0.release(); w->p_filter1.release();
	w->p_mat.release(); w->d_shapes.release();
	for (int k = 0; k < 2; ++k)
	{
		w->c_ids[k].release(); w->c_key[k].release(); w->c_flags[k].release(); w->c_mat[k].release(); w->c_man0[k].release();
		w->c_man1[k].release(); w->c_imp[k].release(); w->c_man3[k].release(); w->c_color[k].release();
	}
	w->ht_keys.release(); w->d_joints.release(); w->d_gears.release(); w->li_ref.release();
	w->jadjStart.release(); w->jadj.release(); w->rootJointStart.release(); w->rootJointCursor.release();
	w->lj_list.release(); w->rootJointOkay.release();
	w->parent.release(); w->rootSeed.release(); w->rootBodies.release(); w->rootContacts.release(); w->rootJoints.release();
	w->rootIsland.release(); w->deg.release(); w->adjStart.release(); w->adjCursor.release(); w->adj.release(); w->adjSlot.release();
	w->rootScanIn.release(); w->rootScanOut.release();
	w->si_root.release(); w->si_bodyStart.release(); w->si_contactStart.release(); w->si_wStart.release(); w->si_maxLevel.release();
	w->si_bodies.release(); w->si_contacts.release(); w->si_level.release(); w->si_stack.release(); w->si_lastLevel.release();
	w->b_slot.release(); w->b_island.release(); w->chunkFirst.release();
	w->li_bodies.release(); w->li_contacts.release(); w->li_roots.release(); w->li_color.release(); w->colorCount.release();
	w->colorStart.release(); w->colorCursor.release(); w->li_sorted.release(); w->bodyClaim.release(); w->rootPen.release();
	w->bodyActive.release(); w->bodyRest.release(); w->b_posv.release(); w->dfRank.release(); w->dfInbox.release(); w->evKey.release(); w->evInfo.release(); w->uncolList.release(); w->compactList.release(); w->gridBar.release(); w->hubRowOf.release(); w->hubList.release(); w->hubDelta.release(); w->hubMeta.release(); w->hubFirst.release();
	w->b_proxyHead.release(); w->p_next.release(); w->toiList.release(); w->toiPos2c.release(); w->toiDestroyList.release(); w->toiNewList.release();
	w->b_toiGroup.release(); w->toiGroups.release(); w->toiGroupCount.release(); w->toiGroupList.release(); w->toiMoved.release(); w->toiNew.release(); w->toiParent.release(); w->toiDomOf.release(); w->toiDomRoot.release(); w->toiDomCount.release(); w->toiDomBase.release(); w->toiDomFill.release(); w->toiDomFailed.release(); w->toiDomEvents.release(); w->toiDomList.release(); w->toiHull.release(); w->snapBody.release(); w->snapFat.release();
	w->c_mgr[0].release(); w->c_mgr[1].release(); w->dbgPreVel.release(); w->dbgVel.release(); w->dbgLi.release();
	w->rootSleepMin.release(); w->bodyColorMask.release(); w->rootDone.release(); w->lc.release();
	w->moveBuf.release(); w->gridCount.release(); w->gridStart.release(); w->gridCursor.release(); w->gridItems.release(); w->gridFat.release(); w->arriveTree.release();
	w->largeProxies.release(); w->largeMoves.release(); w->pairKey.release(); w->pairKey2.release(); w->pairProxy.release(); w->pairProxy2.release();
	w->filterPairs.release();
	w->d_editOps.release();
	w->b_order.release(); w->orderBody.release(); w->bigRoots.release();
	w->pre_o0.release(); w->pre_o1.release(); w->pre_oimp.release(); w->pre_o3.release(); w->preRecs.release();
	w->postRecs.release(); w->filterList.release(); w->hostList.release(); w->toiLog.release(); w->toiVerdict.release();
	w->b_blk1.release(); w->b_adopt.release(); w->b_adoptStage.release(); w->blkRows.release(); w->blkRowStart.release(); w->blkCursor.release(); w->blkBodyCount.release(); w->blkBodyCursor.release();
	w->blkBodyStart.release(); w->blkBodies.release(); w->rowColor.release(); w->b_cutv.release();
	w->pairFirst.release(); w->pairRank.release(); w->scanTmp.release(); w->radixHist.release(); w->radixHistScan.release();
	w->keepFlag.release(); w->keepScan.release(); w->scanTmp4.release(); w->scanFlags.release(); w->stateOut.release(); w->consts.release();
	if (w->h_state) (void)hipHostFree(w->h_state);
	if (w->h_dstate) (void)hipHostFree(w->h_dstate);
	if (w->h_pub) (void)hipHostFree(w->h_pub);
	for (int i = 0; i < 13; ++i)
		if (w->ev[i]) (void)hipEventDestroy(w->ev[i]);
	for (size_t i = 0; i < w->ktEvents.size(); ++i) (void)hipEventDestroy(w->ktEvents[i]);
	{
		GraphSeg* segs[3] = { &w->segCollide, &w->segIslands, &w->segPairs };
		for (int i = 0; i < 3; ++i)
		{
			if (segs[i]->exec) (void)hipGraphExecDestroy(segs[i]->exec);
			if (segs[i]->graph) (void)hipGraphDestroy(segs[i]->graph);
		}
	}
	if (w->stream) (void)hipStreamDestroy(w->stream);
	if (w->stream2) (void)hipStreamDestroy(w->stream2);
	if (w->evFork) (void)hipEventDestroy(w->evFork);
	if (w->evJoin) (void)hipEventDestroy(w->evJoin);
	delete w;
}

int b2hip_set_gravity(b2hip_world* w, float gx, float gy)
{
	if (int rcu = checkUsable(w, "b2hip_set_gravity", true)) return rcu;
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->def.gravity_x = gx;
	w->def.gravity_y = gy;
	return 0;
}

int b2hip_set_flags(b2hip_world* w, int allow_sleep, int warm_starting, int continuous, int sub_stepping)
{
	if (int rcu = checkUsable(w, "b2hip_set_flags", true)) return rcu;
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->def.allow_sleep = allow_sleep;
	w->def.warm_starting = warm_starting;
	w->def.continuous = continuous;
	w->def.sub_stepping = sub_stepping;
	return 0;
}

int b2hip_create_body(b2hip_world* w, const b2hip_body_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rcu = checkUsable(w, "b2hip_create_body", true)) return rcu;
	HostBody b{};
	b.type = def->type;
	b.flags = 0;
	if (def->bullet) b.flags |= BF_BULLET;
	if (def->fixed_rotation) b.flags |= BF_FIXEDROT;
	if (def->allow_sleep) b.flags |= BF_AUTOSLEEP;
	if (def->awake) b.flags |= BF_AWAKE;
	if (def->active) b.flags |= BF_ACTIVE;
	b.px = def->px;
	b.py = def->py;
	b.qs = sinf(def->angle); // b2Rot::Set (b2Math.h:294-299)
	b.qc = cosf(def->angle);
	b.lcx = b.lcy = 0.0f;
	b.c0x = b.cx = def->px;
	b.c0y = b.cy = def->py;
	b.a0 = b.a = def->angle;
	b.vx = def->vx;
	b.vy = def->vy;
	b.w = def->w;
	b.linearDamping = def->linear_damping;
	b.angularDamping = def->angular_damping;
	b.gravityScale = def->gravity_scale;
	b.fx = b.fy = b.torque = 0.0f;
	b.sleepTime = 0.0f;
	b.pullEpoch = w->mirrorEpoch;
	b.forceEpoch = w->stepEpoch;
	if (def->type == B2HIP_DYNAMIC_BODY)
	{
		b.mass = 1.0f;
		b.invMass = 1.0f;
	}
	else
	{
		b.mass = 0.0f;
		b.invMass = 0.0f;
	}
	b.I = 0.0f;
	b.invI = 0.0f;
	b.dirty = true;
	b.worldIndex = -1;
	if (def->type != B2HIP_STATIC_BODY)
	{
		// b2World::CreateBody (b2World.cpp:571-575): appended to m_nonStaticBodies
		b.worldIndex = (int)w->nonStatic.size();
		w->nonStatic.push_back((int)w->bodies.size());
		w->orderDirty = true;
	}
	w->bodies.push_back(b);
	if (w->spatial) w->spOwnersDirty = true; // (the new body falls into the strip of its x at the next step)
	w->dirtyList.push_back((int)w->bodies.size() - 1);
	return (int)w->bodies.size() - 1;
}

int b2hip_create_fixture(b2hip_world* w, int body, const b2hip_fixture_def* def, const b2hip_shape* shape)
{
	if (!w || !def || !shape) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rcu = checkUsable(w, "b2hip_create_fixture", true)) return rcu;
	if (body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad body id");
	if (shape->type != B2HIP_SHAPE_CIRCLE && shape->type != B2HIP_SHAPE_EDGE && shape->type != B2HIP_SHAPE_POLYGON && shape->type != B2HIP_SHAPE_CHAIN)
	{
		return setError(B2HIP_ERR_INVALID, "unknown shape type");
	}
	ShapeRec rec;
	memset(&rec, 0, sizeof(rec));
	rec.type = shape->type;
	rec.count = shape->count;
	rec.radius = shape->radius;
	rec.centroid = v2(shape->centroid[0], shape->centroid[1]);
	int nv = shape->type == B2HIP_SHAPE_POLYGON ? shape->count : (B2D_IS_SEGMENT(shape->type) ? 4 : 1);
	if (nv > B2D_MAX_POLY_VERTS) return setError(B2HIP_ERR_INVALID, "too many polygon vertices");
	for (int i = 0; i < nv; ++i)
	{
		rec.verts[i] = v2(shape->verts[2 * i], shape->verts[2 * i + 1]);
		if (shape->type == B2HIP_SHAPE_POLYGON) rec.normals[i] = v2(shape->normals[2 * i], shape->normals[2 * i + 1]);
	}
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	HostFixture f;
	memset(&f, 0, sizeof(f));
	f.body = body;
	f.shape = internShape(w, rec);
	f.density = def->density;
	f.friction = def->friction;
	f.restitution = def->restitution;
	f.categoryBits = def->category_bits;
	f.maskBits = def->mask_bits;
	f.groupIndex = def->group_index;
	f.isSensor = def->is_sensor != 0;
	f.thick = def->thick_shape != 0;
	// b2Fixture::CreateProxies (b2Fixture.cpp:126-141) + b2DynamicTree::CreateProxy (b2DynamicTree.cpp:105-119)
	AABB aabb = b2dShapeAABB(&rec, hostXf(b));
	f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
	f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
	f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
	f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
	const bool bodyActive = (b.flags & BF_ACTIVE) != 0;
	if (bodyActive)
	{
		f.proxyKey = allocProxyKey(w);
		if (f.proxyKey < 0) return setError(B2HIP_ERR_UNSUPPORTED, "proxy id reuse after the broad-phase tree was emptied is not modelled");
	}
	else
	{
		// (b2Body.cpp:199-203: an inactive body's fixtures get their proxies when it is activated)
		f.proxyKey = -1;
		f.noProxy = true;
	}
	const int id = (int)w->fixtures.size();
	w->fixtures.push_back(f);
	b.fixtures.push_back(id);
	if (bodyActive) w->pendingMoves.push_back(id);
	if (f.density > 0.0f)
	{
		resetMassData(w, b);
	}
	w->newFixture = true;
	return id;
}

int b2hip_create_revolute_joint(b2hip_world* w, const b2hip_revolute_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	RevoluteJoint j;
	memset(&j, 0, sizeof(j));
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.referenceAngle = def->reference_angle;
	j.enableLimit = def->enable_limit;
	j.lowerAngle = def->lower_angle;
	j.upperAngle = def->upper_angle;
	j.enableMotor = def->enable_motor;
	j.motorSpeed = def->motor_speed;
	j.maxMotorTorque = def->max_motor_torque;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_distance_joint(b2hip_world* w, const b2hip_distance_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_DISTANCE;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.length = def->length;
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_prismatic_joint(b2hip_world* w, const b2hip_prismatic_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_PRISMATIC;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.localAxisA = v2(def->local_axis_a[0], def->local_axis_a[1]);
	b2dNormalize(j.localAxisA); // b2PrismaticJoint.cpp:104
	j.referenceAngle = def->reference_angle;
	j.enableLimit = def->enable_limit;
	j.lowerTranslation = def->lower_translation;
	j.upperTranslation = def->upper_translation;
	j.enableMotor = def->enable_motor;
	j.motorSpeed = def->motor_speed;
	j.maxMotorForce = def->max_motor_force;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_weld_joint(b2hip_world* w, const b2hip_weld_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nb = (int)w->bodies.size();
	if (def->body_a < 0 || def->body_a >= nb || def->body_b < 0 || def->body_b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_WELD;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.referenceAngle = def->reference_angle;
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

static int checkJointBodies(b2hip_world* w, int a, int b)
{
	const int nb = w ? (int)w->bodies.size() : 0;
	if (!w) return setError(B2HIP_ERR_INVALID, "null argument");
	if (a < 0 || a >= nb || b < 0 || b >= nb) return setError(B2HIP_ERR_INVALID, "bad body id");
	return 0;
}

int b2hip_create_wheel_joint(b2hip_world* w, const b2hip_wheel_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_WHEEL;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.localAxisA = v2(def->local_axis_a[0], def->local_axis_a[1]);
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.enableMotor = def->enable_motor;
	j.motorSpeed = def->motor_speed;
	j.maxMotorTorque = def->max_motor_torque;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_rope_joint(b2hip_world* w, const b2hip_rope_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_ROPE;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.maxLength = def->max_length;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_friction_joint(b2hip_world* w, const b2hip_friction_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_FRICTION;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.maxForce = def->max_force;
	j.maxTorque = def->max_torque;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_motor_joint(b2hip_world* w, const b2hip_motor_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_MOTOR;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.linearOffset = v2(def->linear_offset[0], def->linear_offset[1]);
	j.angularOffset = def->angular_offset;
	j.maxForce = def->max_force;
	j.maxTorque = def->max_torque;
	j.correctionFactor = def->correction_factor;
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_pulley_joint(b2hip_world* w, const b2hip_pulley_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	if (def->ratio == 0.0f) return setError(B2HIP_ERR_INVALID, "pulley ratio must not be zero");
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_PULLEY;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.localAnchorA = v2(def->local_anchor_a[0], def->local_anchor_a[1]);
	j.localAnchorB = v2(def->local_anchor_b[0], def->local_anchor_b[1]);
	j.groundAnchorA = v2(def->ground_anchor_a[0], def->ground_anchor_a[1]);
	j.s1 = def->ground_anchor_b[0];
	j.s2 = def->ground_anchor_b[1];
	j.ratio = def->ratio;
	j.constant = def->length_a + def->ratio * def->length_b; // b2PulleyJoint.cpp:75
	j.collideConnected = def->collide_connected;
	return addJoint(w, j);
}

int b2hip_create_mouse_joint(b2hip_world* w, const b2hip_mouse_joint_def* def)
{
	if (!def) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rc = checkJointBodies(w, def->body_a, def->body_b)) return rc;
	HostBody& bB = w->bodies[def->body_b];
	if (!bB.dirty) pullBody(w, def->body_b); // current transform of bodyB
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_MOUSE;
	j.bodyA = def->body_a;
	j.bodyB = def->body_b;
	j.targetA = v2(def->target[0], def->target[1]);
	// m_localAnchorB = b2MulT(bodyB transform, target) (b2MouseJoint.cpp:45)
	const float px = def->target[0] - bB.px, py = def->target[1] - bB.py;
	j.localAnchorB = v2(bB.qc * px + bB.qs * py, -bB.qs * px + bB.qc * py);
	j.bodyMass = bB.mass;
	j.maxForce = def->max_force;
	j.frequencyHz = def->frequency_hz;
	j.dampingRatio = def->damping_ratio;
	j.collideConnected = def->collide_connected;
	w->nMouseJoints += 1;
	return addJoint(w, j);
}

int b2hip_joint_set_target(b2hip_world* w, int joint, float x, float y)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_target", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_MOUSE) return setError(B2HIP_ERR_INVALID, "not a mouse joint");
	if (x == j.targetA.x && y == j.targetA.y) return 0;
	setAwake(w, j.bodyB);
	j.targetA = v2(x, y);
	w->jointEdits.push_back(std::make_pair(joint, 2));
	return 0;
}

// b2GearJoint::b2GearJoint (b2GearJoint.cpp:50-129): everything is derived from the two joints and the bodies' current poses
static float gearCoordinate(b2hip_world* w, const JointRec& jt, int moving, int fixed)
{
	const HostBody& bm = w->bodies[moving];
	const HostBody& bf = w->bodies[fixed];
	if (jt.type == B2D_JOINT_REVOLUTE) return bm.a - bf.a - jt.referenceAngle;
	// pA = b2MulT(xfC.q, b2Mul(xfA.q, m_localAnchorA) + (xfA.p - xfC.p)); coordinate = b2Dot(pA - pC, m_localAxisC)
	const V2 la = jt.localAnchorB, lc = jt.localAnchorA;
	const V2 wa = v2(bm.qc * la.x - bm.qs * la.y, bm.qs * la.x + bm.qc * la.y) + v2(bm.px - bf.px, bm.py - bf.py);
	const V2 pa = v2(bf.qc * wa.x + bf.qs * wa.y, -bf.qs * wa.x + bf.qc * wa.y);
	return b2dDot(pa - lc, jt.localAxisA);
}

int b2hip_create_gear_joint(b2hip_world* w, const b2hip_gear_joint_def* def)
{
	if (!w || !def) return setError(B2HIP_ERR_INVALID, "null argument");
	const int nj = (int)w->joints.size();
	if (def->joint1 < 0 || def->joint1 >= nj || def->joint2 < 0 || def->joint2 >= nj) return setError(B2HIP_ERR_INVALID, "bad joint id");
	const JointRec j1 = w->joints[def->joint1], j2 = w->joints[def->joint2];
	if ((j1.type != B2D_JOINT_REVOLUTE && j1.type != B2D_JOINT_PRISMATIC) || (j2.type != B2D_JOINT_REVOLUTE && j2.type != B2D_JOINT_PRISMATIC))
		return setError(B2HIP_ERR_INVALID, "a gear joint connects revolute and / or prismatic joints");
	const int ids[4] = { j1.bodyB, j2.bodyB, j1.bodyA, j2.bodyA }; // A, B, C, D
	for (int k = 0; k < 4; ++k)
		if (!w->bodies[ids[k]].dirty) pullBody(w, ids[k]);
	GearRec g;
	memset(&g, 0, sizeof(g));
	g.bodyC = ids[2];
	g.bodyD = ids[3];
	g.typeA = j1.type;
	g.typeB = j2.type;
	g.localAnchorC = j1.localAnchorA; g.localAnchorA = j1.localAnchorB; g.referenceAngleA = j1.referenceAngle;
	g.localAxisC = j1.type == B2D_JOINT_PRISMATIC ? j1.localAxisA : v2(0.0f, 0.0f);
	g.localAnchorD = j2.localAnchorA; g.localAnchorB = j2.localAnchorB; g.referenceAngleB = j2.referenceAngle;
	g.localAxisD = j2.type == B2D_JOINT_PRISMATIC ? j2.localAxisA : v2(0.0f, 0.0f);
	const float coordinateA = gearCoordinate(w, j1, ids[0], ids[2]);
	const float coordinateB = gearCoordinate(w, j2, ids[1], ids[3]);
	g.ratio = def->ratio;
	g.constant = coordinateA + g.ratio * coordinateB;
	JointRec j;
	memset(&j, 0, sizeof(j));
	j.type = B2D_JOINT_GEAR;
	j.bodyA = ids[0];
	j.bodyB = ids[1];
	j.enableLimit = (int)w->gears.size();
	j.collideConnected = def->collide_connected;
	w->gears.push_back(g);
	return addJoint(w, j);
}

// b2Body::SetAwake(true) on both bodies of a joint whose definition changed (b2RevoluteJoint.cpp:418-500)
static void wakeJointBodies(b2hip_world* w, const JointRec& j)
{
	setAwake(w, j.bodyA);
	setAwake(w, j.bodyB);
}

int b2hip_destroy_joint(b2hip_world* w, int joint)
{
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	if (int rcu = checkUsable(w, "b2hip_destroy_joint", true)) return rcu;
	JointRec& j = w->joints[joint];
	if (j.type == B2D_JOINT_DEAD) return setError(B2HIP_ERR_INVALID, "joint already destroyed");
	// (a gear joint must be destroyed before the joints it couples, as in the reference)
	wakeJointBodies(w, j);
	// contacts between the two bodies are filtered again when the joint kept them from colliding (b2World.cpp:833-845)
	if (j.collideConnected == 0) w->pendingFilter.push_back(std::make_pair(j.bodyA, j.bodyB));
	if (j.type == B2D_JOINT_MOUSE) w->nMouseJoints -= 1;
	j.type = B2D_JOINT_DEAD;
	w->jadjJoints = (size_t)-1; // per-body joint lists are rebuilt without it
	w->jointEdits.push_back(std::make_pair(joint, 3));
	return 0;
}

int b2hip_joint_set_motor(b2hip_world* w, int joint, int enable_motor, float motor_speed, float max_motor)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_motor", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_REVOLUTE && j.type != B2D_JOINT_PRISMATIC && j.type != B2D_JOINT_WHEEL)
		return setError(B2HIP_ERR_INVALID, "joint type has no motor");
	if ((enable_motor != 0) == (j.enableMotor != 0) && motor_speed == j.motorSpeed && max_motor == j.maxMotorTorque) return 0;
	wakeJointBodies(w, j);
	j.enableMotor = enable_motor != 0;
	j.motorSpeed = motor_speed;
	j.maxMotorTorque = max_motor;
	w->jointEdits.push_back(std::make_pair(joint, 0));
	return 0;
}

int b2hip_joint_set_offsets(b2hip_world* w, int joint, float linear_x, float linear_y, float angular)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_offsets", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_MOTOR) return setError(B2HIP_ERR_INVALID, "not a motor joint");
	if (linear_x == j.linearOffset.x && linear_y == j.linearOffset.y && angular == j.angularOffset) return 0;
	wakeJointBodies(w, j);
	j.linearOffset = v2(linear_x, linear_y);
	j.angularOffset = angular;
	w->jointEdits.push_back(std::make_pair(joint, 2)); // rewrite the anchor / offset members
	return 0;
}

int b2hip_joint_set_limits(b2hip_world* w, int joint, int enable_limit, float lower, float upper)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_limits", true)) return rcu;
	if (!w || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_REVOLUTE && j.type != B2D_JOINT_PRISMATIC) return setError(B2HIP_ERR_INVALID, "joint type has no limits");
	if (lower > upper) return setError(B2HIP_ERR_INVALID, "lower limit above upper limit");
	if ((enable_limit != 0) == (j.enableLimit != 0) && lower == j.lowerAngle && upper == j.upperAngle) return 0;
	wakeJointBodies(w, j);
	j.enableLimit = enable_limit != 0;
	j.lowerAngle = lower;
	j.upperAngle = upper;
	w->jointEdits.push_back(std::make_pair(joint, 1)); // the limit impulse restarts from zero
	return 0;
}

// b2RopeJoint::GetLimitState (b2RopeJoint.h:84) and the limit state of revolute / prismatic joints: the solver's, from the
// device record; 0 inactive, 1 at lower, 2 at upper, 3 equal (b2LimitState, b2Joint.h:58-64); negative: error
int b2hip_get_joint_limit_state(b2hip_world* w, int joint)
{
	if (int rcu = checkUsable(w, "b2hip_get_joint_limit_state", true)) return rcu;
	if (joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	DEVICE_GUARD(w);
	int state = w->joints[joint].limitState;
	if ((size_t)joint < w->upJoints && w->d_joints.p != nullptr)
	{
		HIP_TRY(hipMemcpyAsync(&state, (const char*)(w->d_joints.p + joint) + offsetof(JointRec, limitState), sizeof(int), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	return state;
}

int b2hip_get_joint_reaction(b2hip_world* w, int joint, float inv_dt, float out4[4])
{
	if (int rcu = checkUsable(w, "b2hip_get_joint_reaction", true)) return rcu;
	if (!w || !out4 || joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	DEVICE_GUARD(w);
	JointRec rec = w->joints[joint];
	GearRec gear;
	memset(&gear, 0, sizeof(gear));
	const bool isGear = rec.type == B2D_JOINT_GEAR;
	// (the solver's state lives in the device copy; a joint the device has not seen yet has done nothing)
	if ((size_t)joint < w->upJoints && w->d_joints.p != nullptr)
	{
		HIP_TRY(hipMemcpyAsync(&rec, w->d_joints.p + joint, sizeof(JointRec), hipMemcpyDeviceToHost, w->stream));
		if (isGear && (size_t)rec.enableLimit < w->upGears) HIP_TRY(hipMemcpyAsync(&gear, w->d_gears.p + rec.enableLimit, sizeof(GearRec), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	const JointReaction r = b2dJointReaction(&rec, isGear ? &gear : nullptr, inv_dt);
	out4[0] = r.force.x;
	out4[1] = r.force.y;
	out4[2] = r.torque;
	out4[3] = r.motor;
	return 0;
}

int b2hip_body_count(const b2hip_world* w)
{
	return w ? (int)w->bodies.size() : 0;
}

int b2hip_fixture_count(const b2hip_world* w)
{
	return w ? (int)w->fixtures.size() : 0;
}

int b2hip_get_mass_data(const b2hip_world* w, int body, b2hip_mass_data* out)
{
	if (!w || !out || body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad argument");
	const HostBody& b = w->bodies[body];
	out->mass = b.mass;
	// b2Body::GetInertia (b2Body.h:585-588)
	out->inertia = b.I + b.mass * (b.lcx * b.lcx + b.lcy * b.lcy);
	out->local_center[0] = b.lcx;
	out->local_center[1] = b.lcy;
	out->inv_mass = b.invMass;
	out->inv_inertia = b.invI;
	return 0;
}

int b2hip_apply_force(b2hip_world* w, int body, float fx, float fy, float torque, int wake)
{
	if (int rcu = checkUsable(w, "b2hip_apply_force", true)) return rcu;
	if (!w || body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad argument");
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return 0;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	// (a row that was already dirty - an edit from a callback of the last step - has not been through pullBody: the forces it
	// holds are the last step's, which the step cleared on the device, and its epoch must say that THIS force is new; else a
	// later pull of the same step - a PreSolve edit - takes the force for a stale one and drops it)
	if (w->def.auto_clear_forces && b.forceEpoch != w->stepEpoch) { b.fx = b.fy = b.torque = 0.0f; }
	b.forceEpoch = w->stepEpoch;
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE)
	{
		b.fx += fx;
		b.fy += fy;
		b.torque += torque;
	}
	return 0;
}

int b2hip_set_velocity(b2hip_world* w, int body, float vx, float vy, float omega)
{
	if (int rcu = checkUsable(w, "b2hip_set_velocity", true)) return rcu;
	if (!w || body < 0 || body >= (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad argument");
	if (w->bodies[body].type == B2HIP_STATIC_BODY) return 0;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (vx * vx + vy * vy > 0.0f || omega * omega > 0.0f)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	b.vx = vx;
	b.vy = vy;
	b.w = omega;
	return 0;
}

static int stepBeginImpl(b2hip_world* w, float dt, int velocity_iterations, int position_iterations)
{
	int rc = flushEdits(w);
	if (rc) return rc;
	StepParams& sp = w->sp;
	sp.dt = dt;
	sp.inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
	sp.dtRatio = w->inv_dt0 * dt;
	sp.velIters = velocity_iterations;
	sp.posIters = position_iterations;
	sp.warmStarting = w->def.warm_starting;
	sp.allowSleep = w->def.allow_sleep;
	sp.gravity = v2(w->def.gravity_x, w->def.gravity_y);
	if (w->kernelTiming > 1)
	{
		w->ktUsed = 0;
		w->ktKind = 0;
	}
	w->stepActive = true;
	w->stepSolves = w->stepComplete;
	w->dw.toiContinue = w->stepComplete ? 0 : 1;
	w->dw.toiEventCap = w->def.sub_stepping ? 1 : 0;
	// zero the per-step counters (keep nContacts / nMoves / cur)
	Counters zero;
	memset(&zero, 0, sizeof(zero));
	// (b2Profile::step: from the start of this kernel to the end of k_end_step, slots 14 and 13)
	w->dw.stampMask = 1u << 14;
	LAUNCH(w, k_step_begin, 1, 64, w->dw, w->gridBar.p);
	w->toiCountersFresh = true;
	rc = applyPendingFilters(w);
	if (rc) return rc;
	rc = applyEditOps(w, false); // (after k_step_begin: the end events of destroyed contacts belong to this step's list)
	if (rc) return rc;
	if (w->spatial)
	{
		if (listenerOn(w) || hasFilter(w) || w->def.sub_stepping) return setError(B2HIP_ERR_UNSUPPORTED, "contact listeners, filters and sub-stepping are not supported in a spatially sharded world");
		rc = spBeginStep(w);
		if (rc) return rc;
	}
	stampPhase(w, 0);
	// b2World.cpp:1628-1639: new fixtures -> find their contacts before colliding
	if (w->newFixture)
	{
		rc = findNewContacts(w, true);
		if (rc) return rc;
		w->newFixture = false;
	}
	stampPhase(w, 1);
	return 0;
}

// ---- life cycle and mutators ---------------------------------------------------------------------------------------------
static int checkBody(b2hip_world* w, int body, const char* what)
{
	if (int rcu = checkUsable(w, what, true)) return rcu;
	if (body < 0 || body >= (int)w->bodies.size() || w->bodies[body].dead) return setError(B2HIP_ERR_INVALID, std::string(what) + ": bad body id");
	return 0;
}

static int checkFixture(b2hip_world* w, int fixture, const char* what)
{
	if (int rcu = checkUsable(w, what, true)) return rcu;
	if (fixture < 0 || fixture >= (int)w->fixtures.size() || w->fixtures[fixture].dead) return setError(B2HIP_ERR_INVALID, std::string(what) + ": bad fixture id");
	return 0;
}

static void queueOp(b2hip_world* w, int kind, int id)
{
	w->editOps.push_back(make_int2(kind, id));
}

// b2Fixture::DestroyProxies (b2Fixture.cpp:143-157) + the host bookkeeping of a fixture that is gone
static void dropFixture(b2hip_world* w, int fixture)
{
	HostFixture& f = w->fixtures[fixture];
	freeProxyKey(w, f.proxyKey);
	f.dead = true;
	w->proxyEdits.push_back(fixture);
	w->proxyListsStale = true;
	// (a proxy created since the last step and not yet buffered on the device leaves the pending moves too: UnBufferMove)
	w->pendingMoves.erase(std::remove(w->pendingMoves.begin(), w->pendingMoves.end(), fixture), w->pendingMoves.end());
}

int b2hip_destroy_fixture(b2hip_world* w, int fixture)
{
	if (int rc = checkFixture(w, fixture, "b2hip_destroy_fixture")) return rc;
	const int body = w->fixtures[fixture].body;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	queueOp(w, EDIT_DESTROY_FIXTURE, fixture);
	b.fixtures.erase(std::remove(b.fixtures.begin(), b.fixtures.end(), fixture), b.fixtures.end());
	dropFixture(w, fixture);
	resetMassData(w, b);
	return B2HIP_OK;
}

int b2hip_destroy_body(b2hip_world* w, int body)
{
	if (int rc = checkBody(w, body, "b2hip_destroy_body")) return rc;
	// joints first, newest first (the body's joint list is newest first, b2World.cpp:697-710)
	for (int j = (int)w->joints.size() - 1; j >= 0; --j)
	{
		if (w->joints[j].type == B2D_JOINT_DEAD) continue;
		bool touches = w->joints[j].bodyA == body || w->joints[j].bodyB == body;
		if (w->joints[j].type == B2D_JOINT_GEAR)
		{
			const GearRec& g = w->gears[w->joints[j].enableLimit];
			touches = touches || g.bodyC == body || g.bodyD == body;
		}
		if (touches)
		{
			const int rc = b2hip_destroy_joint(w, j);
			if (rc) return rc;
		}
	}
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	queueOp(w, EDIT_DESTROY_BODY, body);
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k) dropFixture(w, b.fixtures[k]); // newest first
	b.fixtures.clear();
	if (b.worldIndex >= 0)
	{
		// b2RemoveAndSwapBack on m_nonStaticBodies (b2World.cpp:662-667)
		const int slot = b.worldIndex, last = w->nonStatic.back();
		w->nonStatic[(size_t)slot] = last;
		w->bodies[last].worldIndex = slot;
		w->nonStatic.pop_back();
		b.worldIndex = -1;
		w->orderDirty = true;
	}
	b.dead = 1;
	b.type = B2HIP_STATIC_BODY;
	b.flags &= ~(BF_ACTIVE | BF_AWAKE | BF_BULLET);
	b.vx = b.vy = b.w = 0.0f;
	b.fx = b.fy = b.torque = 0.0f;
	b.invMass = b.invI = 0.0f;
	return B2HIP_OK;
}

int b2hip_body_is_destroyed(const b2hip_world* w, int body)
{
	return w && body >= 0 && body < (int)w->bodies.size() && w->bodies[body].dead ? 1 : 0;
}

int b2hip_fixture_is_destroyed(const b2hip_world* w, int fixture)
{
	return w && fixture >= 0 && fixture < (int)w->fixtures.size() && w->fixtures[fixture].dead ? 1 : 0;
}

// The fat AABB a fixture's proxy has right now (the device owns it once the fixture is uploaded)
static int currentFat(b2hip_world* w, int fixture, float out4[4])
{
	if ((size_t)fixture >= w->upFixtures || std::find(w->fatEdits.begin(), w->fatEdits.end(), fixture) != w->fatEdits.end())
	{
		memcpy(out4, w->fixtures[fixture].fat, 16);
		return 0;
	}
	DEVICE_GUARD(w);
	HIP_TRY(hipStreamSynchronize(w->stream));
	HIP_TRY(hipMemcpy(out4, w->p_fat.p + fixture, 16, hipMemcpyDeviceToHost));
	return 0;
}

int b2hip_set_transform(b2hip_world* w, int body, float x, float y, float angle)
{
	if (int rc = checkBody(w, body, "b2hip_set_transform")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	b.qs = sinf(angle);
	b.qc = cosf(angle);
	b.px = x;
	b.py = y;
	const V2 c = b2dMulXV(hostXf(b), v2(b.lcx, b.lcy));
	b.cx = b.c0x = c.x;
	b.cy = b.c0y = c.y;
	b.a = b.a0 = angle;
	b.resetSweep = 1;
	// b2Fixture::Synchronize(broadPhase, xf, xf) for every fixture, newest first -> b2DynamicTree::MoveProxy with zero displacement
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
	{
		const int id = b.fixtures[k];
		HostFixture& f = w->fixtures[id];
		float fat[4];
		if (int rc = currentFat(w, id, fat)) return rc;
		const AABB aabb = b2dShapeAABB(&w->shapes[f.shape], hostXf(b));
		if (fat[0] <= aabb.lo.x && fat[1] <= aabb.lo.y && aabb.hi.x <= fat[2] && aabb.hi.y <= fat[3])
		{
			memcpy(f.fat, fat, 16);
			continue;
		}
		f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
		f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
		f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
		f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
		w->proxyEdits.push_back(id);
		w->fatEdits.push_back(id);
		if ((size_t)id < w->upFixtures || std::find(w->pendingMoves.begin(), w->pendingMoves.end(), id) == w->pendingMoves.end()) w->pendingMoves.push_back(id);
		w->newFixture = w->newFixture; // (moves alone do not ask for the top-of-step pair update: the end-of-step one takes them)
	}
	return B2HIP_OK;
}

int b2hip_set_active(b2hip_world* w, int body, int active)
{
	if (int rc = checkBody(w, body, "b2hip_set_active")) return rc;
	if (((w->bodies[body].flags & BF_ACTIVE) != 0) == (active != 0)) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (active)
	{
		b.flags |= BF_ACTIVE;
		// b2Fixture::CreateProxies for every fixture, newest first: fat AABB at the body's transform, a fresh proxy id, a buffered move
		for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
		{
			const int id = b.fixtures[k];
			HostFixture& f = w->fixtures[id];
			const AABB aabb = b2dShapeAABB(&w->shapes[f.shape], hostXf(b));
			f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
			f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
			f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
			f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
			f.proxyKey = allocProxyKey(w);
			if (f.proxyKey < 0) return setError(B2HIP_ERR_UNSUPPORTED, "proxy id reuse after the broad-phase tree was emptied is not modelled");
			f.noProxy = false;
			w->proxyEdits.push_back(id);
			w->fatEdits.push_back(id);
			w->pendingMoves.push_back(id);
		}
		w->proxyListsStale = true;
		return B2HIP_OK;
	}
	b.flags &= ~BF_ACTIVE;
	// b2Fixture::DestroyProxies, newest fixture first, then the body's contacts in its contact-list order
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
	{
		const int id = b.fixtures[k];
		HostFixture& f = w->fixtures[id];
		if (f.noProxy) continue;
		freeProxyKey(w, f.proxyKey);
		f.proxyKey = -1;
		f.noProxy = true;
		w->proxyEdits.push_back(id);
		w->pendingMoves.erase(std::remove(w->pendingMoves.begin(), w->pendingMoves.end(), id), w->pendingMoves.end());
	}
	w->proxyListsStale = true;
	queueOp(w, EDIT_DESTROY_BODY, body);
	return B2HIP_OK;
}

int b2hip_set_type(b2hip_world* w, int body, int type)
{
	if (int rc = checkBody(w, body, "b2hip_set_type")) return rc;
	if (type < B2HIP_STATIC_BODY || type > B2HIP_DYNAMIC_BODY) return setError(B2HIP_ERR_INVALID, "b2hip_set_type: bad body type");
	if (w->bodies[body].type == type) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (b.type == B2HIP_STATIC_BODY)
	{
		// out of m_staticBodies, to the end of m_nonStaticBodies (b2Body.cpp:131-140)
		b.worldIndex = (int)w->nonStatic.size();
		w->nonStatic.push_back(body);
		w->orderDirty = true;
	}
	b.type = type;
	resetMassData(w, b);
	b.resetSweep = 1; // (the mass data moved the sweep origin with the centre)
	if (type == B2HIP_STATIC_BODY)
	{
		b.vx = b.vy = b.w = 0.0f;
		b.a0 = b.a;
		b.c0x = b.cx;
		b.c0y = b.cy;
		b.resetSweep = 1;
		// b2Body::SynchronizeFixtures with xf1 == xf (the sweep origin was just reset): MoveProxy with zero displacement
		for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
		{
			const int id = b.fixtures[k];
			HostFixture& f = w->fixtures[id];
			if (f.noProxy) continue;
			float fat[4];
			if (int rc = currentFat(w, id, fat)) return rc;
			const AABB aabb = b2dShapeAABB(&w->shapes[f.shape], hostXf(b));
			if (fat[0] <= aabb.lo.x && fat[1] <= aabb.lo.y && aabb.hi.x <= fat[2] && aabb.hi.y <= fat[3])
			{
				memcpy(f.fat, fat, 16);
				continue;
			}
			f.fat[0] = aabb.lo.x - B2D_AABB_EXTENSION;
			f.fat[1] = aabb.lo.y - B2D_AABB_EXTENSION;
			f.fat[2] = aabb.hi.x + B2D_AABB_EXTENSION;
			f.fat[3] = aabb.hi.y + B2D_AABB_EXTENSION;
			w->proxyEdits.push_back(id);
			w->fatEdits.push_back(id);
			w->pendingMoves.push_back(id);
		}
		// b2RemoveAndSwapBack on m_nonStaticBodies (b2Body.cpp:154-160)
		const int slot = b.worldIndex, last = w->nonStatic.back();
		w->nonStatic[(size_t)slot] = last;
		w->bodies[(size_t)last].worldIndex = slot;
		w->nonStatic.pop_back();
		b.worldIndex = -1;
		w->orderDirty = true;
	}
	b.flags |= BF_AWAKE;
	b.sleepTime = 0.0f;
	b.fx = b.fy = b.torque = 0.0f;
	// every contact of the body goes, in its contact-list order; TouchProxy on every proxy, newest fixture first
	queueOp(w, EDIT_DESTROY_BODY, body);
	for (int k = (int)b.fixtures.size() - 1; k >= 0; --k)
	{
		const int id = b.fixtures[k];
		if (!w->fixtures[id].noProxy) w->pendingMoves.push_back(id);
	}
	return B2HIP_OK;
}

int b2hip_set_awake(b2hip_world* w, int body, int awake)
{
	if (int rc = checkBody(w, body, "b2hip_set_awake")) return rc;
	if (awake)
	{
		setAwake(w, body);
		return B2HIP_OK;
	}
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	b.flags &= ~BF_AWAKE;
	b.sleepTime = 0.0f;
	b.vx = b.vy = b.w = 0.0f;
	b.fx = b.fy = b.torque = 0.0f;
	return B2HIP_OK;
}

int b2hip_set_bullet(b2hip_world* w, int body, int bullet)
{
	if (int rc = checkBody(w, body, "b2hip_set_bullet")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	const bool was = (b.flags & BF_BULLET) != 0;
	if (bullet) b.flags |= BF_BULLET; else b.flags &= ~BF_BULLET;
	if (was != (bullet != 0)) queueOp(w, EDIT_RECALC_BODY, body);
	return B2HIP_OK;
}

int b2hip_apply_linear_impulse(b2hip_world* w, int body, float ix, float iy, float px, float py, int wake)
{
	if (int rc = checkBody(w, body, "b2hip_apply_linear_impulse")) return rc;
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE)
	{
		// b2Body.h:915-921
		const float sx = b.invMass * ix, sy = b.invMass * iy;
		b.vx += sx;
		b.vy += sy;
		b.w += b.invI * ((px - b.cx) * iy - (py - b.cy) * ix);
	}
	return B2HIP_OK;
}

int b2hip_apply_linear_impulse_to_center(b2hip_world* w, int body, float ix, float iy, int wake)
{
	if (int rc = checkBody(w, body, "b2hip_apply_linear_impulse_to_center")) return rc;
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE)
	{
		const float sx = b.invMass * ix, sy = b.invMass * iy; // b2Body.h:923-942
		b.vx += sx;
		b.vy += sy;
	}
	return B2HIP_OK;
}

int b2hip_apply_angular_impulse(b2hip_world* w, int body, float impulse, int wake)
{
	if (int rc = checkBody(w, body, "b2hip_apply_angular_impulse")) return rc;
	if (w->bodies[body].type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (wake && (b.flags & BF_AWAKE) == 0)
	{
		b.flags |= BF_AWAKE;
		b.sleepTime = 0.0f;
	}
	if (b.flags & BF_AWAKE) b.w += b.invI * impulse;
	return B2HIP_OK;
}

int b2hip_fixture_set_sensor(b2hip_world* w, int fixture, int is_sensor)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_sensor")) return rc;
	HostFixture& f = w->fixtures[fixture];
	if (f.isSensor == (is_sensor != 0)) return B2HIP_OK;
	setAwake(w, f.body);
	f.isSensor = is_sensor != 0;
	w->proxyEdits.push_back(fixture);
	queueOp(w, EDIT_SENSOR_FIXTURE, fixture);
	queueOp(w, EDIT_RECALC_FIXTURE, fixture);
	return B2HIP_OK;
}

int b2hip_fixture_set_thick(b2hip_world* w, int fixture, int thick_shape)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_thick")) return rc;
	HostFixture& f = w->fixtures[fixture];
	if (f.thick == (thick_shape != 0)) return B2HIP_OK;
	f.thick = thick_shape != 0;
	w->proxyEdits.push_back(fixture);
	queueOp(w, EDIT_RECALC_FIXTURE, fixture);
	return B2HIP_OK;
}

int b2hip_fixture_refilter(b2hip_world* w, int fixture)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_refilter")) return rc;
	queueOp(w, EDIT_REFILTER_FIXTURE, fixture);
	w->refilterPending = true;
	// TouchProxy (b2BroadPhase.cpp:70-73): the proxy is buffered as moved so that new pairs can form
	if (w->bodies[w->fixtures[fixture].body].flags & BF_ACTIVE) w->pendingMoves.push_back(fixture);
	return B2HIP_OK;
}

int b2hip_fixture_set_filter(b2hip_world* w, int fixture, uint16_t category_bits, uint16_t mask_bits, int16_t group_index)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_filter")) return rc;
	HostFixture& f = w->fixtures[fixture];
	f.categoryBits = category_bits;
	f.maskBits = mask_bits;
	f.groupIndex = group_index;
	w->proxyEdits.push_back(fixture);
	return b2hip_fixture_refilter(w, fixture);
}

// b2Fixture::SetDensity / SetFriction / SetRestitution (b2Fixture.h:306-334): plain values - the density is read by the next
// ResetMassData, friction and restitution by the contacts created from now on (existing contacts keep their mixture)
int b2hip_fixture_set_material(b2hip_world* w, int fixture, float density, float friction, float restitution)
{
	if (int rc = checkFixture(w, fixture, "b2hip_fixture_set_material")) return rc;
	HostFixture& f = w->fixtures[fixture];
	f.density = density;
	f.friction = friction;
	f.restitution = restitution;
	w->proxyEdits.push_back(fixture);
	return B2HIP_OK;
}

// b2Body::SetLinearDamping / SetAngularDamping / SetGravityScale (b2Body.h:620-648): read by the next Solve
int b2hip_set_body_damping(b2hip_world* w, int body, float linear_damping, float angular_damping, float gravity_scale)
{
	if (int rc = checkBody(w, body, "b2hip_set_body_damping")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	b.linearDamping = linear_damping;
	b.angularDamping = angular_damping;
	b.gravityScale = gravity_scale;
	return B2HIP_OK;
}

// b2Body::SetFixedRotation (b2Body.cpp:546-565): the flag, no spin, mass data again
int b2hip_set_fixed_rotation(b2hip_world* w, int body, int flag)
{
	if (int rc = checkBody(w, body, "b2hip_set_fixed_rotation")) return rc;
	HostBody& probe = w->bodies[body];
	if (((probe.flags & BF_FIXEDROT) != 0) == (flag != 0)) return B2HIP_OK;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (flag) b.flags |= BF_FIXEDROT; else b.flags &= ~BF_FIXEDROT;
	b.w = 0.0f;
	resetMassData(w, b);
	b.resetSweep = 1;
	return B2HIP_OK;
}

// b2Body::SetSleepingAllowed (b2Body.h:674-688): a body that may not sleep is woken
int b2hip_set_sleeping_allowed(b2hip_world* w, int body, int flag)
{
	if (int rc = checkBody(w, body, "b2hip_set_sleeping_allowed")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (flag) b.flags |= BF_AUTOSLEEP;
	else
	{
		b.flags &= ~BF_AUTOSLEEP;
		b.flags |= BF_AWAKE; // SetAwake(true) (b2Body.h:690-718): the sleep timer restarts whether or not the body slept
		b.sleepTime = 0.0f;
	}
	return B2HIP_OK;
}

// b2Body::SetMassData (b2Body.cpp:387-424); mass_data == NULL: b2Body::ResetMassData (b2Body.cpp:310-385)
int b2hip_set_mass_data(b2hip_world* w, int body, const b2hip_mass_data* md)
{
	if (int rc = checkBody(w, body, "b2hip_set_mass_data")) return rc;
	markDirty(w, body);
	HostBody& b = w->bodies[body];
	if (md == nullptr)
	{
		resetMassData(w, b);
		b.resetSweep = 1;
		return B2HIP_OK;
	}
	if (b.type != B2HIP_DYNAMIC_BODY) return B2HIP_OK;
	b.invMass = 0.0f;
	b.I = 0.0f;
	b.invI = 0.0f;
	b.mass = md->mass;
	if (b.mass <= 0.0f) b.mass = 1.0f;
	b.invMass = 1.0f / b.mass;
	const V2 center = v2(md->local_center[0], md->local_center[1]);
	if (md->inertia > 0.0f && (b.flags & BF_FIXEDROT) == 0)
	{
		b.I = md->inertia - b.mass * b2dDot(center, center);
		b.invI = 1.0f / b.I;
	}
	const V2 oldCenter = v2(b.cx, b.cy);
	b.lcx = center.x;
	b.lcy = center.y;
	const V2 c = b2dMulXV(hostXf(b), center);
	b.c0x = b.cx = c.x;
	b.c0y = b.cy = c.y;
	const V2 dv = b2dCrossSV(b.w, c - oldCenter);
	b.vx += dv.x;
	b.vy += dv.y;
	b.resetSweep = 1;
	return B2HIP_OK;
}

// The scalar setters of the joint classes: plain assignments in the reference (b2DistanceJoint.h:117, b2RopeJoint.h:80,
// b2FrictionJoint.cpp:206-228, b2MotorJoint.cpp:222-251, b2MouseJoint.cpp:48-76, b2GearJoint.cpp:402-406)
int b2hip_joint_set_param(b2hip_world* w, int joint, int param, float value)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_param", true)) return rcu;
	if (joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	const int t = j.type;
	int kind = 0;
	if (param == B2HIP_JOINT_LENGTH && (t == B2D_JOINT_DISTANCE || t == B2D_JOINT_ROPE)) { j.length = value; kind = 2; }
	else if (param == B2HIP_JOINT_MAX_FORCE && (t == B2D_JOINT_FRICTION || t == B2D_JOINT_MOTOR || t == B2D_JOINT_MOUSE)) j.maxForce = value;
	else if (param == B2HIP_JOINT_MAX_TORQUE && (t == B2D_JOINT_FRICTION || t == B2D_JOINT_MOTOR)) j.maxTorque = value;
	else if (param == B2HIP_JOINT_RATIO && t == B2D_JOINT_GEAR)
	{
		// (a gear's definition lives in its own record, GearRec; JointRec::enableLimit is its index there)
		const int gi = j.enableLimit;
		if (gi < 0 || gi >= (int)w->gears.size()) return setError(B2HIP_ERR_INVALID, "gear record missing");
		w->gears[(size_t)gi].ratio = value;
		if ((size_t)gi < w->upGears)
		{
			DEVICE_GUARD(w);
			HIP_TRY(hipMemcpy((char*)(w->d_gears.p + gi) + offsetof(GearRec, ratio), &value, sizeof(float), hipMemcpyHostToDevice));
		}
		return B2HIP_OK;
	}
	else if (param == B2HIP_JOINT_CORRECTION_FACTOR && t == B2D_JOINT_MOTOR) j.correctionFactor = value;
	else return setError(B2HIP_ERR_INVALID, "b2hip_joint_set_param: the joint's type has no such parameter");
	w->jointEdits.push_back(std::make_pair(joint, kind));
	return B2HIP_OK;
}

// b2World::ShiftOrigin (b2World.cpp:1862-1887)
int b2hip_shift_origin(b2hip_world* w, float x, float y)
{
	if (int rcu = checkUsable(w, "b2hip_shift_origin", true)) return rcu;
	if (w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_shift_origin inside a step");
	DEVICE_GUARD(w);
	// every edit made so far goes to the device first: from here on the device state is the one that is shifted
	int rc = flushEdits(w);
	if (rc) return rc;
	rc = applyEditOps(w, true);
	if (rc) return rc;
	const int n = std::max(std::max(w->dw.nBodies, w->dw.nProxies), std::max(w->dw.nJoints, 1));
	LAUNCH(w, k_shift_origin, gridFor(n), 256, w->dw, x, y);
	// the host's copies: joint records (uploaded again when a setter edits them), fat AABBs, and the body rows - read back
	for (size_t j = 0; j < w->joints.size(); ++j)
	{
		JointRec& jn = w->joints[j];
		if (jn.type == B2D_JOINT_MOUSE) { jn.targetA.x -= x; jn.targetA.y -= y; }
		else if (jn.type == B2D_JOINT_PULLEY) { jn.groundAnchorA.x -= x; jn.groundAnchorA.y -= y; jn.s1 -= x; jn.s2 -= y; }
	}
	for (size_t f = 0; f < w->fixtures.size(); ++f)
	{
		w->fixtures[f].fat[0] -= x; w->fixtures[f].fat[1] -= y;
		w->fixtures[f].fat[2] -= x; w->fixtures[f].fat[3] -= y;
	}
	rc = downloadState(w, 0);
	if (rc) return rc;
	w->stateCount = w->bodies.size();
	++w->mirrorEpoch;
	return B2HIP_OK;
}

int b2hip_joint_set_spring(b2hip_world* w, int joint, float frequency_hz, float damping_ratio)
{
	if (int rcu = checkUsable(w, "b2hip_joint_set_spring", true)) return rcu;
	if (joint < 0 || joint >= (int)w->joints.size()) return setError(B2HIP_ERR_INVALID, "bad joint id");
	JointRec& j = w->joints[joint];
	if (j.type != B2D_JOINT_WHEEL && j.type != B2D_JOINT_DISTANCE && j.type != B2D_JOINT_WELD && j.type != B2D_JOINT_MOUSE)
		return setError(B2HIP_ERR_INVALID, "joint type has no spring");
	j.frequencyHz = frequency_hz;
	j.dampingRatio = damping_ratio;
	w->jointEdits.push_back(std::make_pair(joint, 0));
	return B2HIP_OK;
}

int b2hip_step_begin(b2hip_world* w, float dt, int velocity_iterations, int position_iterations)
{
	if (int rc = checkUsable(w, "b2hip_step_begin", false)) return rc;
	if (w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_step_begin inside a step (finish it with b2hip_step_end)");
	DEVICE_GUARD(w);
	return stepFailed(w, stepBeginImpl(w, dt, velocity_iterations, position_iterations));
}

static bool keyLess(const std::pair<unsigned long long, int>& a, const std::pair<unsigned long long, int>& b) { return a < b; }

static void toManifold(b2hip_manifold* m, float4 m0, float4 m1, float4 imp, int4 m3)
{
	m->type = m3.z;
	m->point_count = m3.w;
	m->local_normal[0] = m0.x; m->local_normal[1] = m0.y;
	m->local_point[0] = m0.z; m->local_point[1] = m0.w;
	m->point_local[0][0] = m1.x; m->point_local[0][1] = m1.y;
	m->point_local[1][0] = m1.z; m->point_local[1][1] = m1.w;
	m->normal_impulse[0] = imp.x; m->tangent_impulse[0] = imp.y;
	m->normal_impulse[1] = imp.z; m->tangent_impulse[1] = imp.w;
	m->id_key[0] = (uint32_t)m3.x;
	m->id_key[1] = (uint32_t)m3.y;
}

static int collideImpl(b2hip_world* w)
{
	int rc = 0;
	if (hasFilter(w) && w->refilterPending)
	{
		// b2ContactManager::Collide's re-filter (:195-203) with a user filter: the flagged contacts are shown to it first
		LAUNCH(w, k_filter_list, gridFor(w->dw.capContacts), 256, w->dw);
		rc = readState(w);
		if (rc) return rc;
		const int n = std::min(w->h_dstate->c.nFilterList, w->dw.capContacts);
		if (n > 0)
		{
			std::vector<int> list(n), refused;
			HIP_TRY(hipMemcpy(list.data(), w->filterList.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
			std::sort(list.begin(), list.end());
			const int cur = w->h_dstate->cur;
			std::vector<int> asked, verdict;
			for (int k = 0; k < n; ++k)
			{
				int4 ids;
				HIP_TRY(hipMemcpy(&ids, w->c_ids[cur].p + list[k], sizeof(int4), hipMemcpyDeviceToHost));
				asked.push_back(ids.x);
				asked.push_back(ids.y);
			}
			askFilter(w, asked, verdict);
			for (int k = 0; k < n; ++k) if (!verdict[k]) refused.push_back(list[k]);
			rc = applyHostList(w, k_filter_reject, refused);
			if (rc) return rc;
		}
	}
	w->refilterPending = false;
	rc = phaseCollide(w);
	if (rc) return rc;
	if (hasPreSolve(w))
	{
		// b2ContactListener::PreSolve: one record per touching, non-sensor contact this Collide updated; delivered in
		// proxy-id-pair order (b2ContactManager.cpp:431-434); a zero return disables the contact for this step
		rc = readState(w);
		if (rc) return rc;
		const int n = std::min(w->h_dstate->c.nPreSolve, w->dw.capContacts);
		if (n > 0)
		{
			std::vector<PreSolveRec> recs(n);
			HIP_TRY(hipMemcpy((void*)recs.data(), w->preRecs.p, (size_t)n * sizeof(PreSolveRec), hipMemcpyDeviceToHost));
			std::vector<std::pair<unsigned long long, int> > order(n);
			for (int i = 0; i < n; ++i) order[i] = std::make_pair(recs[i].key, i);
			std::sort(order.begin(), order.end(), keyLess);
			std::vector<int> disabled, materials;
			w->callbackWindow = true;
			std::vector<b2hip_pre_solve_record> batch((size_t)n);
			for (int k = 0; k < n; ++k)
			{
				const PreSolveRec& r = recs[order[k].second];
				b2hip_pre_solve_record& b = batch[k];
				b.contact_index = r.info.x;
				b.fixture_a = r.info.y;
				b.fixture_b = r.info.z;
				b.enabled = 1;
				toManifold(&b.old_manifold, r.o0, r.o1, r.oimp, r.o3);
				toManifold(&b.manifold, r.n0, r.n1, r.nimp, r.n3);
				b.material.friction = r.mat.x;
				b.material.restitution = r.mat.y;
				b.material.tangent_speed = r.mat.z;
			}
			if (w->preSolveBatchFn) w->preSolveBatchFn(w->preSolveUser, n, batch.data());
			else for (int k = 0; k < n; ++k)
			{
				b2hip_pre_solve_record& b = batch[k];
				b.enabled = w->preSolveFn(w->preSolveUser, b.contact_index, b.fixture_a, b.fixture_b, &b.old_manifold, &b.manifold, &b.material) ? 1 : 0;
			}
			for (int k = 0; k < n; ++k)
			{
				const PreSolveRec& r = recs[order[k].second];
				const b2hip_contact_material& mat = batch[k].material;
				if (!batch[k].enabled) disabled.push_back(r.info.x);
				if (memcmp(&mat.friction, &r.mat.x, 4) != 0 || memcmp(&mat.restitution, &r.mat.y, 4) != 0 || memcmp(&mat.tangent_speed, &r.mat.z, 4) != 0)
				{
					int bits[3];
					memcpy(bits, &mat, sizeof(bits));
					materials.push_back(r.info.x);
					materials.insert(materials.end(), bits, bits + 3);
				}
			}
			w->callbackWindow = false;
			rc = applyHostList(w, k_presolve_disable, disabled);
			if (rc) return rc;
			if (!materials.empty())
			{
				HIP_TRY(hipMemcpyAsync(w->hostList.p, materials.data(), materials.size() * sizeof(int), hipMemcpyHostToDevice, w->stream));
				LAUNCH(w, k_presolve_material, gridFor(materials.size() / 4), 256, w->dw, (const int*)w->hostList.p, (int)(materials.size() / 4));
				HIP_TRY(hipStreamSynchronize(w->stream));
			}
		}
		// Edits made from inside PreSolve take effect at once, as in the reference, whose deferred callbacks run between
		// Collide and Solve (b2ContactManager::FinishCollide, b2ContactManager.cpp:387-441; Testbed/Tests/TunnelingTest.h
		// switches sensors, thick shapes and bullets there and expects this step's solvers to see it)
		if (!w->dirtyList.empty() || !w->editOps.empty() || !w->proxyEdits.empty() || !w->pendingMoves.empty())
		{
			rc = flushEdits(w);
			if (rc) return rc;
			rc = applyEditOps(w, false);
			if (rc) return rc;
		}
	}
	// the begin / end events of THIS phase are listed now (k_contact_events compares touching with what the host was told and
	// flips CF_REPORTED): what the TOI sub-steps change later in the step is logged by the sub-steps themselves, in order
	if (w->eventsOn && w->def.continuous && w->sp.dt > 0.0f) LAUNCH(w, k_contact_events, gridFor(w->dw.capContacts), 256, w->dw);
	stampPhase(w, 2);
	return 0;
}

int b2hip_collide(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_collide", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_collide outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, collideImpl(w));
}

static int shardExchangeOnStream(b2hip_world* w);

static int solveImpl(b2hip_world* w)
{
	if (w->sp.dt > 0.0f && w->stepSolves) // (b2World.cpp:1668: m_stepComplete && step.dt > 0)
	{
		int rc = phaseSolve(w);
		if (rc) return rc;
		// a connected sharded world (b2hip_shard_connect): the islands the other ranks solved arrive here, on the stream
		if (w->shardComm != nullptr && !w->spatial && (w->dw.shardCount > 1 || w->shardLoopback))
		{
			rc = shardExchangeOnStream(w);
			if (rc) return rc;
		}
	}
	else
	{
		// (a call that continues an open step: no island build to apply the wake-ups Collide asked for)
		if (w->sp.dt > 0.0f) LAUNCH(w, k_wake_apply, gridFor(w->dw.nBodies), 256, w->dw);
		for (int k = 4; k <= 8; ++k) stampPhase(w, k);
	}
	if (w->postSolveOn && w->sp.dt > 0.0f && w->stepSolves) LAUNCH(w, k_postsolve_gather, gridFor(w->dw.capContacts), 256, w->dw);
	stampPhase(w, 3);
	return 0;
}

int b2hip_solve(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_solve", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_solve outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, solveImpl(w));
}

static int syncFixturesImpl(b2hip_world* w)
{
	if (w->sp.dt > 0.0f && w->stepSolves)
	{
		int rc = phaseSyncFixtures(w);
		if (rc) return rc;
		// E1: what the other ranks' bodies did in Solve, and the fat AABBs their SynchronizeFixtures moved
		if (w->spatial) { rc = spExchangeState(w, 0); if (rc) return rc; }
		rc = forkEarlyRows(w);
		if (rc) return rc;
	}
	stampPhase(w, 9);
	return 0;
}

int b2hip_sync_fixtures(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_sync_fixtures", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_sync_fixtures outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, syncFixturesImpl(w));
}

static int findNewContactsImpl(b2hip_world* w)
{
	if (w->sp.dt > 0.0f && w->stepSolves)
	{
		int rc = findNewContactsGraph(w);
		if (rc) return rc;
		rc = startEarlyRows(w); // (if the pair update has not sent them off itself)
		if (rc) return rc;
	}
	stampPhase(w, 10);
	return 0;
}

int b2hip_find_new_contacts(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_find_new_contacts", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_find_new_contacts outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, findNewContactsImpl(w));
}

static int solveToiImpl(b2hip_world* w)
{
	w->toiRan = false;
	w->toiChains = false;
	w->toiSpeculative = false;
	w->toiSnapshotTaken = false;
	w->toiVerdicts.clear();
	w->dw.nToiVerdict = 0;
	w->last.nToiList = w->last.nToiCalls = w->last.nToiEvents = 0;
	if (w->def.continuous && w->sp.dt > 0.0f)
	{
		int rc = phaseToi(w);
		if (rc) return rc;
		// E4: the other ranks' TOI events (after this rank's phase has settled: fallbacks run here, not at the step's end)
		for (int attempt = 0; w->spatial; ++attempt)
		{
			rc = spAfterToi(w);
			if (rc <= 0) { if (rc) return rc; break; }
			// (1: an event reached over an ownership boundary - the phase was taken back and the owners merged: once more)
			if (attempt == 8) return setError(B2HIP_ERR_INVALID, "the TOI phase of a spatially sharded world keeps reaching over ownership boundaries");
			w->toiRan = false; w->toiChains = false; w->toiSpeculative = false; w->toiSnapshotTaken = false; w->toiCountersFresh = false;
			rc = phaseToi(w);
			if (rc) return rc;
		}
	}
	stampPhase(w, 12);
	w->toiEventValid = true;
	return 0;
}

int b2hip_solve_toi(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_solve_toi", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_solve_toi outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, solveToiImpl(w));
}

static int uploadToiVerdicts(b2hip_world* w);

// A contact created inside a TOI sub-step did not fit the array (whatever path ran last, fallbacks included): never a
// silent drop. With the snapshot of this step's TOI phase at hand the phase is undone, the array doubled and the phase
// run again; without one (serial-only mode) it is an error.
static int settleToiOverflow(b2hip_world* w)
{
	int rc = 0;
	for (int attempt = 0; (w->h_dstate->c.overflow & 1) != 0; ++attempt)
	{
		if (!w->toiSnapshotTaken || attempt == 3) return setError(B2HIP_ERR_CAPACITY, "contact array full during a TOI sub-step");
		// (edits can only be pending here if a PreSolve called from a sub-step made them: toiPreSolveRounds)
		if (!w->dirtyList.empty() || !w->editOps.empty() || !w->proxyEdits.empty() || !w->pendingMoves.empty())
			return setError(B2HIP_ERR_CAPACITY, "contact array full during a TOI sub-step whose PreSolve edited the world");
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		rc = ensureCapacity(w, 2 * (size_t)w->dw.capContacts);
		if (rc) return rc;
		rc = uploadToiVerdicts(w);
		if (rc) return rc;
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.overflow, 0, sizeof(int), w->stream));
		w->toiChains = false;
		w->toiSpeculative = false;
		rc = phaseToiSync(w);
		if (rc) return rc;
		if (w->toiChains)
		{
			// (the parallel paths report through toiUnsafe: take their serial fallback here as well)
			rc = downloadState(w, -1);
			if (rc) return rc;
			if (w->h_dstate->c.toiUnsafe != 0)
			{
				LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
				rc = toiSerial(w);
				if (rc) return rc;
				w->toiFallbacks += 1;
			}
		}
		rc = downloadState(w, -1);
		if (rc) return rc;
	}
	return 0;
}

// The answers collected so far go to the device before the phase runs again (DW::toiVerdict, DW::nToiVerdict).
static int uploadToiVerdicts(b2hip_world* w)
{
	const size_t n = std::min(w->toiVerdicts.size(), w->toiVerdict.cap);
	w->dw.nToiVerdict = w->dw.toiVerdict != nullptr ? (int)n : 0;
	if (w->dw.nToiVerdict > 0) HIP_TRY(hipMemcpyAsync((void*)w->toiVerdict.p, w->toiVerdicts.data(), n * sizeof(int4), hipMemcpyHostToDevice, w->stream));
	return 0;
}

static void toiCallbackFromLog(const ToiLogRec& r, b2hip_toi_callback* cb)
{
	memset(cb, 0, sizeof(*cb));
	cb->kind = r.info.x;
	cb->contact_index = r.info.y;
	cb->fixture_a = r.info.z;
	cb->fixture_b = r.info.w;
	toManifold(&cb->old_manifold, r.o0, r.o1, r.oimp, r.o3);
	toManifold(&cb->manifold, r.n0, r.n1, r.nimp, r.n3);
	cb->material.friction = r.mat.x;
	cb->material.restitution = r.mat.y;
	cb->material.tangent_speed = r.mat.z;
}

// b2ContactListener::PreSolve from INSIDE the TOI sub-steps (b2World.cpp:866,946 -> b2Contact::Update -> b2Contact.cpp:283-297).
// The reference calls it in the middle of its event loop, and what it does to the contact changes that sub-step: a contact
// switched off keeps the sweeps of its bodies and stays out of the sub-step's island (b2World.cpp:873-881, 948-954), an
// edited material is what the sub-step's solver reads. The event loop here is one kernel, so the phase is run to its end,
// its log (DW::toiLog: the Updates in the reference's call order) read, and PreSolve called for the logged Updates in
// order - each exactly once. An answer that changes nothing needs nothing: the contact stays on (or off, where the sub-step
// assumed the listener's last answer for this contact: CF_PRESOLVE_OFF, toiPreSolveOutcome) and its material as it was. The
// first one that does makes everything after it void: the phase goes back to its snapshot and runs again with the
// answers so far on the device (the loop applies them at the same log slots - it is deterministic, so the log repeats
// itself up to there), and the calls go on behind the slot that was answered. One extra run of the phase per changing answer.
static int toiPreSolveRounds(b2hip_world* w, std::vector<ToiLogRec>& recs)
{
	int asked = 0; // log slots whose PreSolve has been called
	for (int round = 0;; ++round)
	{
		const int n = std::min(w->h_dstate->c.nToiLog, w->dw.capToiLog);
		recs.resize((size_t)std::max(n, 0));
		if (n > 0) HIP_TRY(hipMemcpy((void*)recs.data(), w->toiLog.p, (size_t)n * sizeof(ToiLogRec), hipMemcpyDeviceToHost));
		if (!hasPreSolve(w) || w->dw.toiVerdict == nullptr) return 0;
		// (the log itself was cut short: no listener call from a truncated log - the step fails with the capacity error below)
		if (w->h_dstate->c.toiOverflow & 64) return 0;
		// The callbacks below may edit bodies (callbackWindow). h_state holds the state AFTER this step's phases by now, but the
		// mirror's epoch is only advanced at the very end of the step (refreshMirror): a body the host had touched before the
		// step (a force applied every frame) would not be pulled again and the edit would land on - and later upload - its
		// pre-step row. The read-back that just happened is the mirror from here on.
		if (n > asked) refreshMirror(w);
		bool again = false;
		for (int k = asked; k < n && !again; ++k)
		{
			const ToiLogRec& r = recs[(size_t)k];
			int4 v = make_int4(0, 0, 0, 0);
			if (r.info.x & 4)
			{
				b2hip_toi_callback cb;
				toiCallbackFromLog(r, &cb);
				b2hip_pre_solve_record rec;
				rec.contact_index = cb.contact_index;
				rec.fixture_a = cb.fixture_a;
				rec.fixture_b = cb.fixture_b;
				rec.enabled = 1;
				rec.old_manifold = cb.old_manifold;
				rec.manifold = cb.manifold;
				rec.material = cb.material;
				// (world edits made from the callback are taken like edits between steps: they reach the device before the next step)
				w->callbackWindow = true;
				if (w->preSolveBatchFn) w->preSolveBatchFn(w->preSolveUser, 1, &rec);
				else rec.enabled = w->preSolveFn(w->preSolveUser, rec.contact_index, rec.fixture_a, rec.fixture_b, &rec.old_manifold, &rec.manifold, &rec.material) ? 1 : 0;
				w->callbackWindow = false;
				int bits[3];
				memcpy(bits, &rec.material, sizeof(bits));
				v = make_int4(1 | (rec.enabled ? 0 : 2), bits[0], bits[1], bits[2]);
				// (bit 4 of the kind: the sub-step went on as if the contact had been switched off - the listener's last answer)
				const bool assumedOff = (r.info.x & 16) != 0;
				again = (rec.enabled != 0) == assumedOff || memcmp(&rec.material.friction, &r.mat.x, 4) != 0 || memcmp(&rec.material.restitution, &r.mat.y, 4) != 0 ||
					memcmp(&rec.material.tangent_speed, &r.mat.z, 4) != 0;
			}
			w->toiVerdicts.push_back(v);
			asked = k + 1;
		}
		if (!again) return 0;
		if (!w->toiSnapshotTaken) return setError(B2HIP_ERR_INVALID, "PreSolve changed a contact inside a TOI sub-step, and the phase kept no snapshot");
		if (round >= w->dw.capToiLog) return setError(B2HIP_ERR_INVALID, "TOI PreSolve rounds do not end");
		// (the host mirror of an edited body was refreshed from the state that is about to be taken back)
		if (!w->dirtyList.empty() || !w->editOps.empty() || !w->proxyEdits.empty() || !w->pendingMoves.empty())
			return setError(B2HIP_ERR_INVALID, "a PreSolve inside a TOI sub-step edited the world AND changed its contact: not supported");
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		int rc = uploadToiVerdicts(w);
		if (rc) return rc;
		w->toiChains = false;
		w->toiSpeculative = false;
		rc = phaseToiSync(w);
		if (rc) return rc;
		rc = downloadState(w, -1);
		if (rc) return rc;
		rc = settleToiOverflow(w);
		if (rc) return rc;
		w->toiPreSolveReruns += 1;
	}
}

static int stepEndImpl(b2hip_world* w)
{
	int rc = downloadState(w, -1, w->sp.dt > 0.0f); // (rows only if this is the last read-back of the step: k_end_step)
	if (rc) return rc;
	// optimistic small-sort path overflowed (or the pair buffer itself): finish the pair update with the radix path (after
	// growing the buffer and searching again), then read back again
	// bit 1: candidate-pair buffer; bit 0 with moves still buffered: the new contacts did not fit and creation was skipped
	// as a whole (createBlocked) - both are cured by growing and running the pair update again. Bit 0 without buffered
	// moves comes from a contact created inside a TOI sub-step: that one is lost.
	const bool pairOverflow = (w->h_dstate->c.overflow & 2) != 0 || ((w->h_dstate->c.overflow & 1) != 0 && w->h_dstate->c.nMoves != 0);
	if ((w->h_dstate->c.overflow & 1) != 0 && !pairOverflow) return setError(B2HIP_ERR_CAPACITY, "contact array full during a TOI sub-step");
	if (pairOverflow && w->sp.dt <= 0.0f) return setError(B2HIP_ERR_CAPACITY, "pair buffer overflow");
	if ((w->h_dstate->c.nMoves != 0 || pairOverflow) && w->sp.dt > 0.0f && w->stepSolves)
	{
		if (w->h_dstate->c.nPairs > COUNT_RANK_MAX || pairOverflow)
		{
			w->pairsLargeSticky = 16; // (the next steps ask for the pair count right after the search: findNewContactsGraph)
			const bool redoToi = w->toiSpeculative;
			if (redoToi && w->h_dstate->c.nToiList > 0)
			{
				// the TOI phase ran without the contacts that are created only now: undo it
				LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
			}
			if (pairOverflow)
			{
				rc = growPairBuffers(w);
				if (rc) return rc;
				rc = findNewContacts(w, true);
			}
			else rc = runSortAndCreate(w, true, w->h_dstate->c.nPairs);
			if (rc) return rc;
			if (redoToi)
			{
				w->toiChains = false;
				w->toiSpeculative = false;
				rc = phaseToiSync(w);
				if (rc) return rc;
			}
			rc = downloadState(w, -1);
			if (rc) return rc;
		}
	}
	if (w->toiSpeculative)
	{
		const Counters& tc = w->h_dstate->c;
		if (tc.nToiList > w->dw.capContacts) return setError(B2HIP_ERR_CAPACITY, "TOI list overflow");
		w->last.nToiList = tc.nToiList;
		w->last.nToiCalls = tc.nToiCalls;
		w->toiRan = tc.nToiList > 0;
		if (tc.nToiList == 0) w->toiChains = false;
	}
	if (w->toiChains)
	{
		if (w->h_dstate->c.nToiMoved > 0) w->toiGridSticky = 16;
		else if (w->toiGridSticky > 0) w->toiGridSticky -= 1;
	}
	if (w->toiChains && w->h_dstate->c.toiUnsafe != 0)
	{
		// a chain met an order-dependent case: back to the state before the chains, then the reference's serial order
		LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		if (w->h_dstate->c.toiUnsafe == 4 /* TOI_UNSAFE_PAIR */ && !w->toiChainsHadGrid && !w->dw.noChainCreate)
		{
			// ... unless all that happened is that a chain moved a proxy out of its fat AABB while the hash grid was not kept
			// up (nothing had moved for 16 steps): the chains once more, with the grid - most such moves find nothing, or a
			// pair the chains' close-out can create itself. (The serial loop costs ~50 us per event: 15 ms for the 290 resting
			// impacts of a 50 000-box pyramid; this costs a second first pass.)
			w->toiGridSticky = 16;
			w->toiChains = false;
			w->toiSpeculative = false;
			rc = phaseToiSync(w);
			if (rc) return rc;
			rc = downloadState(w, -1);
			if (rc) return rc;
			w->toiGridRetries += 1;
			if (w->toiChains && w->h_dstate->c.toiUnsafe != 0) LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
		}
	}
	if (w->toiChains && w->h_dstate->c.toiUnsafe != 0)
	{
		if (getenv("B2HIP_TOI_WHY")) fprintf(stderr, "b2hip: TOI fallback to the serial loop, unsafe bits 0x%x (1 partner, 2 woke, 4 new pair, 8 capacity, 16 moved proxies), %d pending, %d components\n", w->h_dstate->c.toiUnsafe, w->h_dstate->c.nToiList, w->h_dstate->c.nToiDomains);
		rc = toiSerial(w);
		if (rc) return rc;
		w->toiFallbacks += 1;
		w->toiSyncSticky = 16;
		rc = downloadState(w, -1);
		if (rc) return rc;
	}
	rc = settleToiOverflow(w);
	if (rc) return rc;
	w->postSolve.clear();
	if (w->postSolveOn)
	{
		const int n = std::min(w->h_dstate->c.nPostSolve, w->dw.capContacts);
		if (n > 0)
		{
			std::vector<PostSolveRec> recs(n);
			HIP_TRY(hipMemcpy((void*)recs.data(), w->postRecs.p, (size_t)n * sizeof(PostSolveRec), hipMemcpyDeviceToHost));
			std::vector<std::pair<unsigned long long, int> > order(n);
			for (int i = 0; i < n; ++i) order[i] = std::make_pair(recs[i].key, i);
			std::sort(order.begin(), order.end(), keyLess); // b2DeferredPostSolveLessThan: proxy-id pair
			w->postSolve.resize(n);
			for (int k = 0; k < n; ++k)
			{
				const PostSolveRec& r = recs[order[k].second];
				b2hip_contact_impulse& o = w->postSolve[k];
				o.contact_index = r.info.x;
				o.fixture_a = r.info.y;
				o.fixture_b = r.info.z;
				o.count = r.info.w;
				o.normal_impulses[0] = r.imp.x; o.tangent_impulses[0] = r.imp.y;
				o.normal_impulses[1] = r.imp.z; o.tangent_impulses[1] = r.imp.w;
			}
		}
	}
	w->toiCallbacks.clear();
	if (w->dw.toiLog != nullptr)
	{
		std::vector<ToiLogRec> recs;
		rc = toiPreSolveRounds(w, recs);
		if (rc) return rc;
		for (size_t k = 0; k < recs.size(); ++k)
		{
			b2hip_toi_callback cb;
			toiCallbackFromLog(recs[k], &cb);
			cb.kind &= ~(4 | 16); // (PreSolve has been called: toiPreSolveRounds)
			if (cb.kind == 0) continue; // (an Update that called nothing else: the contact neither began nor ended)
			w->toiCallbacks.push_back(cb);
		}
	}
	w->events.clear();
	// (the rows were left out of a read-back because another one was due, and it did not come: safety net, never seen)
	if (w->h_dstate->c.rowsSkipped == 1)
	{
		rc = downloadState(w, -1);
		if (rc) return rc;
	}
	if (w->eventsOn)
	{
		// after every fallback has had its say: one pass over the contacts, then the (usually short) list comes back
		LAUNCH(w, k_contact_events, gridFor(w->dw.capContacts), 256, w->dw);
		int nEv = 0;
		HIP_TRY(hipMemcpyAsync(&nEv, &w->d_state.p->c.nEvents, sizeof(int), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		if (nEv > w->dw.capContacts) return setError(B2HIP_ERR_CAPACITY, "contact event buffer overflow");
		if (nEv > 0)
		{
			std::vector<unsigned long long> keys(nEv);
			std::vector<int4> info(nEv);
			HIP_TRY(hipMemcpy(keys.data(), w->evKey.p, nEv * sizeof(unsigned long long), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(info.data(), w->evInfo.p, nEv * sizeof(int4), hipMemcpyDeviceToHost));
			std::vector<int> order(nEv);
			for (int i = 0; i < nEv; ++i) order[i] = i;
			// begins before ends, each group by proxy-id pair (b2ContactManager.cpp:420-438, b2ContactPointerLessThan :64-67)
			std::sort(order.begin(), order.end(), [&](int a, int b)
			{
				if (info[a].z != info[b].z) return info[a].z < info[b].z;
				return keys[a] < keys[b];
			});
			w->events.resize(nEv);
			for (int i = 0; i < nEv; ++i)
			{
				const int4 q = info[order[i]];
				w->events[i].fixture_a = q.x;
				w->events[i].fixture_b = q.y;
				w->events[i].kind = q.z;
				w->events[i].contact_index = q.w;
			}
		}
	}
	// (no synchronisation here: the read-back has arrived - awaitState - and k_end_step was the last thing on the stream)
	if (w->debugSync) HIP_TRY(hipStreamSynchronize(w->stream));
	refreshMirror(w);
	const Counters& c = w->h_dstate->c;
	w->lastContacts = c.nContacts;
	w->last.nContacts = c.nContacts;
	w->last.nMoves = c.nMoves;
	if (c.nMovesSeen > 256 && !w->gridForced)
	{
		// the grid's cell for the next step, from what this step's pair search went through (b2d_kernels_broadphase.h: gridCell)
		long long rounds = 0;
		for (int k = 0; k < 32; ++k) rounds += c.candRounds[k];
		const double perProxy = 64.0 * (double)rounds / (double)c.nMovesSeen;
		if (perProxy > 128.0) w->gridHalf = true; else if (perProxy < 48.0) w->gridHalf = false;
		w->dw.gridHalf = w->gridHalf ? 1 : 0; // (between steps: every kernel of the next step sees the same geometry)
	}
	w->last.nNewContacts = c.nNewContacts;
	w->last.nPairs = c.nPairs;
	w->last.overflow = c.overflow;
	if (w->colorSmallPending)
	{
		w->colorSmallPending = false;
		w->last.nColors = c.nColors;
		if (c.overflow & 4) return setError(B2HIP_ERR_CAPACITY, "more than 64 constraint colours on one body");
		if (c.nUncolored != 0) return setError(B2HIP_ERR_CAPACITY, "incremental colouring did not converge");
	}
	if (c.overflow & 64) return setError(B2HIP_ERR_HIP, "grid barrier of k_solve_persistent timed out (a workgroup was not resident)");
	if (c.overflow & SCAN_ABORT_BIT) return setError(B2HIP_ERR_HIP, "a single-pass scan gave up waiting for a predecessor tile (k_scan_chain look-back)");
	w->last.posItersLarge = c.posItersLarge;
	w->last.nHubRows = c.nHubRows;
	w->last.hubRounds = c.hubRounds;
	w->last.hubSerialChunks = c.hubSerialChunks;
	w->toiChainContacts += c.nToiChainCreated;
	if (w->toiRan)
	{
		w->last.toiUnsafe = c.toiUnsafe;
		w->last.nToiEvents = c.nToiEvents;
		w->last.nToiCalls = c.nToiCalls;
		w->last.toiOverflow = c.toiOverflow;
		if (c.toiOverflow) return setError(B2HIP_ERR_CAPACITY, "TOI event scratch overflow (flags " + std::to_string(c.toiOverflow) + ")");
	}
	if (w->sp.dt > 0.0f) w->inv_dt0 = w->sp.inv_dt;
	// b2World::m_stepComplete (b2World.cpp:1072, 1084): only SolveTOI changes it
	if (w->def.continuous && w->sp.dt > 0.0f) w->stepComplete = c.toiIncomplete == 0;
	w->stepActive = false;

	// b2Profile from events (milliseconds)
	float ms = 0.0f;
	float* p = w->profile;
	memset(p, 0, sizeof(float) * 13);
	{
		const unsigned long long* pc0 = w->h_dstate->phaseClock;
		p[0] = pc0[13] > pc0[14] ? 1.0e-5f * (float)(pc0[13] - pc0[14]) : 0.0f;      // step (device clock, read-back included)
	}
	if (w->profileDetail)
	{
	// the other figures: device clock (10 ns ticks) at the start of the first kernel of each phase (stampPhase)
	const unsigned long long* pc = w->h_dstate->phaseClock;
	auto span = [pc](int a, int b) -> float { return pc[b] > pc[a] ? 1.0e-5f * (float)(pc[b] - pc[a]) : 0.0f; };
	p[1] = span(1, 2);                                                           // collide
	p[2] = span(2, 3);                                                           // solve (islands + solver)
	p[3] = span(2, 4);                                                           // solveTraversal = island build
	const float dfs = span(4, 5), small = span(5, 6), color = span(6, 7), large = span(7, 8);
	p[3] += dfs + color;
	p[5] = small + large;                                                        // solver kernels (init+velocity+position)
	const float bpTop = span(0, 1), bp0 = span(3, 9), bp1 = span(9, 10);
	p[10] = bp0;                                                                 // broadphaseSyncFixtures
	p[11] = bp1 + bpTop;                                                         // broadphaseFindContacts
	p[9] = bp0 + bp1 + bpTop;                                                    // broadphase
	if (w->toiEventValid) p[7] = span(10, 12);                                   // solveTOI
	w->solverMs = small + large;
	if (w->blocksThisStep && w->h_dstate->stamps[4] > 0)
	{
		// b2Profile::solveInit / solveVelocity / solvePosition (b2TimeStep.h:30-32) from the block solver's own phase stamps:
		// [0] constraints initialised, [1] velocity iterations done, [2] positions integrated, [3] position iterations done,
		// [4] written back (10 ns ticks); scaled to the event-measured span of the launch (stamps are workgroup 0's view)
		const float tick = 1.0e-5f; // ms
		const float total = tick * (float)w->h_dstate->stamps[4];
		const float scale = total > 0.0f ? large / total : 0.0f;
		p[4] = scale * tick * (float)w->h_dstate->stamps[0];
		p[5] = small + scale * tick * (float)(w->h_dstate->stamps[2] - w->h_dstate->stamps[0]);
		p[6] = scale * tick * (float)(w->h_dstate->stamps[4] - w->h_dstate->stamps[2]);
	}
	}
	const int Ct = w->last.nSContacts + w->last.nLContacts;
	const int B = w->last.nSBodies + w->last.nLBodies;
	w->solverConstraints = Ct;
	w->solverBodies = B;
	// Ct*(Nv*220 + Np*136 + 488) + B*240 with Np = configured position iterations
	w->solverBytes = (double)Ct * (w->sp.velIters * 220.0 + w->sp.posIters * 136.0 + 488.0) + (double)B * 240.0;
	w->ktMs = 0.0f;
	w->ktLaunches = 0;
	w->ktBytes = 0.0;
	if (w->kernelTiming && w->ktUsed >= 2)
	{
		for (int k = 0; k + 1 < w->ktUsed; k += 2)
		{
			float t = 0.0f;
			(void)hipEventElapsedTime(&t, w->ktEvents[k], w->ktEvents[k + 1]);
			w->ktMs += t;
			w->ktLaunches += 1;
		}
		// SURVEY.md 8d per-unit figures: collide 480 B per contact of two polygons (230 B otherwise: circles), sync fixtures
		// 250 B per proxy, pair update 16 B per proxy read + 8 B per candidate pair written
		if (w->ktKind == 5) w->ktBytes = (double)w->ktUnitsA * 480.0 + (double)w->ktUnitsB * 230.0;
		else if (w->ktKind == 6) w->ktBytes = (double)w->ktUnitsA * 250.0;
		else if (w->ktKind == 7) w->ktBytes = (double)w->ktUnitsA * 16.0 + (double)w->ktUnitsB * 8.0;
		else if (w->ktKind == 1) w->ktBytes = (double)w->last.nLContacts * 220.0 * w->sp.velIters;
		else if (w->ktKind == 8)
		{
			// the whole family: SURVEY 8d's solver figure for the large islands, position iterations as executed
			w->ktBytes = (double)w->last.nLContacts * (w->sp.velIters * 220.0 + w->last.posItersLarge * 136.0 + 488.0) + (double)w->last.nLBodies * 240.0;
			w->ktLaunches = w->familyLaunches;
		}
		else if (w->ktKind == 3 || w->ktKind == 4) w->ktBytes = (double)w->last.nLContacts * (w->sp.velIters * 220.0 + w->last.posItersLarge * 136.0 + 488.0) + (double)w->last.nLBodies * 240.0;
		else w->ktBytes = (double)w->last.nSContacts * (w->sp.velIters * 220.0 + w->sp.posIters * 136.0 + 488.0) + (double)w->last.nSBodies * 240.0;
	}
	return 0;
}

int b2hip_set_lazy_readback(b2hip_world* w, int enable)
{
	if (int rc = checkUsable(w, "b2hip_set_lazy_readback", true)) return rc;
	ensureRows(w); // (switching off with rows outstanding: they come home now)
	w->lazyReadback = enable != 0;
	return w->failed ? setError(B2HIP_ERR_HIP, w->failedWhy) : B2HIP_OK;
}

int b2hip_step_end(b2hip_world* w)
{
	if (int rc = checkUsable(w, "b2hip_step_end", false)) return rc;
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_step_end outside a step");
	DEVICE_GUARD(w);
	return stepFailed(w, stepEndImpl(w));
}

int b2hip_step(b2hip_world* w, float dt, int velocity_iterations, int position_iterations)
{
	int rc = b2hip_step_begin(w, dt, velocity_iterations, position_iterations);
	if (rc) return rc;
	rc = b2hip_collide(w);
	if (rc) return rc;
	rc = b2hip_solve(w);
	if (rc) return rc;
	rc = b2hip_sync_fixtures(w);
	if (rc) return rc;
	rc = b2hip_find_new_contacts(w);
	if (rc) return rc;
	rc = b2hip_solve_toi(w);
	if (rc) return rc;
	return b2hip_step_end(w);
}

int b2hip_get_body_states(b2hip_world* w, int first, int count, b2hip_body_state* out)
{
	if (!w || !out || first < 0 || count < 0 || first + count > (int)w->bodies.size()) return setError(B2HIP_ERR_INVALID, "bad range");
	ensureRows(w);
	if (w->failed) return setError(B2HIP_ERR_INVALID, "b2hip_get_body_states: the world is in a failed state (" + w->failedWhy + ")");
	for (int i = 0; i < count; ++i)
	{
		const HostBody& b = w->bodies[first + i];
		b2hip_body_state& s = out[i];
		if (!b.dirty && b.pullEpoch != w->mirrorEpoch && (size_t)(first + i) < w->stateCount)
		{
			// straight from the pinned read-back buffer (same 40-byte layout)
			memcpy(&s, w->h_state + 10 * (size_t)(first + i), sizeof(b2hip_body_state));
			s.flags = (s.flags & 0x7cu) | (uint32_t)b.type;
			continue;
		}
		s.px = b.px; s.py = b.py; s.angle = b.a;
		s.vx = b.vx; s.vy = b.vy; s.w = b.w;
		s.cx = b.cx; s.cy = b.cy;
		s.flags = (b.flags & 0x7cu) | (uint32_t)b.type;
		s.sleep_time = b.sleepTime;
	}
	return 0;
}

// Edits queued since the last step (destroyed bodies / fixtures ...) change the contact list at once in the reference:
// whoever looks at the contacts between steps sees them applied.
static int flushForRead(b2hip_world* w)
{
	if (w->editOps.empty() || w->stepActive || w->failed) return 0;
	DEVICE_GUARD(w);
	int rc = flushEdits(w);
	if (rc) return rc;
	return applyEditOps(w, true);
}

int b2hip_contact_count(b2hip_world* w)
{
	if (!w) return 0;
	(void)flushForRead(w);
	return w->lastContacts;
}

int b2hip_enable_contact_events(b2hip_world* w, int enable)
{
	if (int rcu = checkUsable(w, "b2hip_enable_contact_events", true)) return rcu;
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->eventsOn = enable != 0;
	w->dw.eventsOn = w->eventsOn ? 1 : 0;
	w->events.clear();
	return B2HIP_OK;
}

int b2hip_get_contact_events(b2hip_world* w, int cap, b2hip_contact_event* out)
{
	if (!w || (cap > 0 && !out)) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = (int)w->events.size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = w->events[i];
	return n;
}

int b2hip_get_toi_callbacks(b2hip_world* w, int cap, b2hip_toi_callback* out)
{
	if (!w || (cap > 0 && !out)) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = (int)w->toiCallbacks.size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = w->toiCallbacks[i];
	return n;
}

// ---- snapshot -----------------------------------------------------------------------------------------
namespace
{
struct SnapHeader
{
	char magic[8];
	uint32_t version, szDState, szHostBody, szHostFixture, szShape, szJoint;
	uint32_t nBodies, nFixtures, nShapes, nJoints, nFree, nContacts, nToiOrder, nMoves;
	uint32_t stateCount, cur;
	int32_t nextNode, leafCount, lastContacts, newFixture;
	float inv_dt0, cellSize;
	int32_t eventsOn, solverHints; // solverHints: bit 0 serialOrphansNext, bit 1 blocksTooBig, bit 2 step incomplete (sub-stepping), bits 3..7 + 31 freshColors, bits 8..15 adoptSticky, 16..23 largeHintSteps, 24..30 recolorCountdown (what the next island build is told)
};
const uint32_t kSnapVersion = 5;
const char kSnapMagic[8] = { 'B', '2', 'H', 'I', 'P', 'S', 'N', '1' };

struct SnapWriter
{
	std::vector<char> out;
	void host(const void* p, size_t n) { const char* c = (const char*)p; out.insert(out.end(), c, c + n); }
	int dev(const void* p, size_t n)
	{
		const size_t at = out.size();
		out.resize(at + n);
		if (n == 0) return 0;
		HIP_TRY(hipMemcpy(out.data() + at, p, n, hipMemcpyDeviceToHost));
		return 0;
	}
};

struct SnapReader
{
	const char* p;
	size_t left;
	bool ok;
	const void* take(size_t n)
	{
		if (n > left) { ok = false; return nullptr; }
		const void* r = p;
		p += n;
		left -= n;
		return r;
	}
	void host(void* dst, size_t n) { const void* s = take(n); if (s && n) memcpy(dst, s, n); }
	int dev(void* dst, size_t n)
	{
		const void* s = take(n);
		if (!s) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
		if (n) HIP_TRY(hipMemcpy(dst, s, n, hipMemcpyHostToDevice));
		return 0;
	}
};
} // namespace

int b2hip_save_snapshot(b2hip_world* w, void* buffer, size_t cap, size_t* needed)
{
	if (!w || !needed || (cap > 0 && !buffer)) return setError(B2HIP_ERR_INVALID, "null argument");
	if (int rcu = checkUsable(w, "b2hip_save_snapshot", true)) return rcu;
	DEVICE_GUARD(w);
	ensureRows(w); // (the snapshot carries the host's mirror)
	int rc = flushEdits(w); // everything the host has created or edited is on the device now
	if (rc) return rc;
	rc = applyPendingFilters(w); // ... including the re-filter flags of joints created / destroyed since the last step
	if (rc) return rc;
	rc = applyEditOps(w, true);
	if (rc) return rc;
	rc = readState(w);
	if (rc) return rc;
	const DState& ds = *w->h_dstate;
	const size_t nb = w->bodies.size(), np = w->fixtures.size();
	const size_t nC = (size_t)std::max(ds.c.nContacts, 0), nM = (size_t)std::max(std::min(ds.c.nMoves, w->dw.capMoves), 0);
	const size_t nT = (size_t)std::max(ds.c.nToiOrder, 0);
	SnapHeader h;
	memset(&h, 0, sizeof(h));
	memcpy(h.magic, kSnapMagic, 8);
	h.version = kSnapVersion;
	h.szDState = sizeof(DState); h.szHostBody = (uint32_t)offsetof(HostBody, fixtures); h.szHostFixture = sizeof(HostFixture);
	h.szShape = sizeof(ShapeRec); h.szJoint = sizeof(RevoluteJoint);
	h.nBodies = (uint32_t)nb; h.nFixtures = (uint32_t)np; h.nShapes = (uint32_t)w->shapes.size(); h.nJoints = (uint32_t)w->joints.size();
	h.nFree = (uint32_t)w->freeUnits.size(); h.nContacts = (uint32_t)nC; h.nToiOrder = (uint32_t)nT; h.nMoves = (uint32_t)nM;
	h.stateCount = (uint32_t)std::min(w->stateCount, nb); h.cur = (uint32_t)ds.cur;
	h.nextNode = w->nextNode; h.leafCount = w->leafCount; h.lastContacts = w->lastContacts; h.newFixture = w->newFixture ? 1 : 0;
	h.inv_dt0 = w->inv_dt0; h.cellSize = w->dw.cellSize;
	h.eventsOn = w->eventsOn ? 1 : 0;
	h.solverHints = (w->serialOrphansNext ? 1 : 0) | (w->blocksTooBig ? 2 : 0) | (w->stepComplete ? 0 : 4) | ((w->adoptSticky & 0xff) << 8) | ((w->largeHintSteps & 0xff) << 16) | ((w->recolorCountdown & 0x7f) << 24) |
		((w->freshColors & 0x1f) << 3) | (int32_t)(((uint32_t)(w->freshColors >> 5) & 1u) << 31); // (bits 3..7 and 31: freshColors, 0..63)
	SnapWriter o;
	o.host(&h, sizeof(h));
	o.host(&w->def, sizeof(w->def));
	for (size_t i = 0; i < nb; ++i)
	{
		// (forces left in a row from before the last step are stale in an auto-clear world: the loaded world treats every
		// saved force as pending)
		HostBody row;
		memcpy((void*)&row, &w->bodies[i], offsetof(HostBody, fixtures));
		const HostBody& src = w->bodies[i];
		const bool current = src.dirty || src.pullEpoch == w->mirrorEpoch || i >= w->stateCount;
		if (w->def.auto_clear_forces && !(current && src.forceEpoch == w->stepEpoch) && i < w->stateCount) { row.fx = row.fy = row.torque = 0.0f; }
		o.host(&row, offsetof(HostBody, fixtures));
		const char dirty = w->bodies[i].dirty ? 1 : 0;
		o.host(&dirty, 1);
	}
	o.host(w->fixtures.data(), np * sizeof(HostFixture));
	o.host(w->shapes.data(), w->shapes.size() * sizeof(ShapeRec));
	o.host(w->freeUnits.data(), w->freeUnits.size() * sizeof(FreeUnit));
	o.host(w->h_state, (size_t)h.stateCount * 10 * sizeof(float));
	o.host(&ds, sizeof(DState));
#define SNAP_DEV(arr, n) do { rc = o.dev(w->arr.p, (size_t)(n) * sizeof(*w->arr.p)); if (rc) return rc; } while (0)
	SNAP_DEV(d_joints, w->joints.size()); // the device copy carries the accumulated impulses
	SNAP_DEV(b_pos, nb); SNAP_DEV(b_pos0, nb); SNAP_DEV(b_vel, nb); SNAP_DEV(b_xf, nb); SNAP_DEV(b_mass, nb); SNAP_DEV(b_damp, nb);
	SNAP_DEV(b_force, nb); SNAP_DEV(b_flags, nb); SNAP_DEV(b_wake, nb); SNAP_DEV(b_proxyHead, nb); SNAP_DEV(b_blk1, nb);
	SNAP_DEV(p_fat, np); SNAP_DEV(p_body, np); SNAP_DEV(p_shape, np); SNAP_DEV(p_key, np); SNAP_DEV(p_filter0, np); SNAP_DEV(p_filter1, np);
	SNAP_DEV(p_mat, np); SNAP_DEV(p_next, np);
	const int cur = ds.cur;
	SNAP_DEV(c_ids[cur], nC); SNAP_DEV(c_key[cur], nC); SNAP_DEV(c_flags[cur], nC); SNAP_DEV(c_mat[cur], nC); SNAP_DEV(c_man0[cur], nC);
	SNAP_DEV(c_man1[cur], nC); SNAP_DEV(c_imp[cur], nC); SNAP_DEV(c_man3[cur], nC); SNAP_DEV(c_color[cur], nC); SNAP_DEV(c_mgr[cur], nC);
	SNAP_DEV(toiPos2c, nT); SNAP_DEV(moveBuf, nM);
	{
		// trailing section (absent in snapshots of worlds saved before gear joints existed): the gear records
		const uint32_t tail[2] = { (uint32_t)w->gears.size(), (uint32_t)sizeof(GearRec) };
		o.host(tail, sizeof(tail));
		SNAP_DEV(d_gears, w->gears.size());
	}
#undef SNAP_DEV
	*needed = o.out.size();
	if (cap >= o.out.size()) memcpy(buffer, o.out.data(), o.out.size());
	else if (cap > 0) return setError(B2HIP_ERR_CAPACITY, "snapshot buffer too small");
	return B2HIP_OK;
}

// Everything in the blob is checked BEFORE anything is copied to the device or used as an index: the counts against each
// other and against the blob's size, every body / fixture / shape / joint / gear / proxy / contact index against its range.
// A truncated or bit-flipped snapshot is refused with B2HIP_ERR_INVALID; it never writes out of bounds.
int b2hip_load_snapshot(const void* buffer, size_t size, int device, b2hip_world** out)
{
	if (!buffer || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	SnapReader in = { (const char*)buffer, size, true };
	SnapHeader h;
	in.host(&h, sizeof(h));
	if (!in.ok || memcmp(h.magic, kSnapMagic, 8) != 0 || h.version != kSnapVersion || h.szDState != sizeof(DState) ||
		h.szHostBody != offsetof(HostBody, fixtures) || h.szHostFixture != sizeof(HostFixture) || h.szShape != sizeof(ShapeRec) ||
		h.szJoint != sizeof(RevoluteJoint))
		return setError(B2HIP_ERR_INVALID, "not a snapshot of this build of libb2hip");
	b2hip_world_def def;
	in.host(&def, sizeof(def));
	if (!in.ok) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	auto corrupt = [](const char* what) { return setError(B2HIP_ERR_INVALID, std::string("snapshot corrupt: ") + what); };
	const size_t nb = h.nBodies, np = h.nFixtures, nC = h.nContacts, nS = h.nShapes, nJ = h.nJoints, nT = h.nToiOrder, nM = h.nMoves;
	// no count can exceed what the blob could hold at all (this also keeps the size products below from overflowing)
	if (nb > size || np > size || nC > size || nS > size || nJ > size || h.nFree > size || nT > size || nM > size || h.stateCount > size)
		return corrupt("counts exceed the blob");
	if (h.stateCount > nb || h.cur > 1u || nT > nC || h.lastContacts < 0 || (size_t)h.lastContacts > nC) return corrupt("header counts");
	if (h.nextNode < 0 || h.leafCount < 0 || (size_t)h.leafCount > np || (size_t)h.nextNode > 2 * np + 2) return corrupt("proxy id allocator");
	if (!(h.cellSize > 0.0f) || !std::isfinite(h.cellSize)) return corrupt("cell size");

	// ---- host sections -------------------------------------------------------------------------------------------------
	const size_t bodyBytes = offsetof(HostBody, fixtures) + 1;
	const char* bodiesAt = (const char*)in.take(nb * bodyBytes);
	const HostFixture* fixturesAt = (const HostFixture*)in.take(np * sizeof(HostFixture));
	const ShapeRec* shapesAt = (const ShapeRec*)in.take(nS * sizeof(ShapeRec));
	const FreeUnit* freeAt = (const FreeUnit*)in.take((size_t)h.nFree * sizeof(FreeUnit));
	const float* stateAt = (const float*)in.take((size_t)h.stateCount * 10 * sizeof(float));
	const DState* dsAt = (const DState*)in.take(sizeof(DState));
	const RevoluteJoint* jointsAt = (const RevoluteJoint*)in.take(nJ * sizeof(RevoluteJoint));
	if (!in.ok) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	for (size_t i = 0; i < nb; ++i)
	{
		HostBody hb;
		memcpy((void*)&hb, bodiesAt + i * bodyBytes, offsetof(HostBody, fixtures));
		if (hb.type < 0 || hb.type > 2) return corrupt("body type");
	}
	for (size_t f = 0; f < np; ++f)
	{
		HostFixture hf;
		memcpy(&hf, fixturesAt + f, sizeof(hf));
		if (hf.body < 0 || (size_t)hf.body >= nb || hf.shape < 0 || (size_t)hf.shape >= nS || hf.proxyKey < 0) return corrupt("fixture");
	}
	for (size_t k = 0; k < nS; ++k)
	{
		ShapeRec sr;
		memcpy(&sr, shapesAt + k, sizeof(sr));
		if (sr.type < 0 || sr.type > 2 || sr.count < 0 || sr.count > B2D_MAX_POLY_VERTS) return corrupt("shape");
	}
	for (size_t k = 0; k < h.nFree; ++k)
	{
		FreeUnit fu;
		memcpy(&fu, freeAt + k, sizeof(fu));
		if (fu.leaf < -1 || fu.leaf >= h.nextNode) return corrupt("proxy id free list");
	}
	DState ds;
	memcpy(&ds, dsAt, sizeof(ds));
	if (ds.cur != (int)h.cur || ds.c.nContacts != (int)nC || ds.c.nToiOrder != (int)nT || ds.c.nMoves < (int)nM) return corrupt("device state block");

	// ---- device sections: located and range-checked in the blob, uploaded later ---------------------------------------------
	struct Sec { const void* p; size_t bytes; };
	auto sec = [&](size_t n, size_t elem) { Sec x = { in.take(n * elem), n * elem }; return x; };
	const Sec sPos = sec(nb, 16), sPos0 = sec(nb, 16), sVel = sec(nb, 16), sXf = sec(nb, 16), sMass = sec(nb, 16), sDamp = sec(nb, 16), sForce = sec(nb, 16);
	const Sec sFlags = sec(nb, 4), sWake = sec(nb, 4), sHead = sec(nb, 4), sBlk = sec(nb, 4);
	const Sec sFat = sec(np, 16), sPBody = sec(np, 4), sPShape = sec(np, 4), sPKey = sec(np, 4), sF0 = sec(np, 4), sF1 = sec(np, 4), sPMat = sec(np, 8), sNext = sec(np, 4);
	const Sec cIds = sec(nC, 16), cKey = sec(nC, 8), cFlags = sec(nC, 4), cMat = sec(nC, 16), cMan0 = sec(nC, 16), cMan1 = sec(nC, 16), cImp = sec(nC, 16),
		cMan3 = sec(nC, 16), cColor = sec(nC, 4), cMgr = sec(nC, 4);
	const Sec sToi = sec(nT, 4), sMoves = sec(nM, 4);
	if (!in.ok) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	auto inRange = [](const Sec& x, long long lo, long long hi) // every int of the section in [lo, hi)
	{
		const int* v = (const int*)x.p;
		for (size_t i = 0; i < x.bytes / 4; ++i)
		{
			int q;
			memcpy(&q, v + i, 4);
			if (q < lo || q >= hi) return false;
		}
		return true;
	};
	if (!inRange(sHead, -1, (long long)np) || !inRange(sNext, -1, (long long)np)) return corrupt("per-body proxy lists");
	if (!inRange(sBlk, 0, MAX_BLOCKS + 1) || ds.c.nBlocks < 0 || ds.c.nBlocks > MAX_BLOCKS) return corrupt("block partition");
	// (the proxy of a destroyed fixture stays in the table with body -1)
	if (!inRange(sPBody, -1, (long long)nb) || !inRange(sPShape, 0, (long long)nS)) return corrupt("proxy table");
	if (!inRange(sToi, 0, (long long)nC) || !inRange(sMoves, 0, (long long)np)) return corrupt("TOI order / move buffer");
	if (!inRange(cColor, -1, MAX_COLORS) || !inRange(cMgr, -1, (long long)std::max<size_t>(nT, 1))) return corrupt("contact colour / TOI slot");
	for (size_t i = 0; i < nC; ++i)
	{
		int4 ids;
		memcpy(&ids, (const char*)cIds.p + 16 * i, 16);
		if (ids.x < 0 || (size_t)ids.x >= np || ids.y < 0 || (size_t)ids.y >= np || ids.z < 0 || (size_t)ids.z >= nb || ids.w < 0 || (size_t)ids.w >= nb)
			return corrupt("contact ids");
	}
	// trailing section: the gear records (absent in snapshots of worlds that never had one)
	size_t nG = 0;
	const GearRec* gearsAt = nullptr;
	if (in.left >= 2 * sizeof(uint32_t))
	{
		uint32_t tail[2];
		in.host(tail, sizeof(tail));
		if (tail[1] != sizeof(GearRec) || tail[0] > size) return corrupt("gear section");
		nG = tail[0];
		gearsAt = (const GearRec*)in.take(nG * sizeof(GearRec));
		if (!gearsAt && nG) return setError(B2HIP_ERR_INVALID, "snapshot truncated");
	}
	for (size_t k = 0; k < nJ; ++k)
	{
		RevoluteJoint j;
		memcpy((void*)&j, jointsAt + k, sizeof(j));
		if (j.type < B2D_JOINT_DEAD || j.type > B2D_JOINT_GEAR) return corrupt("joint type");
		if (j.bodyA < 0 || (size_t)j.bodyA >= nb || j.bodyB < 0 || (size_t)j.bodyB >= nb) return corrupt("joint bodies");
		if (j.type == B2D_JOINT_GEAR && (j.enableLimit < 0 || (size_t)j.enableLimit >= nG)) return corrupt("gear index");
	}
	for (size_t k = 0; k < nG; ++k)
	{
		GearRec g;
		memcpy((void*)&g, gearsAt + k, sizeof(g));
		if (g.bodyC < 0 || (size_t)g.bodyC >= nb || g.bodyD < 0 || (size_t)g.bodyD >= nb) return corrupt("gear bodies");
	}

	// ---- build the world ---------------------------------------------------------------------------------------------------------
	def.device = device;
	b2hip_world* w = nullptr;
	int rc = b2hip_world_create(&def, &w);
	if (rc) return rc;
	DEVICE_GUARD(w);
	auto fail = [&](int code) { const std::string why = g_lastError; b2hip_world_destroy(w); return setError(code, why); };
	w->bodies.resize(nb);
	for (size_t i = 0; i < nb; ++i)
	{
		memcpy((void*)&w->bodies[i], bodiesAt + i * bodyBytes, offsetof(HostBody, fixtures));
		w->bodies[i].dirty = bodiesAt[i * bodyBytes + offsetof(HostBody, fixtures)] != 0;
		w->bodies[i].pullEpoch = 0;
		w->bodies[i].forceEpoch = w->stepEpoch;
		if (w->bodies[i].dirty) w->dirtyList.push_back((int)i);
	}
	{
		// m_nonStaticBodies from the saved slots
		size_t count = 0;
		for (size_t i = 0; i < nb; ++i) count += w->bodies[i].worldIndex >= 0 ? 1 : 0;
		w->nonStatic.assign(count, -1);
		for (size_t i = 0; i < nb; ++i)
		{
			const int k = w->bodies[i].worldIndex;
			if (k < 0) continue;
			if ((size_t)k >= count || w->nonStatic[(size_t)k] != -1) return fail(corrupt("non-static body order"));
			w->nonStatic[(size_t)k] = (int)i;
		}
		w->orderDirty = true;
	}
	w->fixtures.resize(np);
	if (np) memcpy(w->fixtures.data(), fixturesAt, np * sizeof(HostFixture));
	w->shapes.resize(nS);
	if (nS) memcpy((void*)w->shapes.data(), shapesAt, nS * sizeof(ShapeRec));
	w->freeUnits.resize(h.nFree);
	if (h.nFree) memcpy(w->freeUnits.data(), freeAt, (size_t)h.nFree * sizeof(FreeUnit));
	for (size_t f = 0; f < np; ++f) w->bodies[w->fixtures[f].body].fixtures.push_back((int)f);
	for (size_t k = 0; k < nS; ++k) w->shapeIndex[std::string((const char*)&w->shapes[k], sizeof(ShapeRec))] = (int)k;
	// joints / gears: the device copy is the truth (accumulated impulses); the host vectors mirror it and are uploaded, with
	// the per-body joint lists, by the next flushEdits
	w->joints.resize(nJ);
	if (nJ) memcpy((void*)w->joints.data(), jointsAt, nJ * sizeof(RevoluteJoint));
	for (size_t k = 0; k < nJ; ++k) w->nMouseJoints += w->joints[k].type == B2D_JOINT_MOUSE;
	w->gears.resize(nG);
	if (nG) memcpy((void*)w->gears.data(), gearsAt, nG * sizeof(GearRec));
	w->nextNode = h.nextNode; w->leafCount = h.leafCount; w->lastContacts = h.lastContacts; w->newFixture = h.newFixture != 0;
	w->inv_dt0 = h.inv_dt0;
	w->eventsOn = h.eventsOn != 0;
	w->serialOrphansNext = h.solverHints & 1;
	w->blocksTooBig = (h.solverHints & 2) != 0;
	w->stepComplete = (h.solverHints & 4) == 0;
	w->adoptSticky = (h.solverHints >> 8) & 0xff;
	w->largeHintSteps = (h.solverHints >> 16) & 0xff;
	w->recolorCountdown = (h.solverHints >> 24) & 0x7f;
	w->freshColors = ((h.solverHints >> 3) & 0x1f) | ((int)(((uint32_t)h.solverHints >> 31) & 1u) << 5);
	w->freshColorsPending = false;
	w->adoptPasses = w->adoptSticky > 0;
	rc = ensureCapacity(w, nC);
	if (rc) return fail(rc);
	if (nM > w->moveBuf.cap || nC > (size_t)w->dw.capContacts || (size_t)h.stateCount * 10 > w->h_stateCap) return fail(corrupt("counts exceed the buffers sized for them"));
	if (h.stateCount) memcpy(w->h_state, stateAt, (size_t)h.stateCount * 10 * sizeof(float));
	w->shadowDev = nullptr; // (the host's rows are the snapshot's now, not what the device last sent)
	w->stateCount = h.stateCount;
	*w->h_dstate = ds;
#define SNAP_UP(arr, s) do { if ((s).bytes && hipMemcpy(w->arr.p, (s).p, (s).bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(setError(B2HIP_ERR_HIP, "snapshot upload failed (" #arr ")")); } while (0)
	if (hipMemcpy(w->d_state.p, &ds, sizeof(DState), hipMemcpyHostToDevice) != hipSuccess) return fail(setError(B2HIP_ERR_HIP, "snapshot upload failed (state block)"));
	w->pubSeq = ds.pubCount; // (the device numbers its census publications; the host counts along)
	SNAP_UP(b_pos, sPos); SNAP_UP(b_pos0, sPos0); SNAP_UP(b_vel, sVel); SNAP_UP(b_xf, sXf); SNAP_UP(b_mass, sMass); SNAP_UP(b_damp, sDamp);
	SNAP_UP(b_force, sForce); SNAP_UP(b_flags, sFlags); SNAP_UP(b_wake, sWake); SNAP_UP(b_proxyHead, sHead); SNAP_UP(b_blk1, sBlk);
	SNAP_UP(p_fat, sFat); SNAP_UP(p_body, sPBody); SNAP_UP(p_shape, sPShape); SNAP_UP(p_key, sPKey); SNAP_UP(p_filter0, sF0); SNAP_UP(p_filter1, sF1);
	SNAP_UP(p_mat, sPMat); SNAP_UP(p_next, sNext);
	const int cur = (int)h.cur;
	SNAP_UP(c_ids[cur], cIds); SNAP_UP(c_key[cur], cKey); SNAP_UP(c_flags[cur], cFlags); SNAP_UP(c_mat[cur], cMat); SNAP_UP(c_man0[cur], cMan0);
	SNAP_UP(c_man1[cur], cMan1); SNAP_UP(c_imp[cur], cImp); SNAP_UP(c_man3[cur], cMan3); SNAP_UP(c_color[cur], cColor); SNAP_UP(c_mgr[cur], cMgr);
	SNAP_UP(toiPos2c, sToi); SNAP_UP(moveBuf, sMoves);
#undef SNAP_UP
	if (nS && hipMemcpy(w->d_shapes.p, w->shapes.data(), nS * sizeof(ShapeRec), hipMemcpyHostToDevice) != hipSuccess)
		return fail(setError(B2HIP_ERR_HIP, "snapshot upload failed (shapes)"));
	w->upBodies = nb; w->upFixtures = np; w->upShapes = nS; w->upJoints = 0;
	w->dw.cellSize = h.cellSize;
	w->dw.invCellSize = 1.0f / h.cellSize;
	w->dw.eventsOn = w->eventsOn ? 1 : 0;
	*out = w;
	return B2HIP_OK;
}

int b2hip_get_contacts(b2hip_world* w, int cap, b2hip_contact* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	int rc = flushForRead(w);
	if (rc) return rc;
	rc = readState(w);
	if (rc) return rc;
	const int n = std::min(cap, w->h_dstate->c.nContacts);
	const int cur = w->h_dstate->cur;
	if (n <= 0) return 0;
	std::vector<int4> ids(n), m3(n);
	std::vector<uint32_t> flags(n);
	std::vector<float4> mat(n), m0(n), m1(n), imp(n);
	HIP_TRY(hipMemcpy(ids.data(), w->c_ids[cur].p, n * sizeof(int4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(m3.data(), w->c_man3[cur].p, n * sizeof(int4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(flags.data(), w->c_flags[cur].p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(mat.data(), w->c_mat[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(m0.data(), w->c_man0[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(m1.data(), w->c_man1[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(imp.data(), w->c_imp[cur].p, n * sizeof(float4), hipMemcpyDeviceToHost));
	for (int i = 0; i < n; ++i)
	{
		b2hip_contact& c = out[i];
		c.fixture_a = ids[i].x;
		c.fixture_b = ids[i].y;
		c.body_a = ids[i].z;
		c.body_b = ids[i].w;
		c.flags = ((flags[i] & CF_TOUCHING) ? 1u : 0u) | ((flags[i] & CF_ENABLED) ? 2u : 0u);
		c.manifold_type = m3[i].z;
		c.point_count = m3[i].w;
		c.local_normal[0] = m0[i].x; c.local_normal[1] = m0[i].y;
		c.local_point[0] = m0[i].z; c.local_point[1] = m0[i].w;
		c.point_local[0][0] = m1[i].x; c.point_local[0][1] = m1[i].y;
		c.point_local[1][0] = m1[i].z; c.point_local[1][1] = m1[i].w;
		c.normal_impulse[0] = imp[i].x; c.tangent_impulse[0] = imp[i].y;
		c.normal_impulse[1] = imp[i].z; c.tangent_impulse[1] = imp[i].w;
		c.id_key[0] = (uint32_t)m3[i].x;
		c.id_key[1] = (uint32_t)m3[i].y;
		c.friction = mat[i].x;
		c.restitution = mat[i].y;
		c.tangent_speed = mat[i].z;
	}
	return n;
}

int b2hip_set_shard(b2hip_world* w, int rank, int count)
{
	if (int rcu = checkUsable(w, "b2hip_set_shard", true)) return rcu;
	// (the island census keeps one counter per rank: Counters::shardBodies[SHARD_MAX_RANKS] ...)
	if (count < 1 || count > SHARD_MAX_RANKS || rank < 0 || rank >= count) return setError(B2HIP_ERR_INVALID, "bad rank / count (at most 8 ranks)");
	w->dw.shardRank = rank;
	w->dw.shardCount = count;
	return B2HIP_OK;
}

// ---- the exchange of a sharded world (b2d_kernels_shard.h) -------------------------------------------------------------------
// words of rank r's slab this step, from the island census every rank keeps of every rank (read with the census the solver
// waited for anyway: no extra read-back)
static size_t shardSlabWords(const b2hip_world* w, int r)
{
	const Counters& c = w->h_dstate->c;
	return (size_t)c.shardBodies[r] * SHARD_BODY_WORDS + (size_t)c.shardContacts[r] * SHARD_CONTACT_WORDS + (size_t)c.shardJoints[r] * SHARD_JOINT_WORDS;
}

int b2hip_shard_slab_words(b2hip_world* w, size_t* words_per_rank, int ranks)
{
	if (!w || !words_per_rank) return setError(B2HIP_ERR_INVALID, "null argument");
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_shard_slab_words outside a step");
	if (ranks != w->dw.shardCount) return setError(B2HIP_ERR_INVALID, "rank count differs from b2hip_set_shard");
	for (int r = 0; r < ranks; ++r) words_per_rank[r] = shardSlabWords(w, r);
	return B2HIP_OK;
}

int b2hip_shard_export(b2hip_world* w, void* device_buffer, size_t words)
{
	if (!w || !device_buffer) return setError(B2HIP_ERR_INVALID, "null argument");
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_shard_export outside a step");
	DEVICE_GUARD(w);
	if (words < shardSlabWords(w, w->dw.shardRank)) return setError(B2HIP_ERR_CAPACITY, "slab buffer too small");
	LAUNCH(w, k_shard_export, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, (int*)device_buffer);
	HIP_TRY(hipStreamSynchronize(w->stream)); // (the CALLER's collective runs on a stream of its own; b2hip_shard_connect avoids this)
	return B2HIP_OK;
}

int b2hip_shard_import(b2hip_world* w, const void* device_buffer, size_t stride_words)
{
	if (!w || !device_buffer) return setError(B2HIP_ERR_INVALID, "null argument");
	if (!w->stepActive) return setError(B2HIP_ERR_INVALID, "b2hip_shard_import outside a step");
	DEVICE_GUARD(w);
	for (int r = 0; r < w->dw.shardCount; ++r)
		if (stride_words < shardSlabWords(w, r)) return setError(B2HIP_ERR_CAPACITY, "slab stride too small");
	LAUNCH(w, k_shard_import, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, (const int*)device_buffer, stride_words);
	return B2HIP_OK;
}

// ---- RCCL from inside the library: the all-gather of the slabs on the world's own stream --------------------------------------
// librccl is opened when a world is connected (not a link-time dependency: a single-GPU user never loads it).
namespace
{
struct RcclApi
{
	void* lib = nullptr;
	ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*allGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char* (*errorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rcclLoad()
{
	if (g_rccl.lib) return 0;
	void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) return setError(B2HIP_ERR_UNSUPPORTED, std::string("librccl could not be opened: ") + dlerror());
	g_rccl.getUniqueId = (decltype(g_rccl.getUniqueId))dlsym(lib, "ncclGetUniqueId");
	g_rccl.commInitRank = (decltype(g_rccl.commInitRank))dlsym(lib, "ncclCommInitRank");
	g_rccl.commDestroy = (decltype(g_rccl.commDestroy))dlsym(lib, "ncclCommDestroy");
	g_rccl.allGather = (decltype(g_rccl.allGather))dlsym(lib, "ncclAllGather");
	g_rccl.errorString = (decltype(g_rccl.errorString))dlsym(lib, "ncclGetErrorString");
	if (!g_rccl.getUniqueId || !g_rccl.commInitRank || !g_rccl.commDestroy || !g_rccl.allGather || !g_rccl.errorString)
		return setError(B2HIP_ERR_UNSUPPORTED, "librccl lacks a collective entry point");
	g_rccl.lib = lib;
	g_rcclDestroy = [](void* comm) { (void)g_rccl.commDestroy((ncclComm_t)comm); };
	return 0;
}
#define RCCL_TRY(call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) return setError(B2HIP_ERR_HIP, std::string(#call) + ": " + g_rccl.errorString(_r)); } while (0)
}

int b2hip_shard_unique_id(void* id128)
{
	if (!id128) return setError(B2HIP_ERR_INVALID, "null argument");
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
	if (int rc = rcclLoad()) return rc;
	ncclUniqueId id;
	RCCL_TRY(g_rccl.getUniqueId(&id));
	memcpy(id128, &id, sizeof(id));
	return B2HIP_OK;
}

int b2hip_shard_connect(b2hip_world* w, const void* id128, int rank, int count)
{
	if (int rcu = checkUsable(w, "b2hip_shard_connect", true)) return rcu;
	if (!id128 || count < 1 || count > SHARD_MAX_RANKS || rank < 0 || rank >= count) return setError(B2HIP_ERR_INVALID, "bad rank / count (at most 8 ranks)");
	if (w->shardComm) return setError(B2HIP_ERR_INVALID, "the world is connected already");
	if (int rc = rcclLoad()) return rc;
	DEVICE_GUARD(w);
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	ncclComm_t comm = nullptr;
	RCCL_TRY(g_rccl.commInitRank(&comm, count, id, rank));
	w->shardComm = comm;
	w->shardLoopback = getenv("B2HIP_SHARD_LOOPBACK") != nullptr && atoi(getenv("B2HIP_SHARD_LOOPBACK")) != 0;
	w->dw.shardRank = rank;
	w->dw.shardCount = count;
	return B2HIP_OK;
}

// export -> ncclAllGather -> import, all queued on the world's stream: no event, no host synchronisation
static int shardExchangeOnStream(b2hip_world* w)
{
	const int ranks = w->dw.shardCount;
	size_t stride = 1;
	for (int r = 0; r < ranks; ++r) stride = std::max(stride, shardSlabWords(w, r));
	// (buffers grow by doubling; a grown buffer is new memory, the old one is freed behind a stream synchronisation by ensure)
	int rc = w->shardSend.ensure(stride, w->stream, false, false);
	if (rc) return rc;
	rc = w->shardRecv.ensure(stride * (size_t)ranks, w->stream, false, false);
	if (rc) return rc;
	LAUNCH(w, k_shard_export, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, w->shardSend.p);
	RCCL_TRY(g_rccl.allGather(w->shardSend.p, w->shardRecv.p, stride, ncclInt32, (ncclComm_t)w->shardComm, w->stream));
	LAUNCH(w, k_shard_import, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, (const int*)w->shardRecv.p, stride);
	w->shardExchangeBytes = 4 * stride * (size_t)ranks;
	return 0;
}

int b2hip_shard_exchange_bytes(b2hip_world* w, size_t* bytes)
{
	if (!w || !bytes) return setError(B2HIP_ERR_INVALID, "null argument");
	*bytes = w->shardExchangeBytes;
	return B2HIP_OK;
}

// ---- spatial ownership (b2d_kernels_spatial.h) ----------------------------------------------------------------------------------
// The all-gather of `words` ints per rank from w->spSend into w->spRecv: RCCL on the world's stream when the world is
// connected, else the caller's collective over pinned host memory.
static int spAllGather(b2hip_world* w, size_t words)
{
	const int ranks = w->dw.shardCount;
	w->spBytesStep += 4 * words * (size_t)(ranks - 1);
	static const bool trace = getenv("B2HIP_SHARD_TRACE") && atoi(getenv("B2HIP_SHARD_TRACE"));
	if (trace && w->dw.shardRank == 0) fprintf(stderr, "[b2hip] step %lld: all-gather of %zu words per rank (caps: rows %d proxies %d pairs %d toi %d / %d / %d)\n",
		(long long)w->stepEpoch, words, w->spRowCap, w->spProxyCap, w->spPairCap, w->spToiBodyCap, w->spToiProxyCap, w->spTailCap);
	if (w->spTapeFrom != nullptr)
	{
		// (replay: what the collective delivered in the recorded run, device to device on the world's stream)
		const std::vector<std::pair<int*, size_t> >& tape = w->spTapeFrom->spTape;
		if (w->spTapeCursor >= tape.size() || tape[w->spTapeCursor].second != words * (size_t)ranks)
			return setError(B2HIP_ERR_INVALID, "the replayed run leaves the recorded one (collective " + std::to_string(w->spTapeCursor) + ")");
		HIP_TRY(hipMemcpyAsync(w->spRecv.p, tape[w->spTapeCursor].first, words * (size_t)ranks * sizeof(int), hipMemcpyDeviceToDevice, w->stream));
		// (this rank's own slab as it is NOW: record order inside a slab is not deterministic - atomics - and later kernels may
		// index into both)
		HIP_TRY(hipMemcpyAsync(w->spRecv.p + (size_t)w->dw.shardRank * words, w->spSend.p, words * sizeof(int), hipMemcpyDeviceToDevice, w->stream));
		w->spTapeCursor += 1;
		return 0;
	}
	if (w->shardComm != nullptr)
	{
		RCCL_TRY(g_rccl.allGather(w->spSend.p, w->spRecv.p, words, ncclInt32, (ncclComm_t)w->shardComm, w->stream));
		return 0;
	}
	if (!w->gatherFn) return setError(B2HIP_ERR_INVALID, "a spatially sharded world needs b2hip_shard_connect or b2hip_set_shard_gather");
	const size_t need = words * (size_t)(ranks + 1);
	if (w->spHostWords < need)
	{
		if (w->spHost) (void)hipHostFree(w->spHost);
		w->spHost = nullptr;
		w->spHostWords = 2 * need;
		HIP_TRY(hipHostMalloc((void**)&w->spHost, w->spHostWords * sizeof(int), hipHostMallocDefault));
	}
	HIP_TRY(hipMemcpyAsync(w->spHost, w->spSend.p, words * sizeof(int), hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	if (w->gatherFn(w->gatherUser, w->spHost, words * sizeof(int), w->spHost + words) != 0) return setError(B2HIP_ERR_INVALID, "the caller's all-gather failed");
	HIP_TRY(hipMemcpyAsync(w->spRecv.p, w->spHost + words, words * (size_t)ranks * sizeof(int), hipMemcpyHostToDevice, w->stream));
	if (w->spTapeRecord)
	{
		int* keep = nullptr;
		HIP_TRY(hipMalloc((void**)&keep, words * (size_t)ranks * sizeof(int)));
		HIP_TRY(hipMemcpyAsync(keep, w->spRecv.p, words * (size_t)ranks * sizeof(int), hipMemcpyDeviceToDevice, w->stream));
		w->spTape.push_back(std::make_pair(keep, words * (size_t)ranks));
	}
	return 0;
}

static int spEnsureSlabs(b2hip_world* w, size_t words)
{
	int rc = w->spSend.ensure(words, w->stream, false, false);
	if (rc) return rc;
	return w->spRecv.ensure(words * (size_t)w->dw.shardCount, w->stream, false, false);
}

// The header of the send slab is zero before an export counts into it: wiped by the import kernel of the exchange before
// (spSendWiped: by which launch, for which buffer), by a fill otherwise (first exchange, a slab that grew, an exchange that
// was repeated or left before its import).
static int spPrepareSend(b2hip_world* w)
{
	if (w->spSendWiped != w->spSend.p) HIP_TRY(hipMemsetAsync(w->spSend.p, 0, SP_HEADER_WORDS * sizeof(int), w->stream));
	w->spSendWiped = nullptr;
	return 0;
}

// the headers of all ranks' slabs, on the host (one small copy + synchronisation)
static int spReadHeaders(b2hip_world* w, size_t strideWords, int (*hdr)[SP_HEADER_WORDS], const int* extraDev = nullptr, int* extra = nullptr)
{
	const int ranks = w->dw.shardCount;
	if (!w->spHdrHost)
	{
		HIP_TRY(hipHostMalloc((void**)&w->spHdrHost, (2 + SHARD_MAX_RANKS * SP_HEADER_WORDS) * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void**)&w->spHdrDev, w->spHdrHost, 0));
		w->spHdrHost[0] = 0;
	}
	w->spHdrSeq = (w->spHdrSeq + 1) & 0x3fffffff;
	if (w->spHdrSeq == 0) w->spHdrSeq = 1;
	LAUNCH(w, k_sp_collect_headers, 1, 64, (const int*)w->spRecv.p, strideWords, ranks, extraDev, w->spHdrDev, w->spHdrSeq);
	if (int rc = pollPublished(w, (volatile const int*)&w->spHdrHost[0], w->spHdrSeq, "exchange headers of a spatially sharded world")) return rc;
	memcpy(hdr, w->spHdrHost + 2, (size_t)ranks * SP_HEADER_WORDS * sizeof(int));
	if (extra) *extra = w->spHdrHost[1];
	return 0;
}

// A slab capacity grows when a header says it was too small and is halved again when it has been more than twice what any
// rank needed for 4 exchanges in a row (the burst of the first steps - every proxy new, tens of thousands of pairs - would otherwise size every
// later collective). Every rank reads the same headers: the capacities stay equal on all ranks.
static void spCapDecay(int* cap, int* idle, int need, int floor)
{
	// (four exchanges in a row that used less than half: down to twice the last need - a burst, all rows of a rank in the
	// first step, must not be paid for in every slab of the next thirty steps)
	if (2 * need < *cap && *cap > floor)
	{
		if (++*idle >= 4)
		{
			int c = floor;
			while (c < 2 * need) c *= 2;
			*cap = std::min(*cap, c);
			*idle = 0;
		}
	}
	else *idle = 0;
}

// E1 (mode 0, behind SynchronizeFixtures) and E4 (mode 1, behind SolveTOI). E1 is sized from the owner census every rank
// keeps of every rank - no size exchange, nothing for the host to wait for; E4 is small (the bodies TOI events advanced) and
// sized by a capacity every rank grows alike when any rank's header says it did not fit.
static int spExchangeState(b2hip_world* w, int mode)
{
	const int ranks = w->dw.shardCount;
	if (ranks < 2) return 0;
	DW& d = w->dw;
	for (int attempt = 0; attempt < 12; ++attempt)
	{
		int capB = 1, capP = 1, capT = 0;
		bool exactFit = false; // (sized from the owner census every rank keeps of every rank: nothing can overflow, no header to read)
		if (mode == 0)
		{
			int mostB = 1, mostP = 1;
			for (int r = 0; r < ranks; ++r) { mostB = std::max(mostB, w->spOwned[r]); mostP = std::max(mostP, w->spOwnedProxies[r]); }
			if (w->spFullRows) { capB = mostB; capP = mostP; exactFit = true; }
			// (lean: the capacities follow the need the headers report - also DOWN, which is why the headers are read even when
			// the slab could hold everything a rank owns)
			else { capB = std::min(w->spRowCap, mostB); capP = std::min(w->spProxyCap, mostP); }
		}
		else { capB = w->spToiBodyCap; capP = w->spToiProxyCap; capT = w->spTailCap; }
		const int proxyWords = mode == 0 ? SP_PROXY_WORDS : SP_TOI_PROXY_WORDS;
		const size_t tailAt = SP_HEADER_WORDS + (size_t)capB * SP_BODY_WORDS + (size_t)capP * proxyWords;
		const size_t words = tailAt + (size_t)capT * SP_TAIL_WORDS;
		int rc = spEnsureSlabs(w, words);
		if (rc) return rc;
		rc = spPrepareSend(w);
		if (rc) return rc;
		if (mode == 0 || w->toiSnapshotTaken)
			LAUNCH(w, k_sp_export_state, gridFor(std::max(d.nBodies, d.capMoves)), 256, d, w->spSend.p, mode, capB, capP);
		if (mode == 1)
		{
			// the contacts this rank's TOI phase created (behind the array all ranks shared when the phase began): their
			// descriptors, for the merge of the tails; header words 5 and 6 = how many, how many of them with another rank's body
			LAUNCH(w, k_sp_export_tail, gridFor(capT), 256, d, w->spSend.p + tailAt, w->spSend.p, w->spContactsBeforeToi, capT, w->toiChains ? 1 : 0, (int*)(w->spVirt.p + SP_TAIL_MAX));
		}
		rc = spAllGather(w, words);
		if (rc) return rc;
		int created = 0;
		if (mode == 1)
		{
			// what no rank could see by itself: contacts created over an ownership boundary, proxies of different ranks' events
			// that came to overlap (k_sp_tail_pairs: every rank finds the same list in the same records)
			// (the count lives behind the pairs; k_sp_export_tail has wiped it)
			LAUNCH(w, k_sp_tail_pairs, gridFor(std::max(capP * ranks, capT)), 256, d, (const int*)w->spRecv.p, words, tailAt, capB, capP, capT, w->spVirt.p, (int*)(w->spVirt.p + SP_TAIL_MAX));
			int nVirt = 0;
			int hdr[SHARD_MAX_RANKS][SP_HEADER_WORDS];
			rc = spReadHeaders(w, words, hdr, (const int*)(w->spVirt.p + SP_TAIL_MAX), &nVirt);
			if (rc) return rc;
			int needB = 0, needP = 0, needT = 0, straddle = 0;
			{
				// this rank's own phase, as its header shows it (what a read-back before the exchange would have said)
				const int* mine = hdr[d.shardRank];
				if (mine[7] & 0x40000000) return setError(B2HIP_ERR_CAPACITY, "contact array full during a TOI sub-step of a spatially sharded world");
				if (w->toiChains)
				{
					if (mine[4] > 0) w->toiGridSticky = 16;
					else if (w->toiGridSticky > 0 && !w->spToiSettled) w->toiGridSticky -= 1;
				}
				w->spToiUnsafe = mine[7] & 0x3fffffff;
				bool any = false;
				for (int r = 0; r < ranks; ++r) any = any || (hdr[r][7] & 0x3fffffff) != 0;
				// (a parallel TOI path of some rank met an order-dependent case: that rank settles its phase - serially - and
				// everybody exchanges again; nothing of this exchange has been imported)
				if (any) return 2;
			}
			for (int r = 0; r < ranks; ++r)
			{
				needB = std::max(needB, hdr[r][0]); needP = std::max(needP, hdr[r][1]); needT = std::max(needT, hdr[r][5]);
				created += hdr[r][5];
				straddle += hdr[r][6];
			}
			if (needB > capB || needP > capP || needT > capT)
			{
				// (every rank reads the same headers and grows alike)
				while (w->spToiBodyCap < needB) w->spToiBodyCap *= 2;
				while (w->spToiProxyCap < needP) w->spToiProxyCap *= 2;
				while (w->spTailCap < needT) w->spTailCap *= 2;
				continue;
			}
			if (straddle != 0 && nVirt == 0) return setError(B2HIP_ERR_INVALID, "a rank of a spatially sharded world reported a TOI contact over an ownership boundary that no descriptor shows");
			if (nVirt > 0)
			{
				// an event reached over an ownership boundary: every rank takes its phase back, the components of such pairs
				// merge as if the contact existed, and the phase runs again (the pair lies inside one rank then)
				if (nVirt > SP_TAIL_MAX) return setError(B2HIP_ERR_CAPACITY, "more than 4 096 TOI conflicts between the ranks of a spatially sharded world");
				if (w->toiSnapshotTaken) LAUNCH(w, k_toi_snapshot, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 1);
				rc = spResolve(w, nVirt);
				if (rc) return rc;
				w->spToiRedos += 1;
				return 1;
			}
			if (created > SP_TAIL_MAX) return setError(B2HIP_ERR_CAPACITY, "more than 4 096 contacts created inside one TOI phase of a spatially sharded world");
			spCapDecay(&w->spToiBodyCap, &w->spIdle[2], needB, 256);
			spCapDecay(&w->spToiProxyCap, &w->spIdle[3], needP, 512);
			spCapDecay(&w->spTailCap, &w->spIdle[4], needT, 64);
		}
		if (mode == 0 && !exactFit)
		{
			int hdr[SHARD_MAX_RANKS][SP_HEADER_WORDS];
			rc = spReadHeaders(w, words, hdr);
			if (rc) return rc;
			int needB = 0, needP = 0;
			for (int r = 0; r < ranks; ++r) { needB = std::max(needB, hdr[r][0]); needP = std::max(needP, hdr[r][1]); }
			if (needB > capB || needP > capP)
			{
				// (an export that did not fit has not marked its rows as sent: k_sp_export_state checks the capacity first)
				while (w->spRowCap < needB) w->spRowCap *= 2;
				while (w->spProxyCap < needP) w->spProxyCap *= 2;
				continue;
			}
			spCapDecay(&w->spRowCap, &w->spIdle[0], needB, 1024);
			spCapDecay(&w->spProxyCap, &w->spIdle[1], needP, 4096);
		}
		// (+ what was sent is marked as sent, now that the exchange has gone through: the lean form's k_sp_mark_sent, same launch)
		LAUNCH(w, k_sp_import_state, gridFor(std::max(std::max(capB, capP), w->spFullRows ? 1 : d.nBodies)), 256, d, (const int*)w->spRecv.p, words, capB, proxyWords, w->spFullRows ? 0 : 1, w->spSend.p);
		w->spSendWiped = w->spSend.p;
		if (created > 0)
		{
			rc = ensureCapacity(w, (size_t)w->spContactsBeforeToi + (size_t)created);
			if (rc) return rc;
			LAUNCH(w, k_sp_merge_tails, 1, 1024, w->dw, (const int*)w->spRecv.p, words, tailAt, w->spContactsBeforeToi, w->spToiOrderBefore);
			rc = spResolve(w); // (CF_FOREIGN of the merged tail; nothing straddles - the phases would have said so)
			if (rc) return rc;
			if (w->h_dstate->c.overflow & 2048) return setError(B2HIP_ERR_CAPACITY, "the merge of the TOI tails of a spatially sharded world did not fit");
		}
		return 0;
	}
	return setError(B2HIP_ERR_CAPACITY, "the TOI exchange of a spatially sharded world did not fit");
}

// E2: this rank's new pairs out, everybody's in (behind ours in the pair buffer; Counters::nPairs counts all of them)
static int spExchangePairs(b2hip_world* w, long long* totalPairs, int* straddling)
{
	const int ranks = w->dw.shardCount;
	*totalPairs = 0;
	*straddling = 0;
	if (ranks < 2)
	{
		int rc = readState(w);
		if (rc) return rc;
		*totalPairs = w->h_dstate->c.nPairs;
		return 0;
	}
	DW& d = w->dw;
	for (int attempt = 0; attempt < 12; ++attempt)
	{
		const size_t words = SP_HEADER_WORDS + (size_t)w->spPairCap * SP_PAIR_WORDS;
		int rc = spEnsureSlabs(w, words);
		if (rc) return rc;
		rc = spPrepareSend(w);
		if (rc) return rc;
		LAUNCH(w, k_sp_export_pairs, gridFor(w->spPairCap), 256, d, w->spSend.p, w->spPairCap);
		rc = spAllGather(w, words);
		if (rc) return rc;
		int hdr[SHARD_MAX_RANKS][SP_HEADER_WORDS];
		rc = spReadHeaders(w, words, hdr);
		if (rc) return rc;
		int most = 0, strad = 0;
		long long total = 0;
		bool overflowed = false;
		for (int r = 0; r < ranks; ++r)
		{
			overflowed = overflowed || (hdr[r][5] & 3) != 0;
			most = std::max(most, hdr[r][2]);
			total += hdr[r][2];
			strad += hdr[r][6];
		}
		// (creation is all or nothing, and the unsharded world's way out of a full contact array - the host grows it at the end of
		// the step and runs the update again - does not exist for a sharded one: room for every candidate pair before anything
		// is created; the contact structure is replicated, so every rank sees the same need)
		const bool needContacts = (long long)w->lastContacts + total > (long long)d.capContacts;
		if (overflowed || needContacts)
		{
			// A rank's search did not fit its pair buffer (a dense start: every proxy is new). The unsharded world recovers from
			// that (growPairBuffers: size the buffer from the true count, clear the flag, search again) and so does this one: every
			// rank reads the same headers, so all of them grow alike - room for the union - and all of them search again
			// (findNewContacts repeats on 1; the collectives stay in step). ADVICE round 4.
			w->pairCapHint = std::max(w->pairCapHint, 2 * (size_t)total + 4096);
			rc = ensureCapacity(w, (size_t)w->lastContacts + (size_t)total + 1024);
			if (rc) return rc;
			HIP_TRY(hipMemsetAsync(&w->d_state.p->c.overflow, 0, sizeof(int), w->stream));
			w->spSendWiped = nullptr; // (this slab was filled and never imported: wiped again before the next export)
			return 1;
		}
		if (most > w->spPairCap)
		{
			while (w->spPairCap < most) w->spPairCap *= 2;
			continue;
		}
		if (total > (long long)d.capPairs)
		{
			// (the union does not fit the pair buffer: every rank grows it alike and keeps its own pairs)
			w->pairCapHint = (size_t)total + 4096;
			rc = ensureCapacity(w, (size_t)w->lastContacts);
			if (rc) return rc;
		}

		w->spPairsSent += hdr[d.shardRank][2];
		*totalPairs = total;
		*straddling = strad;
		const int capNow = w->spPairCap;
		spCapDecay(&w->spPairCap, &w->spIdle[5], most, 2048);
		LAUNCH(w, k_sp_import_pairs, gridFor(capNow), 256, w->dw, (const int*)w->spRecv.p, words, capNow);
		LAUNCH(w, k_sp_import_pairs_commit, 1, 1, w->dw, (const int*)w->spRecv.p, words, capNow, w->spSend.p);
		w->spSendWiped = w->spSend.p;
		return 0;
	}
	return setError(B2HIP_ERR_CAPACITY, "the pair exchange of a spatially sharded world did not fit");
}

// who owns how much (every rank counts every rank: the hosts size E1 from it)
static int spOwnerCensus(b2hip_world* w)
{
	HIP_TRY(hipMemsetAsync(w->d_state.p->c.spBodies, 0, 2 * SHARD_MAX_RANKS * sizeof(int), w->stream));
	LAUNCH(w, k_sp_owner_census, gridFor(w->dw.nBodies), 256, w->dw);
	int rc = readState(w);
	if (rc) return rc;
	for (int r = 0; r < SHARD_MAX_RANKS; ++r) { w->spOwned[r] = w->h_dstate->c.spBodies[r]; w->spOwnedProxies[r] = w->h_dstate->c.spProxies[r]; }
	return 0;
}

// E3. CF_FOREIGN of every contact from the owner table; contacts (and joints) that join bodies of different owners make
// their components merge under the owner that holds most of the bodies, and the losers ship the content.
static int spResolve(b2hip_world* w, int nVirt)
{
	const int ranks = w->dw.shardCount;
	DW& d = w->dw;
	for (int round = 0, grown = 0; round < 4; ++round)
	{
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.nStraddle, 0, 4 * sizeof(int), w->stream)); // nStraddle, nStraddleJoints, nResolve, nMigrated
		LAUNCH(w, k_sp_flag_contacts, gridFor(d.capContacts), 256, d);
		if (d.nJoints > 0) LAUNCH(w, k_sp_flag_joints, gridFor(d.nJoints), 256, d);
		int rc = readState(w);
		if (rc) return rc;
		const Counters& c0 = w->h_dstate->c;
		if (c0.nStraddle == 0 && c0.nStraddleJoints == 0 && nVirt == 0) return 0;
		if (ranks < 2) return setError(B2HIP_ERR_INVALID, "owners other than this rank in a world of one rank");
		if (round == 3) break;
		if (c0.nStraddle > d.capStraddle)
		{
			rc = w->spStraddle.ensure((size_t)c0.nStraddle, w->stream, false, false);
			if (rc) return rc;
			d.spStraddle = w->spStraddle.p;
			d.capStraddle = (int)w->spStraddle.cap;
			// (growing the list is not a round of the resolution: ADVICE round 4)
			if (++grown > 8) return setError(B2HIP_ERR_CAPACITY, "the list of straddling contacts of a spatially sharded world keeps growing");
			round -= 1;
			continue;
		}
		// components of the replicated structure (contacts between non-static bodies, joints), then the rows of those to merge
		LAUNCH(w, k_toi_dom_init, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_toi_dom_union, gridFor(d.capContacts), 256, d);
		if (d.nJoints > 0) LAUNCH(w, k_sp_union_joints, gridFor(d.nJoints), 256, d);
		if (nVirt > 0) LAUNCH(w, k_sp_union_virtual, gridFor(nVirt), 256, d, (const int2*)w->spVirt.p, nVirt);
		LAUNCH(w, k_toi_dom_flatten, gridFor(d.nBodies), 256, d);
		HIP_TRY(hipMemsetAsync(w->d_state.p->c.spContacts, 0, 3 * SHARD_MAX_RANKS * sizeof(int), w->stream));
		LAUNCH(w, k_sp_resolve_mark, gridFor(c0.nStraddle + d.nJoints + nVirt), 256, d, (const int2*)w->spVirt.p, nVirt);
		nVirt = 0; // (merged now: the rounds after this one look at real contacts and joints only)
		LAUNCH(w, k_sp_resolve_count, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_sp_resolve_pick, gridFor(SP_RESOLVE_MAX), 256, d);
		LAUNCH(w, k_sp_content_census, gridFor(std::max(std::max(d.capContacts, d.nJoints), d.nBodies)), 256, d);
		rc = readState(w);
		if (rc) return rc;
		const Counters& c1 = w->h_dstate->c;
		if (c1.overflow & 1024) return setError(B2HIP_ERR_CAPACITY, "more than 65 536 components to merge in one resolution of a spatially sharded world");
		int capC = 1, capJ = 1, capM = 1;
		for (int r = 0; r < ranks; ++r) { capC = std::max(capC, c1.spContacts[r]); capJ = std::max(capJ, c1.spJoints[r]); capM = std::max(capM, c1.spMigBodies[r]); }
		const size_t words = SP_HEADER_WORDS + (size_t)capC * SP_CONTENT_WORDS + (size_t)capJ * SP_JOINT_WORDS + (size_t)capM * SP_BODY_WORDS;
		rc = spEnsureSlabs(w, words);
		if (rc) return rc;
		rc = spPrepareSend(w);
		if (rc) return rc;
		LAUNCH(w, k_sp_export_content, gridFor(std::max(std::max(d.capContacts, d.nJoints), d.nBodies)), 256, d, w->spSend.p, capC, capJ);
		rc = spAllGather(w, words);
		if (rc) return rc;
		LAUNCH(w, k_sp_apply_owners, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_sp_import_content, gridFor(std::max(std::max(capC, capJ), capM)), 256, d, (const int*)w->spRecv.p, words, capC, capJ);
		LAUNCH(w, k_sp_commit_owners, gridFor(d.nBodies), 256, d);
		rc = spOwnerCensus(w);
		if (rc) return rc;
		w->spMigratedTotal += w->h_dstate->c.nMigrated;
		w->spResolves += 1;
		w->spOwnersDirty = false;
		w->spOwners.clear(); // (the host's copy is stale: b2hip_get_body_owners reads the device's)
	}
	return setError(B2HIP_ERR_INVALID, "straddling contacts remain after a resolution of a spatially sharded world");
}

// Behind SolveTOI: this rank's phase is settled here (the fallbacks b2hip_step_end would run), then E4.
static int spAfterToi(b2hip_world* w)
{
	// (what b2hip_step_end does for an unsharded world's parallel TOI paths - the grid's stickiness, the second run of the
	// chains with the grid, the serial replay - happens here, before the other ranks take this rank's result: the counters
	// that decide it travel in this rank's own header, so the usual step costs no read-back of its own)
	w->spToiSettled = false;
	for (int attempt = 0; attempt < 6; ++attempt)
	{
		int rc = spExchangeState(w, 1);
		if (rc != 2) { w->toiChains = false; w->toiSpeculative = false; return rc; }
		// (whichever parallel path this rank's phase took: unsafe with toiChains false - the components - used to do nothing
		// here for six exchanges and then fail the step)
		if (w->spToiUnsafe != 0)
		{
			LAUNCH(w, k_toi_snapshot, gridFor(std::max(w->dw.nBodies, w->dw.capContacts)), 256, w->dw, 1);
			bool serial = true;
			if (w->toiChains && w->spToiUnsafe == 4 /* TOI_UNSAFE_PAIR */ && !w->toiChainsHadGrid && !w->dw.noChainCreate)
			{
				// a chain moved a proxy out of its fat AABB while the hash grid was not kept up: the chains once more, with the grid
				w->toiGridSticky = 16;
				w->toiChains = false;
				w->toiCountersFresh = false;
				rc = phaseToiSync(w);
				if (rc) return rc;
				w->toiGridRetries += 1;
				serial = false; // (its outcome comes with the next exchange's headers)
			}
			if (serial)
			{
				rc = toiSerial(w); // (toiChains = false: nothing left to be unsafe about)
				if (rc) return rc;
				w->toiFallbacks += 1;
				w->toiSyncSticky = 16;
			}
			w->spToiSettled = true;
		}
	}
	return setError(B2HIP_ERR_INVALID, "the TOI phases of a spatially sharded world do not settle");
}

static uint8_t spStripOf(const b2hip_world* w, float x)
{
	int r = 0;
	while (r + 1 < w->dw.shardCount && x >= w->spBounds[r + 1]) ++r;
	return (uint8_t)r;
}

// The owner table reaches the device (assignment, bodies created since), and whatever straddles is resolved before Collide.
static int spBeginStep(b2hip_world* w)
{
	w->spBytesStep = 0;
	if (!w->spOwnersDirty) return 0;
	const size_t nb = w->bodies.size();
	if (w->spOwners.size() != nb)
	{
		// bodies created since the table was last known here: the device's table for the old ones, the strips for the new
		std::vector<uint8_t> cur(nb, 0);
		size_t covered = 0;
		if (w->spOwners.empty())
		{
			covered = std::min(w->spUp, nb);
			if (covered) HIP_TRY(hipMemcpy(cur.data(), w->b_owner.p, covered, hipMemcpyDeviceToHost));
		}
		else
		{
			covered = std::min(w->spOwners.size(), nb);
			memcpy(cur.data(), w->spOwners.data(), covered);
		}
		for (size_t i = covered; i < nb; ++i) cur[i] = w->bodies[i].type == B2HIP_STATIC_BODY ? 0 : spStripOf(w, w->bodies[i].cx);
		w->spOwners.swap(cur);
	}
	HIP_TRY(hipMemcpyAsync(w->b_owner.p, w->spOwners.data(), nb, hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->spUp = nb;
	int rc = spOwnerCensus(w);
	if (rc) return rc;
	rc = spResolve(w); // (clears the host's copy if owners changed)
	if (rc) return rc;
	w->spOwnersDirty = false;
	return 0;
}

int b2hip_shard_tape(b2hip_world* w, int mode, b2hip_world* from)
{
	if (int rcu = checkUsable(w, "b2hip_shard_tape", true)) return rcu;
	if (mode == 2 && (!from || from == w)) return setError(B2HIP_ERR_INVALID, "replay needs the world that recorded");
	w->spTapeRecord = mode == 1;
	w->spTapeFrom = mode == 2 ? from : nullptr;
	w->spTapeCursor = 0;
	return B2HIP_OK;
}

int b2hip_set_shard_gather(b2hip_world* w, b2hip_all_gather_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_shard_gather", true)) return rcu;
	w->gatherFn = fn;
	w->gatherUser = user;
	return B2HIP_OK;
}

int b2hip_shard_spatial(b2hip_world* w, int rank, int count, const uint8_t* owners)
{
	if (int rcu = checkUsable(w, "b2hip_shard_spatial", true)) return rcu;
	if (count < 1 || count > SHARD_MAX_RANKS || rank < 0 || rank >= count) return setError(B2HIP_ERR_INVALID, "bad rank / count (at most 8 ranks)");
	if (listenerOn(w) || hasFilter(w) || w->def.sub_stepping) return setError(B2HIP_ERR_UNSUPPORTED, "contact listeners, filters and sub-stepping are not supported in a spatially sharded world");
	DEVICE_GUARD(w);
	const size_t nb = w->bodies.size();
	w->spOwners.assign(nb, 0);
	if (owners)
	{
		for (size_t i = 0; i < nb; ++i)
		{
			if (w->bodies[i].type != B2HIP_STATIC_BODY && owners[i] >= count) return setError(B2HIP_ERR_INVALID, "owner out of range");
			w->spOwners[i] = owners[i] < count ? owners[i] : 0;
		}
		for (int r = 0; r <= count; ++r) w->spBounds[r] = 0.0f;
	}
	else
	{
		// strips of equal body count along x (positions as the host knows them: the same on every rank)
		std::vector<std::pair<float, int> > xs;
		for (size_t i = 0; i < nb; ++i)
		{
			if (w->bodies[i].type == B2HIP_STATIC_BODY || w->bodies[i].dead) continue;
			pullBody(w, (int)i);
			xs.push_back(std::make_pair(w->bodies[i].cx, (int)i));
		}
		std::sort(xs.begin(), xs.end());
		w->spBounds[0] = -3.0e38f;
		for (int r = 1; r < count; ++r) w->spBounds[r] = xs.empty() ? 0.0f : xs[std::min(xs.size() - 1, xs.size() * (size_t)r / (size_t)count)].first;
		for (int r = count; r <= SHARD_MAX_RANKS; ++r) w->spBounds[r] = 3.0e38f;
		w->dw.shardCount = count;
		for (size_t k = 0; k < xs.size(); ++k) w->spOwners[(size_t)xs[k].second] = spStripOf(w, xs[k].first);
	}
	w->spatial = true;
	w->spFullRows = getenv("B2HIP_SHARD_FULL_ROWS") != nullptr && atoi(getenv("B2HIP_SHARD_FULL_ROWS")) != 0;
	w->spOwnersDirty = true;
	w->dw.shardRank = rank;
	w->dw.shardCount = count;
	w->spMigratedTotal = 0;
	w->spResolves = 0;
	w->spPairsSent = 0;
	return B2HIP_OK;
}

int b2hip_get_body_owners(b2hip_world* w, int cap, uint8_t* owners)
{
	if (int rcu = checkUsable(w, "b2hip_get_body_owners", true)) return rcu;
	if (!w->spatial || !owners) return setError(B2HIP_ERR_INVALID, "not a spatially sharded world");
	DEVICE_GUARD(w);
	const size_t nb = std::min((size_t)std::max(cap, 0), w->bodies.size());
	if (w->spOwnersDirty) { memcpy(owners, w->spOwners.data(), std::min(nb, w->spOwners.size())); return (int)nb; }
	HIP_TRY(hipMemcpy(owners, w->b_owner.p, nb, hipMemcpyDeviceToHost));
	return (int)nb;
}

int b2hip_get_own_body_states(b2hip_world* w, int cap, int32_t* ids, b2hip_body_state* out)
{
	if (int rcu = checkUsable(w, "b2hip_get_own_body_states", true)) return rcu;
	if (!w->spatial || w->spFullRows || !w->spOwnHost) return setError(B2HIP_ERR_INVALID, "not a spatially sharded world with the lean exchange");
	const int n = std::min(std::min(w->h_dstate->c.spOwnRows, (int)w->spOwnCapRows), std::max(cap, 0));
	for (int k = 0; k < n; ++k)
	{
		const int* q = w->spOwnHost + (size_t)k * 11;
		if (ids) ids[k] = q[0];
		if (out) memcpy(&out[k], q + 1, 10 * sizeof(int));
	}
	return n;
}

int b2hip_get_shard_stats(b2hip_world* w, b2hip_shard_stats* out)
{
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	memset(out, 0, sizeof(*out));
	out->rank = w->dw.shardRank;
	out->count = w->dw.shardCount;
	if (!w->spatial) return B2HIP_OK;
	out->owned_bodies = w->spOwned[w->dw.shardRank];
	out->owned_proxies = w->spOwnedProxies[w->dw.shardRank];
	out->islands_solved = w->last.nIslands;
	out->constraint_rows = w->last.nSContacts + w->last.nLContacts;
	out->migrated_bodies = w->spMigratedTotal;
	out->resolutions = w->spResolves;
	out->bytes_received_last_step = (int64_t)w->spBytesStep;
	out->pairs_sent = w->spPairsSent;
	out->toi_redos = w->spToiRedos;
	// contacts whose content this rank maintains
	if (!w->stepActive)
	{
		DEVICE_GUARD(w);
		const int n = w->lastContacts;
		if (n > 0)
		{
			std::vector<uint32_t> f((size_t)n);
			int cur = 0;
			HIP_TRY(hipMemcpy(&cur, &w->d_state.p->cur, sizeof(int), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(f.data(), w->c_flags[cur].p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
			int own = 0;
			for (int i = 0; i < n; ++i) own += (f[(size_t)i] & CF_FOREIGN) ? 0 : 1;
			out->owned_contacts = own;
		}
	}
	return B2HIP_OK;
}

int b2hip_set_contact_filter(b2hip_world* w, b2hip_should_collide_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_contact_filter", true)) return rcu;
	w->filterFn = fn;
	w->filterUser = user;
	w->dw.userFilter = hasFilter(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_set_contact_filter_batch(b2hip_world* w, b2hip_should_collide_batch_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_contact_filter_batch", true)) return rcu;
	w->filterBatchFn = fn;
	if (fn) w->filterUser = user;
	w->dw.userFilter = hasFilter(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_default_should_collide(b2hip_world* w, int fixture_a, int fixture_b)
{
	if (!w || fixture_a < 0 || fixture_b < 0 || fixture_a >= (int)w->fixtures.size() || fixture_b >= (int)w->fixtures.size())
		return setError(B2HIP_ERR_INVALID, "bad fixture id");
	// b2ContactFilter::ShouldCollide (b2WorldCallbacks.cpp:24-38)
	const HostFixture& a = w->fixtures[fixture_a];
	const HostFixture& b = w->fixtures[fixture_b];
	if (a.groupIndex == b.groupIndex && a.groupIndex != 0) return a.groupIndex > 0 ? 1 : 0;
	return ((a.maskBits & b.categoryBits) != 0 && (a.categoryBits & b.maskBits) != 0) ? 1 : 0;
}

int b2hip_set_pre_solve(b2hip_world* w, b2hip_pre_solve_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_pre_solve", true)) return rcu;
	w->preSolveFn = fn;
	w->preSolveUser = user;
	w->dw.preSolveOn = hasPreSolve(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_set_pre_solve_batch(b2hip_world* w, b2hip_pre_solve_batch_fn fn, void* user)
{
	if (int rcu = checkUsable(w, "b2hip_set_pre_solve_batch", true)) return rcu;
	w->preSolveBatchFn = fn;
	if (fn) w->preSolveUser = user;
	w->dw.preSolveOn = hasPreSolve(w) ? 1 : 0;
	return B2HIP_OK;
}

int b2hip_enable_post_solve(b2hip_world* w, int enable)
{
	if (int rcu = checkUsable(w, "b2hip_enable_post_solve", true)) return rcu;
	w->postSolveOn = enable != 0;
	w->dw.postSolveOn = enable ? 1 : 0;
	w->postSolve.clear();
	return B2HIP_OK;
}

int b2hip_get_post_solve(b2hip_world* w, int cap, b2hip_contact_impulse* out)
{
	if (!w || (cap > 0 && !out)) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = (int)w->postSolve.size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = w->postSolve[i];
	return n;
}

int b2hip_get_island_labels(b2hip_world* w, int cap, int32_t* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	const int n = std::min(cap, (int)w->bodies.size());
	if (n <= 0) return 0;
	std::vector<int> parent(n), tier(n);
	HIP_TRY(hipMemcpy(parent.data(), w->parent.p, n * sizeof(int), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(tier.data(), w->rootIsland.p, n * sizeof(int), hipMemcpyDeviceToHost));
	for (int i = 0; i < n; ++i)
	{
		const HostBody& b = w->bodies[i];
		if (b.type == B2HIP_STATIC_BODY) { out[i] = -1; continue; }
		int r = parent[i];
		out[i] = (r >= 0 && r < n && tier[r] != ROOT_NONE) ? r : -1;
	}
	return n;
}

int b2hip_get_fat_aabb(b2hip_world* w, int fixture, float out4[4])
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out4 || fixture < 0 || fixture >= (int)w->fixtures.size()) return setError(B2HIP_ERR_INVALID, "bad fixture id");
	if ((size_t)fixture >= w->upFixtures)
	{
		memcpy(out4, w->fixtures[fixture].fat, 16);
		return 0;
	}
	HIP_TRY(hipMemcpy(out4, w->p_fat.p + fixture, 16, hipMemcpyDeviceToHost));
	return 0;
}

int b2hip_get_fat_aabbs(b2hip_world* w, int first, int count, float* out4n)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || (count > 0 && !out4n) || first < 0 || count < 0 || (size_t)(first + count) > w->fixtures.size())
		return setError(B2HIP_ERR_INVALID, "bad fixture range");
	const int onDevice = std::max(0, std::min(first + count, (int)w->upFixtures) - first);
	if (onDevice > 0) HIP_TRY(hipMemcpy(out4n, w->p_fat.p + first, (size_t)onDevice * 16, hipMemcpyDeviceToHost));
	for (int i = onDevice; i < count; ++i) memcpy(out4n + 4 * (size_t)i, w->fixtures[first + i].fat, 16); // not uploaded yet
	return 0;
}

// Debug / test hook: FNV-1a over a group of device arrays, read back after a stream sync. Valid between
// phase calls (b2hip_collide ... b2hip_step_end), so two worlds can be compared phase by phase.
//   which 0: bodies (pos, pos0, vel, xf, flags)   1: contacts (ids, key, flags & 0x7f, manifold, impulses)
//         2: proxies (fat AABBs)                  3: contact impulses only
static uint64_t fnv(uint64_t h, const void* data, size_t n)
{
	const unsigned char* p = (const unsigned char*)data;
	for (size_t i = 0; i < n; ++i)
	{
		h ^= p[i];
		h *= 1099511628211ull;
	}
	return h;
}

int b2hip_debug_hash(b2hip_world* w, int which, uint64_t* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	HIP_TRY(hipStreamSynchronize(w->stream));
	DState st;
	HIP_TRY(hipMemcpy(&st, w->d_state.p, sizeof(DState), hipMemcpyDeviceToHost));
	uint64_t h = 1469598103934665603ull;
	std::vector<unsigned char> buf;
	auto pull = [&](const void* dev, size_t bytes) -> int
	{
		buf.resize(bytes);
		if (bytes == 0) return 0;
		if (hipMemcpy(buf.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
		h = fnv(h, buf.data(), bytes);
		return 0;
	};
	const size_t nb = w->upBodies, np = w->upFixtures, nc = (size_t)st.c.nContacts;
	const int cur = st.cur;
	int bad = 0;
	if (which == 0)
	{
		bad |= pull(w->b_pos.p, nb * 16); bad |= pull(w->b_pos0.p, nb * 16); bad |= pull(w->b_vel.p, nb * 16);
		bad |= pull(w->b_xf.p, nb * 16);
		std::vector<uint32_t> f(nb);
		if (nb && hipMemcpy(f.data(), w->b_flags.p, nb * 4, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
		for (size_t i = 0; i < nb; ++i) f[i] &= 0x7fu;
		h = fnv(h, f.data(), nb * 4);
	}
	else if (which == 1)
	{
		bad |= pull(w->c_ids[cur].p, nc * 16); bad |= pull(w->c_key[cur].p, nc * 8);
		std::vector<uint32_t> f(nc);
		if (nc && hipMemcpy(f.data(), w->c_flags[cur].p, nc * 4, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
		for (size_t i = 0; i < nc; ++i) f[i] &= 0x1fu;
		h = fnv(h, f.data(), nc * 4);
		bad |= pull(w->c_man0[cur].p, nc * 16); bad |= pull(w->c_man1[cur].p, nc * 16);
		bad |= pull(w->c_imp[cur].p, nc * 16); bad |= pull(w->c_man3[cur].p, nc * 16);
	}
	else if (which == 2)
	{
		bad |= pull(w->p_fat.p, np * 16);
	}
	else
	{
		bad |= pull(w->c_imp[cur].p, nc * 16);
	}
	if (bad) return setError(B2HIP_ERR_HIP, "debug hash read-back failed");
	*out = h;
	return 0;
}

// Debug hook: raw read of a device array (0 b_pos, 1 b_vel, 2 li_bodies, 3 b_force, 4 b_flags, 5 b_damp, 6 b_mass,
// 7 counters as ints) into `out` (bytes).
int b2hip_debug_read(b2hip_world* w, int which, int first, int count, void* out)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	DEVICE_GUARD(w);
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	HIP_TRY(hipStreamSynchronize(w->stream));
	const void* src = nullptr;
	size_t elem = 16;
	switch (which)
	{
	case 0: src = w->b_pos.p; break;
	case 1: src = w->b_vel.p; break;
	case 2: src = w->li_bodies.p; elem = 4; break;
	case 3: src = w->b_force.p; break;
	case 4: src = w->b_flags.p; elem = 4; break;
	case 5: src = w->b_damp.p; break;
	case 6: src = w->b_mass.p; break;
	case 7: src = w->d_state.p; elem = 4; break;
	case 8: src = w->dbgPreVel.p; break;
	case 9: src = w->dbgVel.p; break;
	case 10: src = w->dbgLi.p; elem = 4; break;
	case 11: src = w->gridBar.p; elem = 4; break;
	case 12:
	{
		// (the colour census: counter c at colorSlot(c) - the first 65 a 128-byte line apart)
		if (first < 0 || count < 0 || first + count > COLOR_SLOT_PADDED) return setError(B2HIP_ERR_INVALID, "colour census: [0, 65)");
		HIP_TRY(hipStreamSynchronize(w->stream));
		if (count > 0) HIP_TRY(hipMemcpy2D(out, sizeof(int), w->colorCount.p + colorSlot(first), COLOR_SLOT_STRIDE * sizeof(int), sizeof(int), (size_t)count, hipMemcpyDeviceToHost));
		return 0;
	}
	case 13: src = w->bodyColorMask.p; elem = 8; break;
	case 14: src = w->deg.p; elem = 4; break;
	case 15: src = w->hubList.p; elem = 4; break;
	case 16: src = w->bodyActive.p; elem = 8; break;
	case 17: src = w->b_blk1.p; elem = 4; break;
	case 18: src = w->li_ref.p; break;
	case 19: src = w->rowColor.p; elem = 4; break;
	case 20: src = w->blkRowStart.p; elem = 4; break;
	default: return setError(B2HIP_ERR_INVALID, "bad array id");
	}
	HIP_TRY(hipMemcpy(out, (const char*)src + (size_t)first * elem, (size_t)count * elem, hipMemcpyDeviceToHost));
	return 0;
}

// Debug hook (B2HIP_TRACE=1): stage labels + state hashes recorded by the last b2hip_solve.
int b2hip_debug_trace(b2hip_world* w, int index, char* label, int label_cap, uint64_t* hash)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	if (index < 0 || index >= (int)w->trace.size()) return 1;
	if (label && label_cap > 0)
	{
		strncpy(label, w->trace[index].first.c_str(), (size_t)label_cap - 1);
		label[label_cap - 1] = 0;
	}
	if (hash) *hash = w->trace[index].second;
	return 0;
}

int b2hip_get_profile(b2hip_world* w, float ms[13])
{
	if (!w || !ms) return setError(B2HIP_ERR_INVALID, "null argument");
	memcpy(ms, w->profile, sizeof(float) * 13);
	return 0;
}

int b2hip_get_counters(b2hip_world* w, b2hip_counters* out)
{
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	memset(out, 0, sizeof(*out));
	out->bodies = (int)w->bodies.size();
	out->proxies = (int)w->fixtures.size();
	out->contacts = w->last.nContacts;
	out->touching_contacts = w->last.nTouching;
	out->islands = w->last.nIslands;
	out->small_islands = w->last.nSIslands;
	out->large_islands = w->last.nLIslands;
	out->small_island_bodies = w->last.nSBodies;
	out->small_island_contacts = w->last.nSContacts;
	out->large_island_bodies = w->last.nLBodies;
	out->large_island_contacts = w->last.nLContacts;
	out->colors = w->last.nColors;
	out->moved_proxies = w->last.nMoves;
	out->new_contacts = w->last.nNewContacts;
	out->destroyed_contacts = w->last.nDestroy;
	out->solver_chunks = w->last.nChunks;
	out->pos_iterations_large = w->last.posItersLarge;
	out->overflow_flags = w->last.overflow;
	out->toi_events = w->last.nToiEvents;
	out->toi_calls = w->last.nToiCalls;
	out->toi_pending_first_pass = w->last.nToiList;
	out->toi_serial_fallbacks = w->toiFallbacks;
	out->blocks = w->last.nBlocks;
	out->cut_constraints = w->last.nCutRows;
	out->block_max_rows = w->last.blkMaxRows;
	out->partitions = w->last.partitions;
	out->block_solver_steps = w->blockSteps;
	out->free_islands = w->last.nFreeIslands;
	out->sweep_solver_steps = w->sweepSteps;
	out->hub_constraints = w->last.nHubRows;
	out->hub_fixpoint_rounds = w->last.hubRounds;
	out->hub_serial_chunks = w->last.hubSerialChunks;
	out->toi_chain_contacts = w->toiChainContacts;
	out->toi_pre_solve_reruns = w->toiPreSolveReruns;
	return 0;
}

int b2hip_get_solver_timing(b2hip_world* w, float* ms, double* algorithmic_bytes, int* constraints, int* bodies)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	if (ms) *ms = w->solverMs;
	if (algorithmic_bytes) *algorithmic_bytes = w->solverBytes;
	if (constraints) *constraints = w->solverConstraints;
	if (bodies) *bodies = w->solverBodies;
	return 0;
}

int b2hip_set_kernel_timing(b2hip_world* w, int enable)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->kernelTiming = enable;
	return 0;
}

int b2hip_set_kernel_timing_units(b2hip_world* w, long long units_a, long long units_b)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	w->ktUnitsA = units_a;
	w->ktUnitsB = units_b;
	return 0;
}

int b2hip_get_kernel_timing(b2hip_world* w, char* name, int name_cap, float* total_ms, int* launches, double* algorithmic_bytes)
{
	if (!w) return setError(B2HIP_ERR_INVALID, "null world");
	const char* n = w->ktKind == 8 ? "large-island solver family (k_large_integrate / init / velocity / rest / k_sweep_end / position / store_impulses / finalize / sleep)" : w->ktKind == 5 ? "k_collide" : w->ktKind == 6 ? "k_sync_fixtures" : w->ktKind == 7 ? "k_find_pairs_small" : w->ktKind == 4 ? "k_solve_blocks" : w->ktKind == 1 ? "k_large_velocity" : (w->ktKind == 2 ? "k_solve_small" : (w->ktKind == 3 ? (w->solverBarriers ? "k_solve_persistent" : (w->solverRows ? "k_solve_dataflow" : "k_solve_mailbox")) : ""));
	if (name && name_cap > 0)
	{
		strncpy(name, n, (size_t)name_cap - 1);
		name[name_cap - 1] = 0;
	}
	if (total_ms) *total_ms = w->ktMs;
	if (launches) *launches = w->ktLaunches;
	if (algorithmic_bytes) *algorithmic_bytes = w->ktBytes;
	return 0;
}

// (diagnostics, not part of include/b2hip.h: the device clock stamps around the mid-step census read-back, in 10 ns ticks)
int b2hip_debug_gap_clocks(b2hip_world* w, unsigned long long out[4])
{
	if (!w || !out) return setError(B2HIP_ERR_INVALID, "null argument");
	for (int k = 0; k < 4; ++k) out[k] = w->h_dstate->gapClock[k];
	return 0;
}

} // extern "C"
