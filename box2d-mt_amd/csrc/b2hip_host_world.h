// b2hip_host_world.h - part of the ONE translation unit b2hip.hip (included there, nowhere else): the world behind the C ABI -
// error reporting, device arrays, the host's mirrors of bodies / fixtures / joints, struct b2hip_world with every switch the
// environment can set, the launch macros, capacity management, the proxy-id allocator, uploads of what the host edited.
// (No include guard on purpose: b2hip.hip includes it exactly once, in order - the fragments share one scope.)

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
	DevArray<int> b_wake, b_rowDirty;
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
	DevArray<RevoluteJoint> d_joints, solveSnapJoints;
	DevArray<GearRec> d_gears, solveSnapGears;
	DevArray<float4> solveSnapBody, solveSnapImp; DevArray<uint32_t> solveSnapCFlags; // k_solver_snapshot (b2d_kernels_sweep_end.h)
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
	DevArray<float4> hubDelta, warmDelta;
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
	int solidRoundsEnv = 0;      // B2HIP_SOLID_ROUNDS=1 / 2 / 4 / 8: the tile of the island build's passes over the contacts (x 256 contacts), else by the contact count
	int collideSplitEnv = -1;    // B2HIP_COLLIDE_SPLIT=0 / 1: k_collide with the TOI-order replay inside / as a launch of its own (four waves per SIMD), else by the contact count
	bool collideUniOff = false;  // B2HIP_COLLIDE_UNI=0: k_collide reads every shape record from memory (no staged pair of records)
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
	bool forceOnDevice = true;   // some body's force / torque row on the device may be non-zero (set by uploads, reset by a clearing read-back): the
	                             // read-back of a large world skips untouched tiles only when there is nothing to clear in them
	int rowMarks = 1;            // the read-back behind an early launch looks at marked tiles only (DW::b_rowDirty; B2HIP_NO_ROW_MARKS=1: compares every row; B2HIP_ROW_MARKS_CHECK=1: compares every row AND fails the step if an unmarked one differs)
	bool rowsWentEarly = false;  // this step's rows left behind SynchronizeFixtures (startEarlyRows): the shadow is that state
	bool sweepStamps = false;    // B2HIP_SWEEP_STAMPS=1: k_sweep_end<1> leaves its phase stamps where the block solver's go (diagnostics)
	bool bodyWarm = true;        // the warm start of a launch-per-colour solve body by body in one launch (k_large_warm; B2HIP_NO_BODY_WARM=1: a sweep of launches)
	bool restFlow = true;        // the small colours of a sweep as data flow per body in one launch (k_large_rest; B2HIP_NO_REST=1: launches / tail)
	bool noHubBuild = false;     // B2HIP_NO_HUB_BUILD=1: the hub list by k_hub_flag + scan + k_hub_fill + k_hub_order (four launches) always
	bool noHubOrder = false;     // B2HIP_NO_HUB_ORDER=1: the hub rows in contact order (round 5)
	bool hubOrderAll = false;    // B2HIP_HUB_ORDER=1: ... ordered also where every hub row is swept lane after lane (B2HIP_HUB_WIDE=0 / B2HIP_HUB_SERIAL=1: comparison runs)
	bool colorAheadOff = false;  // B2HIP_NO_COLOR_AHEAD=1: round 5's flow (the queued k_color_small returns where there is no partition, the colour count comes by copy)
	bool noCensusGrid = false;   // B2HIP_NO_CENSUS_GRID=1: every colour launch sized from the mean colour (round 5)
	int colorLanes = 256;        // lanes per workgroup of a colour launch (k_large_velocity / k_large_position; B2HIP_COLOR_LANES = 64 | 128 | 256)
	bool recoverOn = true;       // a timed-out wait between workgroups of the large-island solver is recovered from (saved state back, the
	                             // solve once more launch by launch; B2HIP_NO_RECOVER=1: the step fails as in round 5)
	int* h_solverWord = nullptr; // mapped host memory: [0] the overflow word behind the solver, [1] the publication's number (k_solver_status)
	int* d_solverWord = nullptr;
	int solverSeq = 0;
	int colorRecoveries = 0;     // steps whose colouring ran out of colours / of rounds and went on (swept in order / finished by the grid-wide rounds)
	int solverRecoveries = 0;    // solves run a second time so far (b2hip_get_counters: solver_recoveries)
	int sweepRowsMax = 60000;    // islands with joints / hubs above this many constraints run launch per colour, not k_blocks_sweep (B2HIP_SWEEP_ROWS_MAX)
	int restHub = 2;             // k_large_rest + k_sweep_end of a sweep as ONE launch (k_rest_hub; B2HIP_REST_HUB=0: two launches, 1: the velocity sweeps only)
	int restHubMaxWG = 0;        // co-resident workgroups of k_rest_hub (0: do not use it)
	int restMaxWG = 0;           // ... of k_large_rest (0: unknown)
	int restArrived = 0;         // workgroups of this step's fused launches that sweep rest rows (what the verdict of a position iteration waits for: bar[5])
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
	DState* h_pub2 = nullptr;    // ... and where k_color_small publishes the state behind the colouring (a buffer and a count of its own: pubSeq2)
	DState* d_pub2 = nullptr;
	int pubSeq2 = 0;
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
	int lastToiList = 0;      // pending impacts of the last step that took the component path (sizes the speculative launches)
	bool gridFreshLast = false; // ... and whether its pair update had left the grid fresh (what the speculative flow assumes of the next)
	bool toiSpecDomains = false, toiSpecGridAssumed = false; // this step's TOI phase was queued without the census (phaseToi) / assuming a fresh grid
	bool debugAssumeFreshGrid = false; // B2HIP_DEBUG_ASSUME_FRESH_GRID=1 (tests: the speculative component path's wrong-guess handling)
	bool noToiSpecDomains = false; // B2HIP_TOI_NO_SPEC_DOMAINS=1: always look at the census first (phaseToiSync)
	int toiDomWide = 0;       // steps for which the components' event loops get 512 lanes again (one of them met more candidate contacts than a wave has lanes)
	bool toiDomWideOnly = false; // B2HIP_TOI_DOM_WIDE=1: always (comparison)
	int toiDomainsSticky = 0; // steps for which the component-wise event loops' preparations start beside k_toi_first (phaseToiSync)
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

// The state behind k_color_small (published into the second buffer with the number the host gave it).
static int awaitColors(b2hip_world* w)
{
	if (int rc = pollPublished(w, (volatile const int*)&w->h_pub2->pubSeq, w->pubSeq2, "colour state")) return rc;
	memcpy(w->h_dstate, w->h_pub2, offsetof(DState, pubSeq));
	return 0;
}

static int awaitCensus(b2hip_world* w)
{
	// (B2HIP_TEST_POLL_DELAY_US: the host comes late to its poll - what a descheduled thread does to it now and then;
	// tests/test_gpu_recovery.py: a second publication must not have overwritten the first by then)
	static const int delayUs = getenv("B2HIP_TEST_POLL_DELAY_US") ? atoi(getenv("B2HIP_TEST_POLL_DELAY_US")) : 0;
	if (delayUs > 0) std::this_thread::sleep_for(std::chrono::microseconds(delayUs));
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
	ENS(b_flags, nb); ENS(b_wake, nb); ENS(b_rowDirty, nb); ENS(b_order, nb); ENS(orderBody, nb); ENS(bigRoots, SHARD_BIG_MAX);
	ENS(p_fat, np); ENS(p_body, np); ENS(p_shape, np); ENS(p_key, np); ENS(p_filter0, np); ENS(p_filter1, np); ENS(p_mat, np);
	ENS(b_proxyHead, nb); ENS(p_next, np);
	ENS(d_shapes, std::max<size_t>(w->shapes.size(), 1));
	ENS(d_joints, std::max<size_t>(w->joints.size(), 1));
	ENS(d_gears, std::max<size_t>(w->gears.size(), 1));
	ENS(solveSnapJoints, w->recoverOn ? std::max<size_t>(w->joints.size(), 1) : 1); ENS(solveSnapGears, w->recoverOn ? std::max<size_t>(w->gears.size(), 1) : 1);
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
	ENS(bodyClaim, nb); ENS(bodyColorMask, nb); ENS(bodyActive, nb); ENS(bodyRest, nb); ENS(b_posv, nb); ENS(dfRank, B2HIP_HAVE_VALIDATION_SOLVERS ? nb * 32 : 1); ENS(dfInbox, B2HIP_HAVE_VALIDATION_SOLVERS ? 2 * cc : 1); /* (mailbox tables of the test build's k_solve_mailbox: DF_RANKS = 32 slots per body) */ ENS(evKey, cc); ENS(evInfo, cc); ENS(uncolList, COLOR_SMALL_MAX); ENS(compactList, COLOR_SMALL_MAX); ENS(hubRowOf, cc); ENS(hubList, cc); ENS(hubDelta, cc); ENS(hubMeta, 8); ENS(hubFirst, nb); ENS(solveSnapBody, w->recoverOn ? 6 * nb : 1); ENS(solveSnapImp, w->recoverOn ? cc : 1); ENS(solveSnapCFlags, w->recoverOn ? cc : 1); ENS(rootPen, ROOT_PEN_SLOTS * nb); ENS(rootDone, nb); ENS(rootSleepMin, nb);
	if (w->lc.cap < (size_t)LC_WORDS * cc)
	{
		rc = w->lc.ensure((size_t)LC_WORDS * cc, s, false, false);
		if (rc) return rc;
	}
	if (w->warmDelta.cap < 4 * cc)
	{
		rc = w->warmDelta.ensure(4 * cc, s, false, false); // (per-step scratch: k_large_init -> k_large_warm)
		if (rc) return rc;
	}
	ENS(moveBuf, 2 * np + 64);
	const size_t gridSize = (size_t)nextPow2(2 * np);
	ENS(gridCount, gridSize); ENS(gridStart, gridSize + 2); ENS(gridCursor, gridSize); ENS(gridItems, np); ENS(gridFat, np); ENS(arriveTree, (size_t)ARRIVE_SITES * TREE_WORDS); ENS(largeProxies, np); ENS(largeMoves, 2 * np + 64);
	ENS(pairKey, capPairs); ENS(pairKey2, capPairs); ENS(pairProxy, capPairs); ENS(pairProxy2, capPairs);
	ENS(pairFirst, capPairs + 1); ENS(pairRank, capPairs + 2);
	const size_t maxScanN = std::max(std::max(nb + 2, gridSize + 2), std::max(cc + 2, capPairs + 2));
	const size_t radixTiles = capPairs / RADIX_TILE + 2;
	ENS(radixHist, RADIX_DIGITS * radixTiles + 2); ENS(radixHistScan, RADIX_DIGITS * radixTiles + 4);
	ENS(scanTmp, 3 * (std::max(maxScanN, RADIX_DIGITS * radixTiles) / SCAN_TILE + 8));
	ENS(scanTmp4, 3 * (maxScanN / SCAN_TILE + 8));
	ENS(scanFlags, std::max(maxScanN, RADIX_DIGITS * radixTiles) / SCAN_TILE + 8);
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
	ENS(b_blk1, nb); ENS(b_adopt, nb); ENS(b_adoptStage, 3 * nb); ENS(blkRows, (size_t)(MAX_BLOCKS + 2) * BLK_SLOT); ENS(blkRowStart, MAX_BLOCKS + 2); ENS(blkCursor, (size_t)(MAX_BLOCKS + 2) * BLK_SLOT); ENS(blkBodyCount, (size_t)(MAX_BLOCKS + 2) * BLK_SLOT); ENS(blkBodyCursor, (size_t)(MAX_BLOCKS + 2) * BLK_SLOT);
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
	d.testMaxColors = getenv("B2HIP_TEST_MAX_COLORS") ? atoi(getenv("B2HIP_TEST_MAX_COLORS")) : 0;
	d.testColorRounds = getenv("B2HIP_TEST_COLOR_ROUNDS") ? atoi(getenv("B2HIP_TEST_COLOR_ROUNDS")) : 0;
	d.testSpinMax = getenv("B2HIP_TEST_SPIN_MAX") ? std::max(0, atoi(getenv("B2HIP_TEST_SPIN_MAX"))) : 0;
	d.restPoll = getenv("B2HIP_REST_POLL") ? std::max(1, std::min(16, atoi(getenv("B2HIP_REST_POLL")))) : 1;
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
	w->sweepStamps = getenv("B2HIP_SWEEP_STAMPS") != nullptr;
	w->rowMarks = (getenv("B2HIP_ROW_MARKS_CHECK") && atoi(getenv("B2HIP_ROW_MARKS_CHECK"))) ? 2 : (getenv("B2HIP_NO_ROW_MARKS") && atoi(getenv("B2HIP_NO_ROW_MARKS"))) ? 0 : 1;
	w->bodyWarm = !(getenv("B2HIP_NO_BODY_WARM") && atoi(getenv("B2HIP_NO_BODY_WARM")));
	w->restFlow = w->sweepEnd && !(getenv("B2HIP_NO_REST") && atoi(getenv("B2HIP_NO_REST")));
	w->noHubBuild = getenv("B2HIP_NO_HUB_BUILD") && atoi(getenv("B2HIP_NO_HUB_BUILD"));
	w->noHubOrder = getenv("B2HIP_NO_HUB_ORDER") && atoi(getenv("B2HIP_NO_HUB_ORDER"));
	w->hubOrderAll = getenv("B2HIP_HUB_ORDER") && atoi(getenv("B2HIP_HUB_ORDER"));
	w->colorAheadOff = getenv("B2HIP_NO_COLOR_AHEAD") && atoi(getenv("B2HIP_NO_COLOR_AHEAD"));
	w->noCensusGrid = getenv("B2HIP_NO_CENSUS_GRID") && atoi(getenv("B2HIP_NO_CENSUS_GRID"));
	if (const char* e = getenv("B2HIP_COLOR_LANES")) { const int v = atoi(e); w->colorLanes = v == 64 ? 64 : (v == 128 ? 128 : 256); }
	if (getenv("B2HIP_SWEEP_ROWS_MAX")) w->sweepRowsMax = std::max(1, atoi(getenv("B2HIP_SWEEP_ROWS_MAX")));
	w->restHub = getenv("B2HIP_REST_HUB") ? std::max(0, std::min(2, atoi(getenv("B2HIP_REST_HUB")))) : 2;
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
	d.b_damp = w->b_damp.p; d.b_force = w->b_force.p; d.b_flags = w->b_flags.p; d.b_wake = w->b_wake.p; d.b_rowDirty = w->b_rowDirty.p;
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
	d.joints = w->d_joints.p; d.solveSnapJoints = w->solveSnapJoints.p; d.solveSnapGears = w->solveSnapGears.p; d.solveSnapBody = w->solveSnapBody.p; d.solveSnapImp = w->solveSnapImp.p; d.solveSnapCFlags = w->solveSnapCFlags.p;
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
	d.bodyClaim = w->bodyClaim.p; d.bodyColorMask = w->bodyColorMask.p; d.bodyActive = w->bodyActive.p; d.bodyRest = w->bodyRest.p; d.b_posv = w->b_posv.p; d.dfRank = w->dfRank.p; d.dfInbox = w->dfInbox.p; d.evKey = w->evKey.p; d.evInfo = w->evInfo.p; d.eventsOn = w->eventsOn ? 1 : 0; d.uncolList = w->uncolList.p; d.compactList = w->compactList.p; d.hubRowOf = w->hubRowOf.p; d.hubList = w->hubList.p; d.hubDelta = w->hubDelta.p; d.hubMeta = w->hubMeta.p; d.hubFirst = w->hubFirst.p; d.lc = w->lc.p; d.warmDelta = w->warmDelta.p; d.rootPen = w->rootPen.p;
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
			if (b.fx != 0.0f || b.fy != 0.0f || b.torque != 0.0f) w->forceOnDevice = true;
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

