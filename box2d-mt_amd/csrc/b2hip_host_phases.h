// b2hip_host_phases.h - part of the ONE translation unit b2hip.hip: the phases of a step as launch sequences on the world's
// stream - pair update (findNewContacts), Collide, Solve (island build, census, colouring, the solver tiers: phaseSolve),
// SynchronizeFixtures, SolveTOI with its parallel paths and fallbacks, the read-back (downloadState).
// (No include guard on purpose: b2hip.hip includes it exactly once, in order - the fragments share one scope.)

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

// The passes of a stable LSD sort over bits [first, first + bits) of the keys: (shift, width) pairs of at most RADIX_BITS bits
// each, as few as cover the range, of (almost) equal width.
static void radixPasses(int first, int bits, std::vector<std::pair<int, int>>& out)
{
	const int passes = (bits + RADIX_BITS - 1) / RADIX_BITS;
	for (int p = 0, at = 0; p < passes; ++p)
	{
		const int width = (bits - at + (passes - p) - 1) / (passes - p);
		out.push_back(std::make_pair(first + at, width));
		at += width;
	}
}

// keys / payloads in (kin, vin), *nPtr of them; sorted by the bits the passes name; which buffers hold the result comes back in kin / vin
static int radixSort(b2hip_world* w, uint64_t*& kin, uint64_t*& kout, int2*& vin, int2*& vout, const int* nPtr, int tilesCap, const std::vector<std::pair<int, int>>& passes, int overflowBit = 0)
{
	DW& d = w->dw;
	// (the length of the histogram matrix depends on the count only: once per sort, not once per pass)
	LAUNCH(w, k_radix_count, 1, 1, nPtr, 0, w->consts.p + 2, tilesCap, overflowBit ? &d.st->c.overflow : (int*)nullptr, overflowBit);
	for (size_t p = 0; p < passes.size(); ++p)
	{
		LAUNCH(w, k_radix_hist, tilesCap, RADIX_THREADS, kin, d.radixHist, nPtr, 0, passes[p].first, passes[p].second, tilesCap);
		deviceExclusiveScan<int>(w->stream, d.radixHist, w->radixHistScan.p, d.scanTmp, w->scanCtx, w->consts.p + 2, RADIX_DIGITS * tilesCap);
		LAUNCH(w, k_radix_scatter, tilesCap, RADIX_THREADS, kin, vin, kout, vout, w->radixHistScan.p, nPtr, 0, passes[p].first, passes[p].second);
		std::swap(kin, kout);
		std::swap(vin, vout);
	}
	return 0;
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
		std::vector<std::pair<int, int>> passes;
		radixPasses(0, bits, passes);
		radixPasses(32, bits, passes);
		uint64_t* kin = d.pairKey;
		uint64_t* kout = d.pairKey2;
		int2* vin = d.pairProxy;
		int2* vout = d.pairProxy2;
		int tilesCap = d.capPairs / RADIX_TILE + 1;
		if (knownPairs >= 0) tilesCap = (int)std::min<long long>(tilesCap, knownPairs / RADIX_TILE + 2);
		if (int rcs = radixSort(w, kin, kout, vin, vout, &d.st->c.nPairs, tilesCap, passes)) return rcs;
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
	// (Not looking at all - the radix passes queued behind the search for twice the last update's count, checking the size
	// themselves - was built and measured in round 5, same box, same states: Tumbler 3.49 against 3.50 ms, Pyramid 316 1.37
	// against 1.39, and the 1 M field 2.6 against 2.2: its updates alternate between 3 000 and 150 000 pairs, every second guess
	// was too small and the update ran twice. The gaps a kernel trace shows behind this read-back are the profiler's.)
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
	// (the form of k_collide, b2d_kernels_collide.h: from 262 144 contacts on - a thousand workgroups - the replay of the dying
	// TOI candidates is a launch of its own and the evaluation runs four waves per SIMD; below, the launch costs more than the
	// occupancy gives)
	const bool split = w->collideSplitEnv < 0 ? w->last.nContacts >= 262144 : w->collideSplitEnv != 0;
	return runSegment(w, w->segCollide, 1 + (w->dw.preSolveOn ? 32 : 0) + (split ? 64 : 0), [w, split]() -> int
	{
		DW& d = w->dw;
		if (int rk = ktBracket(w, 2, 5)) return rk;
		// (shape records staged through LDS where the world holds many distinct ones: b2d_kernels_collide.h)
		// Measured (tools/gpu_collide_variants.py, profiles/r04_collide_variants.txt): on the 1 M-body field (a record per body,
		// circles / boxes / n-gons mixed) staging + sorting a tile by shape-pair class 133 -> 124 us; on the 100 000-box Tumbler
		// (one record, one class) the sort costs 2 % - so both follow the number of distinct records unless the environment says otherwise.
		const bool many = w->shapes.size() > 4096;
		const bool stage = w->collideStage < 0 ? many : w->collideStage != 0;
		// (round 6, measured and left out: sorting also by "the old manifold had points" - the pairs that will run the clipping - to
		// give the one touching contact in six of a dense pile waves of its own: Tumbler 316 3.275 -> 3.338 ms, 1 M field 2.10 -> 2.14)
		const int sort = w->collideSortEnv < 0 ? (many ? 1 : 0) : (w->collideSortEnv != 0 ? 1 : 0);
		// (bit 2: the records of a workgroup's first contact staged in LDS, two staged 4-gons evaluated with unrolled loops -
		// b2d_kernels_collide.h; Tumbler 316: 2.95 -> 2.88 ms. B2HIP_COLLIDE_UNI=0: off, the comparison form of the tests)
		const int uni = w->collideUniOff ? 0 : 4;
		if (stage) LAUNCH(w, (k_collide<1, true>), gridFor(d.capContacts), 256, d, sort);
		else if (split)
		{
			LAUNCH(w, (k_collide<0, false>), gridFor(d.capContacts), 256, d, sort | uni);
			LAUNCH(w, k_toi_order_destroy, 1, 256, d);
		}
		else LAUNCH(w, (k_collide<0, true>), gridFor(d.capContacts), 256, d, sort | uni);
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
	std::vector<std::pair<int, int>> passes;
	const int idBits = radixBits(d.nBodies + 1);
	radixPasses(0, idBits, passes);
	radixPasses(32, 32, passes);
	const int tilesCap = std::min(d.capPairs / RADIX_TILE + 1, d.nBodies / RADIX_TILE + 2); // (at most every body)
	const int* nPtr = &d.st->c.nLBodies;
	if (int rcs = radixSort(w, kin, kout, vin, vout, nPtr, tilesCap, passes)) return rcs;
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
	// (the passes over the contacts gather the solid ones of a tile of `solidRounds` x 256 contacts in LDS, b2d_kernels_island.h:
	// tiles as large as leave a thousand workgroups with one each - a lane per contact below 512 000 contacts)
	int solidRounds = 1;
	while (solidRounds < SOLID_TILE_ROUNDS_MAX && (long long)w->last.nContacts >= 2ll * solidRounds * 256 * 1024) solidRounds *= 2;
	if (w->solidRoundsEnv > 0) solidRounds = w->solidRoundsEnv;
	const int solidIdx = solidRounds >= 8 ? 3 : (solidRounds >= 4 ? 2 : (solidRounds >= 2 ? 1 : 0));
	int rc = runSegment(w, w->segIslands, (2 + 16ull * (uint64_t)forceLarge + (largeHint ? 8ull : 0ull) + 64ull * (uint64_t)pubBy + (adopt ? 512ull : 0ull) + 1024ull * (uint64_t)solidIdx) ^ (spHash << 12), [w, forceLarge, sp, largeHint, pubBy, adopt, solidRounds]() -> int
	{
		DW& d = w->dw;
		LAUNCH(w, k_island_init, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_island_union, gridFor((size_t)(d.capContacts + solidRounds - 1) / solidRounds), 256, d, solidRounds);
		LAUNCH(w, k_island_flatten, gridFor(d.nBodies), 256, d);
		LAUNCH(w, k_island_count, gridFor((size_t)(d.capContacts + solidRounds - 1) / solidRounds), 256, d, solidRounds);
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
		LAUNCH(w, k_island_edges, gridFor((size_t)(d.capContacts + solidRounds - 1) / solidRounds), 256, d, pubBy == 2 ? w->d_pub : (DState*)nullptr, solidRounds);
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
	bool colorAheadPublished = false; // the queued k_color_small ran in its no-partition mode and publishes the state behind it
	int aheadMinRows = 0;
	if (poll)
	{
		w->pubSeq = (w->pubSeq + 1) & 0x3fffffff; // (the device counts its publications the same way: DState::pubCount)
		// what the host would launch next in the usual case (a few new contacts on a settled pile to colour, a colour class to
		// compact) goes behind the census at once and runs while the host is busy with it; the kernel looks at the same
		// counters and returns if the case is another one
		// (round 6: ... and where no partition can be made this step - the large islands are beyond what the block solvers take,
		// the host's own hysteresis below - the queued launch runs too and PUBLISHES what it leaves: the launch-per-colour path
		// polls that instead of a copy + synchronise)
		aheadMinRows = (forceLarge != 2 && !w->colorAheadOff) ? (w->noBlocks ? 1 : (w->blocksTooBig ? 650 * std::max(w->blocksMaxWG, w->sweepMaxWG[2]) : 0)) : 0;
		if (largeHint && forceLarge != 2) { LAUNCH(w, k_color_small, 1, 1024, d, 1, aheadMinRows, w->d_pub2, (w->pubSeq2 + 1) & 0x3fffffff); colorSmallQueued = true; }
		rc = awaitCensus(w);
		if (rc == 0)
		{
			const bool settled = b2dPartitionSettled(w->h_dstate->c);
			colorAheadPublished = colorSmallQueued && b2dColorAheadNoPartition(w->h_dstate->c, aheadMinRows);
			if (colorAheadPublished) w->pubSeq2 = (w->pubSeq2 + 1) & 0x3fffffff; // (its publication is on its way, with this number)
			if (!settled && !colorAheadPublished) colorSmallQueued = false; // (it saw the same and returned)
		}
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
	w->adoptPasses = w->adoptSticky > 0 && !(w->blocksTooBig && c.nBlocks == 0); // (no partition, none to come: nobody has a block to hand on)
	const bool plainIslands = d.nJoints == 0 && c.maxDegree <= HUB_DEGREE;
	// ---- large islands that no block solver can take: more constraints than the blocks that fit the device together hold
	// (1024-lane blocks of ~750 rows: ~190 000; the settled 100 000-box Tumbler has 350 000). They run launch per colour
	// whatever happens - and a partition would only cost them: it splits the colours into two ranges (interior / cut), 27
	// colours in use where one range needs 18, and every colour is a launch of every sweep (Tumbler: 6.1 -> 5.2 ms per step).
	// So the partition is dissolved (all constraints are one class again, coloured afresh once) until the islands have shrunk.
	{
		const int cap1024 = plainIslands ? w->blocksMaxWG : w->sweepMaxWG[2];
		const bool was = w->blocksTooBig;
		// (round 6: islands with joints / hubs leave the block sweeps much earlier - k_blocks_sweep is a launch per sweep whose
		// cut constraints go from workgroup to workgroup through memory, ~45 us per sweep with 256-lane blocks but ~78 us with the
		// 1024-lane blocks an island above ~60 000 constraints needs, while launch per colour with the top colours as rest
		// rows and the hub in the same launch has come down to ~70 us there: the Tumbler with 10 000 boxes 1.19 ms per step in
		// blocks against 1.45 without, with 22 500 boxes 1.92 against 1.61 (profiles/r06_sweep_blocks_crossover.txt))
		const long long tooBigAt = plainIslands ? 800ll * cap1024 : std::min<long long>(800ll * cap1024, w->sweepRowsMax);
		const long long fitsAgainAt = plainIslands ? 650ll * cap1024 : std::min<long long>(650ll * cap1024, (long long)w->sweepRowsMax * 13 / 16);
		if (!w->blocksTooBig && cap1024 > 0 && c.nLContacts > tooBigAt) w->blocksTooBig = true;
		else if (w->blocksTooBig && c.nLContacts < fitsAgainAt) w->blocksTooBig = false;
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
				// (a colouring from scratch lands anywhere between the busiest plain body's degree + 1 and + 8: the measure is the
				// better of what the last one reached and degree + 3, so that a poor one is not kept as the yardstick)
				const int yard = std::min(w->freshColors, c.maxDegreePlain + 3);
				if (w->freshColors <= 0 || w->recolorSlack < 0 || c.nColors > yard + w->recolorSlack)
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
	bool hasHubs = !exactLarge && (c.maxDegree > HUB_DEGREE || c.nSerialOrphans > 0); // (anything for k_large_hub)
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
		bool usePersistent = !exactLarge && !hasJoints && !hasHubs && !w->debugTrace && !w->kernelTimingLaunches &&
			persistMaxWG > 0 && persistWG <= persistMaxWG;
		// the block solver (bodies in LDS, one workgroup per block of the partition): whenever the partition fits
		const bool partitionFits = blockShape && c.nBlocks > 0 && c.nOrphanRows == 0 && c.blkMaxRows <= c.blkLanes && c.blkMaxBodies <= c.blkLanes;
		bool useBlocks = usePersistent && partitionFits && plainIslands && !w->solverBarriers && !w->solverRows && !w->solverMailbox &&
			c.nBlocks <= w->blocksMaxWG &&
			(sp.velIters + 2) * (MAX_COLORS + 1) < 65536 && (sp.posIters + 1) * (MAX_COLORS + 1) < 65536;
		// one launch per sweep over the same blocks for islands that need joint walks / hub sweeps in between
		const int sweepMaxWG = c.blkLanes == 512 ? w->sweepMaxWG[1] : (c.blkLanes == BLOCK_LANES ? w->sweepMaxWG[2] : w->sweepMaxWG[0]);
		bool useSweep = !useBlocks && !exactLarge && partitionFits && !plainIslands && !w->debugTrace && !w->kernelTimingLaunches &&
			c.nBlocks <= sweepMaxWG;
		d.blockSort = (useBlocks || useSweep) ? 1 : 0;
		bool useResident = usePersistent && (useBlocks || B2HIP_HAVE_VALIDATION_SOLVERS);
		bool colorsOnDevice = false;
		bool censusVoid = false;
		bool colorSpill = false; // some constraint found no free colour this step: it sits in the hub group and is swept in order
		bool roundsToo = false;  // the grid-wide colouring rounds run (asked for by the census, or to finish what k_color_small left)
		if (!exactLarge && (c.needRecolor || c.nUncolored > 0 || c.nCompact > 0))
		{
			if (!c.needRecolor && c.nUncolored <= COLOR_SMALL_MAX)
			{
				// the usual case (a few new contacts on a settled island, a colour class to compact): one workgroup colours
				// them; the resident solver reads the colour count from the device, the launch-per-colour path reads it back
				// (the launch-per-colour path needs the colour count: published by the kernel and polled - a copy + synchronise
				// otherwise, B2HIP_NO_CENSUS_POLL=1)
				const bool pubColors = poll && !useResident && !w->colorAheadOff;
				if (!colorSmallQueued)
				{
					if (pubColors) w->pubSeq2 = (w->pubSeq2 + 1) & 0x3fffffff;
					LAUNCH(w, k_color_small, 1, 1024, d, 0, 0, pubColors ? w->d_pub2 : (DState*)nullptr, w->pubSeq2); // (else: it went out behind the census)
				}
				if (useResident)
				{
					colorsOnDevice = true;
					w->colorSmallPending = true;
				}
				else
				{
					if ((colorSmallQueued && colorAheadPublished) || (!colorSmallQueued && pubColors))
					{
						rc = awaitColors(w);
						if (rc == 0) c.nColors = w->h_dstate->c.nColors;
						if (rc == 0) memcpy(c.colorRows, w->h_dstate->c.colorRows, sizeof(c.colorRows));
					}
					else rc = readState(w);
					if (rc) return rc;
					nColors = w->h_dstate->c.nColors;
					// (round 6: neither ends the step any more. A constraint without a free colour on its two bodies is swept in order
					// with the hub rows - k_hub_flag; rounds that did not converge are finished by the grid-wide rounds below)
					if (w->h_dstate->c.overflow & 4) { if (!w->recoverOn) return setError(B2HIP_ERR_CAPACITY, "more than 64 constraint colours on one body"); colorSpill = true; }
					if (w->h_dstate->c.nUncolored != 0)
					{
						if (!w->recoverOn) return setError(B2HIP_ERR_CAPACITY, "incremental colouring did not converge");
						roundsToo = true;
						c.nUncolored = w->h_dstate->c.nUncolored;
						w->colorRecoveries += 1;
						if (getenv("B2HIP_TRACE_RECOVERY")) fprintf(stderr, "[b2hip] k_color_small left %d constraints without a colour: the grid-wide rounds finish\n", c.nUncolored);
					}
				}
			}
			else roundsToo = true;
			if (roundsToo)
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
				if (w->h_dstate->c.overflow & 4) { if (!w->recoverOn) return setError(B2HIP_ERR_CAPACITY, "more than 64 constraint colours on one body"); colorSpill = true; }
				batch = 8;
			}
			if (w->freshColorsPending) { w->freshColors = nColors < 63 ? nColors : 63; w->freshColorsPending = false; }
			}
		}
		if (colorSpill) w->colorRecoveries += 1;
		// (a constraint without a colour carries HUB_COLOR: the colours to launch end below it)
		if (!exactLarge && nColors > HUB_COLOR) nColors = HUB_COLOR;
		// ---- the row layout and the solver proper, as ONE unit that can run a second time in its SAFE form (round 6, VERDICT r05
		// item 5): b2World::Step has no failure path (b2World.cpp:1613-1710), and a wait between workgroups that times out - a
		// co-tenant on the device, a workgroup that was not resident - used to fail the step and leave a dead world. The unit
		// saves what the solver changes (k_solver_snapshot: rows of the large islands' bodies, impulses and flags of their
		// contacts, their joints) before its first solver launch; behind its last one a one-lane kernel publishes the overflow
		// word, the host looks at it (a poll of mapped memory: ~5 us with the next launch's latency - only where the unit used a
		// kernel whose workgroups wait for one another), and on a timed-out wait puts the saved state back and runs the unit
		// again on the plain path: rows by colour, a launch per colour, tail colours and hub rows in k_sweep_end's ONE workgroup -
		// nothing in it waits for another workgroup. Same colouring, same arithmetic, same order on every body: the bits the
		// fast path would have produced (tests/test_gpu_recovery.py).
		bool waitsBetweenWorkgroups = false;
		auto runLarge = [&](const bool safe) -> int
		{
		if (safe)
		{
			usePersistent = useBlocks = useSweep = useResident = false;
			colorsOnDevice = false;
			d.blockSort = 0;
		}
		if (colorSpill) hasHubs = true; // (k_hub_flag lists the constraints that found no colour)
		if (!d.blockSort) LAUNCH(w, k_color_scan, 1, 64, d); // (segments by colour: the block solvers sort their rows themselves)
		// ---- the end of every sweep without a launch per colour (b2d_kernels_sweep_end.h). The REST colours - from the highest
		// colour down while this step's colour census (k_color_check, published with the island census) keeps them below
		// restRowsMax rows together - are swept by ONE launch of k_large_rest, as data flow per body; k_color_fill notes them
		// on their bodies. Which colours are "rest" changes nothing in the result (the order on every body is the launches'):
		// a launch-count matter. Not on a step that colours afresh: its census is void.
		const bool useSweepEnd = w->sweepEnd && !exactLarge && !w->debugTrace;
		int restFirst = 0x7fffffff; // (none)
		if (useSweepEnd && w->restFlow && !safe && !d.blockSort && !c.needRecolor && !censusVoid && nColors > 0 && nColors < MAX_COLORS)
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
		// (round 6: one single-workgroup launch sorts the hub group's segment of the row array in place - k_hub_build - where the
		// last step's list was short enough for its keys to sit in LDS; the four launches below otherwise)
		const bool hubBuildOne = hasHubs && !w->noHubBuild && w->last.nHubRows <= HUB_BUILD_LDS;
		if (hubBuildOne)
		{
			LAUNCH(w, k_hub_build, 1, 1024, d, d.hubWide ? 1 : 0, (!w->noHubOrder && (d.hubWide || w->hubOrderAll)) ? 1 : 0, d.blockSort ? 1 : 0);
			w->hubSteps += 1;
		}
		else if (hasHubs)
		{
			// the hub constraints in contact-index order (deterministic whatever the atomics of k_color_fill did)
			LAUNCH(w, k_hub_flag, gridFor(d.capContacts), 256, d);
			deviceExclusiveScan<int>(w->stream, d.keepFlag, d.keepScan, d.scanTmp, w->scanCtx, &d.st->c.nContacts, d.capContacts);
			LAUNCH(w, k_hub_fill, gridFor(d.capContacts), 256, d);
			// (the hub meets its partners in the order of their highest colour: b2d_kernels_solve_large.h, k_hub_order)
			if (!w->noHubOrder && (d.hubWide || w->hubOrderAll)) LAUNCH(w, k_hub_order, 1, 1024, d, d.hubWide ? 0 : 1);
			w->hubSteps += 1;
		}
		stampPhase(w, 7);
		const int gK = gridFor(std::max(nLContacts / std::max(nColors, 1), 1) * 2);
		// A colour launch is a chain of dependent loads per lane (~4.5 us whatever it holds): a lane that makes a second trip
		// of the grid-stride loop pays the chain twice - the Tumbler's five biggest colours (41 000 - 42 000 rows against a grid
		// of 39 000 lanes sized from the MEAN colour) took 8 us instead of 6. Sized per colour from this step's census now
		// (+ what k_color_small may still add), in workgroups of colorLanes lanes so that the rows spread over all CUs.
		const int kLanes = w->colorLanes;
		const bool censusGrid = !exactLarge && nColors > 0 && nColors <= MAX_COLORS && !c.needRecolor && !censusVoid && !w->noCensusGrid;
		auto gridOfColor = [&](int col) -> int
		{
			if (!censusGrid || col < 0 || col >= MAX_COLORS) return gK;
			return gridFor((size_t)c.colorRows[col] + 512, kLanes, 1 << 16);
		};
		const int gJ = gridFor(std::max(nLIslands, 1), 64, 1 << 16);
		waitsBetweenWorkgroups = useResident || useSweep || (useSweepEnd && restFirst < nColors);
		if (!safe && w->recoverOn && waitsBetweenWorkgroups) LAUNCH(w, k_solver_snapshot, gridFor(std::max(nLBodies, nLContacts)), 256, d, 0);
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
		if (w->kernelTiming == 5 && !safe) { rc = ktRecord(w); if (rc) return rc; w->ktKind = 8; w->familyLaunchesAtStart = w->launchCount; }
		const bool sortInIntegrate = hasJoints && !exactLarge && !w->debugTrace;
		LAUNCH(w, k_large_integrate, gB, 256, d, sp, sortInIntegrate ? 1 : 0);
		if (w->debugTrace) HIP_TRY(hipMemcpyAsync(w->dbgVel.p, w->b_vel.p, w->bodies.size() * 16, hipMemcpyDeviceToDevice, w->stream));
		TRACE("integrate");
		if (hasJoints && !exactLarge && !sortInIntegrate) LAUNCH(w, k_joints_sort, gJ, 64, d);
		// (the warm start body by body in one launch - k_large_warm - where the sweep would be a launch per colour: k_large_init
		// leaves the deltas for it)
		const bool bodyWarm = w->bodyWarm && w->sweepEnd && !exactLarge && !w->debugTrace && !useSweep && sp.warmStarting && nColors <= MAX_COLORS;
		LAUNCH(w, k_large_init, gC, 256, d, sp, bodyWarm ? 1 : 0);
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
			int gR = (int)((rows + COLOR_SMALL_MAX + 255) / 256);
			// (the workgroups wait for one another: no more of them than are resident together - the rest walk in strides)
			if (w->restMaxWG > 0) gR = std::min(gR, w->restMaxWG);
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
			if (mode == 0) LAUNCH(w, k_sweep_end<0>, 1, SWEEP_END_LANES, d, sp, tf, te, what, w->sweepStamps ? w->gridBar.p : (int*)nullptr);
			else if (mode == 1) LAUNCH(w, k_sweep_end<1>, 1, SWEEP_END_LANES, d, sp, tf, te, what, w->sweepStamps ? w->gridBar.p : (int*)nullptr);
			else LAUNCH(w, k_sweep_end<2>, 1, SWEEP_END_LANES, d, sp, tf, te, what, w->sweepStamps ? w->gridBar.p : (int*)nullptr);
			w->lastSweepLaunches += 1;
			return 0;
		};
		// ---- k_large_rest and k_sweep_end of one sweep as ONE launch (k_rest_hub, round 6): the rest rows in gF workgroups of
		// 512 lanes, the hub rows / joints / verdict in one more that waits for what it needs of them as tagged rows. All
		// workgroups must be resident together (restHubMaxWG: the occupancy query at world creation).
		const bool fuseHub = useRest && w->restHub && (hasHubs || hasJoints) && !leftoverApart && w->restHubMaxWG > 1;
		auto restHubLaunch = [&](int mode, int what) -> int
		{
			if ((w->dfEpoch >> 14) != w->dfWipedAt)
			{
				HIP_TRY(hipMemsetAsync(w->b_cutv.p, 0, w->b_cutv.cap * sizeof(float4), w->stream));
				HIP_TRY(hipMemsetAsync(w->b_posv.p, 0, w->b_posv.cap * sizeof(float4), w->stream));
				w->dfWipedAt = w->dfEpoch >> 14;
			}
			long long rows = 0;
			for (int col = restFirst; col < nColors && col < MAX_COLORS; ++col) rows += c.colorRows[col];
			// (the census is this step's before k_color_small handed out its colours - at most COLOR_SMALL_MAX rows more; a grid
			// that does not hold every row walks them in strides)
			int gF = (int)((rows + COLOR_SMALL_MAX + SWEEP_END_LANES - 1) / SWEEP_END_LANES);
			gF = std::max(1, std::min(gF, w->restHubMaxWG - 1));
			if (mode == 2) w->restArrived += gF; // (the verdict of a position iteration waits for them: bar[5])
			if (mode == 1) LAUNCH(w, k_rest_hub<1>, gF + 1, SWEEP_END_LANES, d, sp, restFirst, nColors, what, w->gridBar.p, w->dfEpoch, w->restArrived, w->sweepStamps ? w->gridBar.p : (int*)nullptr);
			else LAUNCH(w, k_rest_hub<2>, gF + 1, SWEEP_END_LANES, d, sp, restFirst, nColors, what, w->gridBar.p, w->dfEpoch, w->restArrived, (int*)nullptr);
			w->dfEpoch += 1;
			w->lastSweepLaunches += 1;
			return 0;
		};
		bool noTail = false;
		auto sweepEndLaunch = [&](int mode, int what) -> int
		{
			if (!what && (!tailAny || noTail)) return 0;
			const int tf = (useSweep || useRest || noTail) ? 0 : tailFirst, te = (useSweep || useRest || noTail) ? 0 : nColors;
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
			else if (bodyWarm) LAUNCH(w, k_large_warm, gB, 256, d);
			else
			{
				for (int col = 0; col < bigEnd; ++col)
					if (colorUsed(col)) LAUNCH(w, k_large_velocity, gridOfColor(col), censusGrid ? kLanes : 256, d, col, 0);
				rc = restLaunch(0);
				if (rc) return rc;
			}
			if (useSweepEnd)
			{
				// (b2Island.cpp:256-268: the joints' InitVelocityConstraints follows the contacts' warm start; the first velocity
				// iteration then begins with the joints)
				const int what0 = (hasHubs ? SE_HUB : 0) | (hasJoints ? SE_JOINTS_INIT | (sp.velIters > 0 ? SE_JOINTS_VEL : 0) : 0);
				// (after k_large_warm every colour has had its warm start: only what follows the colours is left)
				noTail = bodyWarm; // (after k_large_warm every colour has had its warm start)
				rc = sweepEndLaunch(0, what0);
				noTail = false;
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
				LAUNCH(w, k_large_velocity, gridOfColor(col), censusGrid ? kLanes : 256, d, col, 1);
				if (w->kernelTiming == 1) { rc = ktRecord(w); if (rc) return rc; }
				if (w->debugTrace) TRACE(("vel" + std::to_string(it) + "_c" + std::to_string(col)).c_str());
			}
			const int what1 = (hasHubs ? SE_HUB | (it > 0 ? SE_GUESS : 0) : 0) | (hasJoints && it + 1 < sp.velIters ? SE_JOINTS_VEL : 0);
			if (fuseHub && !useSweep) { rc = restHubLaunch(1, what1); if (rc) return rc; }
			else
			{
				if (!useSweep) { rc = restLaunch(1); if (rc) return rc; }
				if (useSweepEnd)
				{
					rc = sweepEndLaunch(1, what1);
					if (rc) return rc;
				}
				else if (hasHubs) { rc = hubSweepLaunch(1, it > 0 ? 1 : 0); if (rc) return rc; }
			}
		}
		// (one launch for the three: k_large_after_velocity; apart where a trace wants to see each)
		const bool afterFused = !w->debugTrace && sp.posIters > 0;
		if (afterFused) LAUNCH(w, k_large_after_velocity, gridFor(std::max(nLContacts, nLBodies)), 256, d, sp);
		else
		{
			LAUNCH(w, k_large_store_impulses, gC, 256, d);
			TRACE("store_impulses");
			LAUNCH(w, k_large_integrate_positions, gB, 256, d, sp);
			TRACE("integrate_positions");
		}
		for (int it = 0; it < sp.posIters; ++it)
		{
			if ((!useSweepEnd || it == 0) && !(afterFused && it == 0)) LAUNCH(w, k_large_pos_begin, gridFor(nLIslands), 256, d);
			if (useSweep) { rc = sweep(2); if (rc) return rc; }
			else
			for (int col = 0; col < bigEnd; ++col)
			{
				if (!colorUsed(col)) continue;
				LAUNCH(w, k_large_position, gridOfColor(col), censusGrid ? kLanes : 256, d, col);
				if (w->debugTrace) TRACE(("pos" + std::to_string(it) + "_c" + std::to_string(col)).c_str());
			}
			const int what2 = (hasHubs ? SE_HUB | (it > 0 ? SE_GUESS : 0) : 0) | (hasJoints ? SE_JOINTS_POS : 0) | SE_POS_END | (it + 1 < sp.posIters ? SE_POS_BEGIN : 0);
			if (fuseHub && !useSweep && w->restHub > 1) { rc = restHubLaunch(2, what2); if (rc) return rc; continue; }
			if (!useSweep) { rc = restLaunch(2); if (rc) return rc; }
			if (useSweepEnd)
			{
				rc = sweepEndLaunch(2, what2);
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
		return 0;
		};
		rc = runLarge(false);
		if (rc) return rc;
		if (w->kernelTiming == 5 && w->ktKind == 8) { w->familyLaunches = (int)(w->launchCount - w->familyLaunchesAtStart); rc = ktRecord(w); if (rc) return rc; }
		if (w->recoverOn && waitsBetweenWorkgroups)
		{
			// did every wait come to its end? (the overflow word, published by one lane behind the solver; polled)
			w->solverSeq = (w->solverSeq + 1) & 0x7fff;
			LAUNCH(w, k_solver_status, 1, 64, d, w->d_solverWord, w->solverSeq);
			rc = pollPublished(w, (volatile const int*)&w->h_solverWord[1], w->solverSeq, "solver status");
			if (rc) return rc;
			const int flags = w->h_solverWord[0];
			// (a resident solver read the colours on the device: were they complete?)
			const bool colorsBad = colorsOnDevice && ((flags & 4) != 0 || w->h_solverWord[2] != 0);
			if (colorsOnDevice) w->colorSmallPending = false; // (looked at here)
			if ((flags & 64) || colorsBad)
			{
				if (w->tracePartition || getenv("B2HIP_TRACE_RECOVERY")) fprintf(stderr, "[b2hip] a wait between workgroups of the large-island solver timed out (overflow 0x%x): saved state back, the solve once more launch by launch\n", flags);
				LAUNCH(w, k_solver_snapshot, gridFor(std::max(nLBodies, nLContacts)), 256, d, 1);
				LAUNCH(w, k_solver_recover_reset, gridFor(d.nBodies), 256, d, w->gridBar.p);
				w->restArrived = 0;
				rc = readState(w);
				if (rc) return rc;
				if (colorsBad)
				{
					if (getenv("B2HIP_TRACE_RECOVERY")) fprintf(stderr, "[b2hip] the resident solver ran on an incomplete colouring (overflow 0x%x, %d without a colour): colouring finished, the solve once more\n", flags, w->h_solverWord[2]);
					// finish the colouring first: the grid-wide rounds over what k_color_small left open
					int uncolored = w->h_dstate->c.nUncolored;
					while (uncolored > 0)
					{
						for (int r = 0; r < 8; ++r)
						{
							LAUNCH(w, k_color_claim, gC, 256, d);
							LAUNCH(w, k_color_resolve, gC, 256, d);
						}
						rc = readState(w);
						if (rc) return rc;
						uncolored = w->h_dstate->c.nUncolored;
					}
					if (w->h_dstate->c.overflow & 4) colorSpill = true;
					censusVoid = true;
					w->colorRecoveries += 1;
				}
				nColors = w->h_dstate->c.nColors;
				if (nColors > HUB_COLOR) nColors = HUB_COLOR;
				c.nColors = nColors;
				memcpy(c.colorRows, w->h_dstate->c.colorRows, sizeof(c.colorRows));
				if (flags & 64) w->solverRecoveries += 1;
				rc = runLarge(true);
				if (rc) return rc;
			}
		}
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
static int toiBuildAdjacency(b2hip_world* w, hipStream_t s)
{
	DW& d = w->dw;
	LAUNCH_ON(w, s, k_toi_adj_clear, gridFor(d.nBodies + 1), 256, d);
	LAUNCH_ON(w, s, k_toi_adj_count, gridFor(d.capContacts), 256, d);
	deviceExclusiveScan<int>(s, d.deg, d.adjStart, d.scanTmp, w->scanCtx, w->consts.p + 4, d.nBodies + 1);
	LAUNCH_ON(w, s, k_toi_adj_fill, gridFor(d.capContacts), 256, d);
	return 0;
}

static int toiBuildIndexes(b2hip_world* w, bool csr, bool gridKnownFresh = false)
{
	DW& d = w->dw;
	if (csr)
	{
		int rc = toiBuildAdjacency(w, w->stream);
		if (rc) return rc;
	}
	// make the grid reflect every fat AABB as of now (the end-of-step pair update skips the rebuild when nothing
	// moved, and TOI moves of earlier steps never enter the move buffer) - unless this step's pair update has just built it
	// from every proxy's box and nothing moved one since (Counters::gridFresh; the kernels below check it themselves, the
	// host saves their launches when the read-back it already has says so: 128 us of a million-proxy world's step)
	// (a sharded rank launches them anyway: other ranks' boxes may have arrived since the host last looked - the kernels know)
	if (gridKnownFresh && !w->spatial) return 0;
	LAUNCH(w, k_grid_clear, gridFor(d.gridMask + 1), 256, d, 1);
	LAUNCH(w, k_grid_count, gridFor(d.nProxies), 256, d, 1);
	deviceExclusiveScan<int>(w->stream, d.gridCount, d.gridStart, d.scanTmp, w->scanCtx, w->consts.p + 1, (int)(d.gridMask + 1));
	LAUNCH(w, k_grid_fill, gridFor(d.nProxies), 256, d, 1);
	return 0;
}

static int toiSerial(b2hip_world* w)
{
	DW& d = w->dw;
	// (every caller has read this step's counters back since the pair update: phaseToiSync, the fall-backs of b2hip_step_end)
	int rc = toiBuildIndexes(w, true, w->h_dstate->c.gridFresh != 0);
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

// What the component-wise event loops need besides the first pass, on the side stream beside it (see phaseToiSync).
static inline bool toiAsideWanted(const b2hip_world* w)
{
	return w->toiDomainsSticky > 0 && w->toiCountersFresh && !w->spatial && w->stream2 != nullptr && !w->noSideStream && !w->debugSync && !w->debugTrace &&
		!w->toiSnapshotTaken && !w->toiSerialOnly && !w->toiNoDomains && !listenerOn(w) && w->dw.toiEventCap == 0 && !w->dw.toiContinue;
}
static int toiAsideLaunch(b2hip_world* w)
{
	DW& d = w->dw;
	HIP_TRY(hipEventRecord(w->evFork, w->stream));
	HIP_TRY(hipStreamWaitEvent(w->stream2, w->evFork, 0));
	// (the short launches first: the snapshot's bandwidth then falls into the tail of k_toi_first and the host's look at its
	// census - beside its start it took the first pass from 120 to 220 us: loaded latency)
	int rc = toiBuildAdjacency(w, w->stream2);
	if (rc) return rc;
	LAUNCH_ON(w, w->stream2, k_toi_dom_init, gridFor(d.nBodies), 256, d);
	LAUNCH_ON(w, w->stream2, k_toi_dom_union, gridFor(d.capContacts), 256, d);
	LAUNCH_ON(w, w->stream2, k_toi_dom_flatten, gridFor(d.nBodies), 256, d);
	LAUNCH_ON(w, w->stream2, k_toi_snapshot, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 2);
	HIP_TRY(hipEventRecord(w->evJoin, w->stream2));
	return 0;
}

static inline int toiFirstGrid(b2hip_world* w);

// The component path behind the first pass, its components and its snapshot: pending lists per component, the event loops, the
// cross-component check, what has to be taken back and replayed serially. pending: the pending impacts (the host's count, or
// its guess: every kernel strides). snapshot: 0 - taken here; 1 - taken beside the first pass, the candidates' part copied
// here; 2 - running on the side stream, waited for here.
static int toiDomainLaunches(b2hip_world* w, int pending, int snapshot)
{
	DW& d = w->dw;
	pending = std::max(pending, 1);
	LAUNCH(w, k_toi_dom_mark, gridFor(pending), 256, d);
	LAUNCH(w, k_toi_dom_count, gridFor(d.capContacts), 256, d);
	LAUNCH(w, k_toi_dom_scan, 1, 1024, d);
	LAUNCH(w, k_toi_dom_fill, gridFor(pending), 256, d);
	if (snapshot == 1) LAUNCH(w, k_toi_snap_cands, toiFirstGrid(w), 256, d);
	else if (snapshot == 2) HIP_TRY(hipStreamWaitEvent(w->stream, w->evJoin, 0));
	else LAUNCH(w, k_toi_snapshot, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 0);
	w->toiSnapshotTaken = true;
	if (w->toiDomWide > 0 || w->toiDomWideOnly)
	{
		LAUNCH(w, k_toi_domains<TOI_LANES>, std::min(pending, 2048), TOI_LANES, d, w->sp);
		if (w->toiDomWide > 0) w->toiDomWide -= 1;
	}
	else LAUNCH(w, k_toi_domains<64>, std::min(pending, 2048), 64, d, w->sp);
	LAUNCH(w, k_toi_domains_end, std::min(pending, 1024), 256, d);
	// components tied together by a new contact: back to the snapshot, then the serial loop over just those
	LAUNCH(w, k_toi_dom_rollback, gridFor(std::max(std::max(d.nBodies, d.nProxies), d.capContacts)), 256, d);
	LAUNCH(w, k_toi_loop_partial, 1, TOI_LANES, d, w->sp);
	w->toiChainsHadGrid = true; // (the components always have it)
	w->toiChains = true;
	return 0;
}

// k_toi_first takes a lane per TOI candidate (grid-stride over the manager's slot table): sized from the candidate count of
// the last read-back, generously - a wrong guess only makes the lanes loop.
static inline int toiFirstGrid(b2hip_world* w)
{
	const size_t hint = (size_t)std::max(w->h_dstate->c.nToiOrder, 0);
	return gridFor(std::min<size_t>((size_t)w->dw.capContacts, std::max<size_t>(2 * hint + 4096, 65536)));
}

// The phase without a host round trip: k_toi_first, then the chain kernels at once. Each of them leaves immediately if no
// impact is pending (or if k_toi_first saw a bullet / kinematic partner: toiUnsafe), so the host learns the outcome from
// the read-back b2hip_step_end makes anyway, and falls back there (snapshot restore + serial loop, or - when the pair
// update had overflowed its optimistic small-sort path, so this phase did not see every contact - restore, finish the
// contacts, and the synchronous phase). One read-back and ~45 us less per step with continuous physics on.
static int phaseToi(b2hip_world* w)
{
	if (w->toiSerialOnly || w->toiSyncOnly || listenerOn(w) || w->dw.toiEventCap > 0 || w->dw.toiContinue || w->spatial) return phaseToiSync(w);
	if (toiAsideWanted(w) && w->lastToiList > 0 && !w->noToiSpecDomains)
	{
		// The component path without the look at the first pass's census (round 5): while it has been the path of the last
		// steps, its launches are queued behind k_toi_first at once, sized from the last step's pending count (they stride);
		// every one of them leaves if nothing is pending. What the census would have told - the pair update unfinished, nothing
		// pending, a capacity cut - b2hip_step_end learns from the read-back it makes anyway and settles as it does for the
		// speculative chains; the one thing assumed - that the pair update of THIS step has left the grid fresh, as it had
		// the step before - is checked there too (Counters::gridFresh), and a wrong guess is a serial replay from the snapshot.
		DW& d = w->dw;
		int rc = toiAsideLaunch(w);
		if (rc) return rc;
		w->toiCountersFresh = false;
		LAUNCH(w, k_toi_first, toiFirstGrid(w), 256, d);
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.toiUnsafe, 0, sizeof(int), w->stream)); // (k_toi_first's "not chains" bit)
		HIP_TRY(hipStreamWaitEvent(w->stream, w->evJoin, 0));
		// (B2HIP_DEBUG_ASSUME_FRESH_GRID=1, for the tests: the assumption is made whatever the last step said, and the device is
		// told the grid is stale - every such step takes the wrong-guess path of b2hip_step_end)
		const bool assumeFresh = w->gridFreshLast || w->debugAssumeFreshGrid;
		if (w->debugAssumeFreshGrid) HIP_TRY(hipMemsetAsync(&w->d_state.p->c.gridFresh, 0, sizeof(int), w->stream));
		if (!assumeFresh) { rc = toiBuildIndexes(w, false, false); if (rc) return rc; }
		rc = toiDomainLaunches(w, 2 * w->lastToiList + 256, 1);
		if (rc) return rc;
		// (k_toi_clear's work is done by k_end_step, which follows)
		w->toiSpeculative = true;
		w->toiSpecDomains = true;
		w->toiSpecGridAssumed = assumeFresh;
		return 0;
	}
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
	LAUNCH(w, k_toi_first, toiFirstGrid(w), 256, d);
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
	// While the component-wise event loops have been in use (bullets: config 5), what they need besides the first pass - the
	// snapshot (230 MB for a million bodies: bandwidth), the adjacency of all contacts and the components (a dozen short
	// launches: latency) - does not wait for it: none of that reads what k_toi_first writes except the candidates' flags and
	// impact times, which k_toi_snap_cands copies afterwards. It runs on the side stream (idle behind Solve) beside
	// k_toi_first - a few heavy lanes, ~120 us - and the host's look at its census: ~180 us of the step's critical path.
	// Void if the pair update has to be finished first (the contact array grows under it): then everything is done again below.
	bool aside = false;
	if (toiAsideWanted(w))
	{
		rc = toiAsideLaunch(w);
		if (rc) return rc;
		aside = true;
	}
	bool asideValid = aside;
	for (int pass = 0; pass < 2; ++pass)
	{
		if (pass == 1 && aside)
		{
			// (the contact array is about to change, or has: the side stream's work is void - and must have ended)
			HIP_TRY(hipStreamWaitEvent(w->stream, w->evJoin, 0));
			asideValid = false;
		}
		if (pass == 1 || !w->toiCountersFresh)
		{
			// (the first pass of a step starts from the zeros of k_step_begin)
			HIP_TRY(hipMemsetAsync(&w->d_state.p->c.nToiList, 0, sizeof(int) * 5, w->stream));
			HIP_TRY(hipMemsetAsync(&w->d_state.p->c.toiUnsafe, 0, sizeof(int) * 3, w->stream));
		}
		w->toiCountersFresh = false;
		LAUNCH(w, k_toi_first, toiFirstGrid(w), 256, d);
		rc = readState(w);
		if (rc) return rc;
		if (w->h_dstate->c.overflow & 3)
		{
			// the end-of-step pair update overflowed its buffer (or the contact array): grow, run the whole update again, look again
			if (pass == 1) return setError(B2HIP_ERR_CAPACITY, "pair buffer overflow");
			if (aside && asideValid) { HIP_TRY(hipStreamWaitEvent(w->stream, w->evJoin, 0)); asideValid = false; }
			rc = growPairBuffers(w);
			if (rc) return rc;
			rc = findNewContacts(w, true);
			if (rc) return rc;
			continue;
		}
		if (pass == 1 || w->h_dstate->c.nMoves == 0 || w->h_dstate->c.nPairs <= COUNT_RANK_MAX) break;
		if (aside && asideValid) { HIP_TRY(hipStreamWaitEvent(w->stream, w->evJoin, 0)); asideValid = false; }
		rc = runSortAndCreate(w, true, w->h_dstate->c.nPairs);
		if (rc) return rc;
	}
	// (whatever follows writes what the side stream reads: it has had ~200 us, the wait is a formality)
	if (aside) HIP_TRY(hipStreamWaitEvent(w->stream, w->evJoin, 0));
	if (w->toiDomainsSticky > 0) w->toiDomainsSticky -= 1;
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
			rc = toiBuildIndexes(w, false, w->h_dstate->c.gridFresh != 0);
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
		// The snapshot (230 MB for a million bodies: bandwidth) runs on the side stream beside the dozen short launches that build
		// the adjacency and the components (latency): nothing they write is anything it reads, the stream is drained (readState
		// above), and the side stream is idle behind Solve. The event loops wait for it.
		w->toiDomainsSticky = 16;
		const bool snapAside = !asideValid && w->stream2 != nullptr && !w->noSideStream && !w->debugSync && !w->debugTrace;
		if (snapAside)
		{
			LAUNCH_ON(w, w->stream2, k_toi_snapshot, gridFor(std::max(d.nBodies, d.capContacts)), 256, d, 0);
			HIP_TRY(hipEventRecord(w->evJoin, w->stream2));
		}
		// (asideValid: snapshot, adjacency and components are there - the grid, if this step's pair update has not left one)
		rc = toiBuildIndexes(w, !asideValid, w->h_dstate->c.gridFresh != 0);
		if (rc) return rc;
		if (!asideValid)
		{
			LAUNCH(w, k_toi_dom_init, gridFor(d.nBodies), 256, d);
			LAUNCH(w, k_toi_dom_union, gridFor(d.capContacts), 256, d);
			LAUNCH(w, k_toi_dom_flatten, gridFor(d.nBodies), 256, d);
		}
		HIP_TRY(hipMemsetAsync(&w->d_state.p->c.toiUnsafe, 0, sizeof(int), w->stream)); // (k_toi_first's "not chains" bit)
		w->lastToiList = w->h_dstate->c.nToiList;
		w->gridFreshLast = w->h_dstate->c.gridFresh != 0;
		rc = toiDomainLaunches(w, w->h_dstate->c.nToiList, asideValid ? 1 : snapAside ? 2 : 0);
		if (rc) return rc;
		if (!w->spatial) LAUNCH(w, k_toi_clear, gridFor(d.nBodies), 256, d);
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
	hipLaunchKernelGGL(k_end_step, dim3(gridFor(d.nBodies)), dim3(256), 0, w->rowStream, dEarly, 0, (const int*)nullptr, w->stateOut.p, 0, END_STEP_EARLY, (float*)nullptr, 0, 0);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(w->h_state, w->stateOut.p, (size_t)d.nBodies * 10 * sizeof(float), hipMemcpyDeviceToHost, w->rowStream));
	HIP_TRY(hipEventRecord(w->rowJoin, w->rowStream));
	shadowWritten(w);
	w->rowsEarlyPending = true;
	w->rowsWentEarly = true;
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
		LAUNCH(w, k_end_step, gridFor(d.nBodies), 256, d, clear, (const int*)w->gridBar.p, w->stateOut.p, w->stateSeq, 0, (float*)nullptr, 0, 0);
		w->rowsWentEarly = false;
		if (clear) w->forceOnDevice = false;
		HIP_TRY(hipMemcpyAsync(w->h_state, w->stateOut.p, B2D_STATE_TAIL(nb) * sizeof(float) + sizeof(DState), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		memcpy(w->h_dstate, w->h_state + B2D_STATE_TAIL(nb), offsetof(DState, pubSeq));
		w->shadowDev = nullptr; // (the staging array is the shadow's memory)
		return 0;
	}
	const int rowMode = shadowValid(w) ? 2 : 1;
	// The first read-back behind this step's early launch looks only at the tiles somebody has written a row of since
	// (DW::b_rowDirty) - unless it has forces to clear in every tile, or sweeps to reset that earlier calls of a sub-stepped
	// step advanced. Any later read-back of the step (fall-backs, a finished pair update) compares every row again.
	const bool clearing = clear && w->forceOnDevice;
	if (w->traceLaunches) fprintf(stderr, "[b2hip] host: read-back: rows went early %d, row mode %d, lazy %d, clearing %d (clear %d), marks %d\n", (int)w->rowsWentEarly, rowMode, (int)lazy, (int)clearing, clear, w->rowMarks);
	const int marks = (w->rowsWentEarly && rowMode == 2 && !lazy && !w->spatial && !clearing && !d.toiContinue && d.toiEventCap == 0) ? w->rowMarks : 0;
	w->rowsWentEarly = false;
	LAUNCH(w, k_end_step, gridFor(d.nBodies), 256, d, clear, (const int*)w->gridBar.p, w->d_hstate, w->stateSeq,
		lazy ? END_STEP_LAZY : skipRowsIfRedo ? END_STEP_SKIP_IF_REDO : END_STEP_FULL, w->stateOut.p, rowMode, marks);
	if (w->traceLaunches) { fprintf(stderr, "[b2hip] host: awaiting the read-back %d\n", w->stateSeq); fflush(stderr); }
	const int rc = awaitState(w, nb);
	if (w->traceLaunches) { fprintf(stderr, "[b2hip] host: read-back %d arrived (rc %d), marked tiles %d of %d\n", w->stateSeq, rc, w->h_dstate->c.endBlocksDone, (d.nBodies + 255) / 256); fflush(stderr); }
	// (rowsSkipped: 0 - the rows were stored; the shadow of a full write is valid from here on)
	if (rc == 0 && rowMode == 1 && w->h_dstate->c.rowsSkipped == 0) shadowWritten(w);
	// (the forces are cleared row by row inside the tile loop: not when the rows were skipped)
	if (rc == 0 && clear && w->h_dstate->c.rowsSkipped != 1) w->forceOnDevice = false;
	if (rc == 0 && (w->h_dstate->c.overflow & 0x1000)) return setError(B2HIP_ERR_INVALID, "B2HIP_ROW_MARKS_CHECK: a body's read-back row changed behind the early launch without a mark (DW::b_rowDirty)");
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
	LAUNCH(w, k_end_step, gridFor(d.nBodies), 256, d, 0, (const int*)nullptr, w->d_hstate, w->stateSeq, END_STEP_ROWS, w->stateOut.p, rowMode, 0);
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

