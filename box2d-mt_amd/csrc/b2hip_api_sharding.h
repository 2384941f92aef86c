// b2hip_api_sharding.h - part of the ONE translation unit b2hip.hip, inside its extern "C" block: one world over the GPUs of a
// node - island-owner sharding (round 3), RCCL from inside the library, spatial ownership (round 4: the sp* exchange functions
// E1 - E4, the resolution of straddling components) and their C ABI.
// (No include guard on purpose: b2hip.hip includes it exactly once, in order - the fragments share one scope.)

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

