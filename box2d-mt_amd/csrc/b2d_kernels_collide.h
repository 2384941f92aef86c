// b2d_kernels_collide.h - narrow phase: one lane per contact (b2ContactManager::Collide +
// b2Contact::UpdateImpl), then the stable compaction that plays the role of the reference's
// sorted destroy list (b2ContactManager.cpp:388-439).
#ifndef B2D_KERNELS_COLLIDE_H
#define B2D_KERNELS_COLLIDE_H

#include "b2d_world.h"
#include "b2d_toi.h"

__device__ __forceinline__ Xf loadXf(const float4* b_xf, int body)
{
	float4 t = b_xf[body];
	Xf xf;
	xf.p = v2(t.x, t.y);
	xf.q.s = t.z;
	xf.q.c = t.w;
	return xf;
}

__device__ __forceinline__ AABB loadAabb(const float4* p_fat, int proxy)
{
	float4 t = p_fat[proxy];
	AABB a;
	a.lo = v2(t.x, t.y);
	a.hi = v2(t.z, t.w);
	return a;
}

__device__ __forceinline__ bool bodyActiveForContact(uint32_t f)
{
	// b2ContactManager::IsContactActive (b2ContactManager.cpp:94-107)
	return (f & BF_AWAKE) != 0 && (f & BF_TYPE_MASK) != BT_STATIC;
}

// b2ContactFilter::ShouldCollide (b2WorldCallbacks.cpp:24-38)
__device__ __forceinline__ bool filterShouldCollide(uint32_t f0A, int f1A, uint32_t f0B, int f1B)
{
	int groupA = (int)(short)(f1A & 0xffff);
	int groupB = (int)(short)(f1B & 0xffff);
	if (groupA == groupB && groupA != 0)
	{
		return groupA > 0;
	}
	uint32_t catA = f0A & 0xffffu, maskA = f0A >> 16;
	uint32_t catB = f0B & 0xffffu, maskB = f0B >> 16;
	return (maskA & catB) != 0 && (catA & maskB) != 0;
}

// b2Body::ShouldCollide (b2Body.cpp:428-449): at least one dynamic body, and no joint between
// the two bodies with collideConnected == false.
__device__ __forceinline__ bool bodiesShouldCollide(const DW& W, int bodyA, int bodyB)
{
	uint32_t tA = W.b_flags[bodyA] & BF_TYPE_MASK;
	uint32_t tB = W.b_flags[bodyB] & BF_TYPE_MASK;
	if (tA != BT_DYNAMIC && tB != BT_DYNAMIC) return false;
	if (W.nJoints == 0) return true;
	// the joint edges of ONE of the two bodies (the reference walks bodyA's joint list too); the per-body lists are the ones
	// island building uses. A scan of every joint of the world here cost 3 ms per step with 20 000 joints.
	for (int e = W.jadjStart[bodyA], end = W.jadjStart[bodyA + 1]; e < end; ++e)
	{
		const JointRec& jn = W.joints[W.jadj[e]];
		if (jn.bodyA == bodyB || jn.bodyB == bodyB)
		{
			if (jn.collideConnected == 0) return false;
		}
	}
	return true;
}

// b2Contact::IsToiCandidate (b2Contact.cpp:300-324)
__device__ __forceinline__ bool isToiCandidate(const DW& W, int proxyA, int proxyB, int bodyA, int bodyB)
{
	const int fA = W.p_filter1[proxyA], fB = W.p_filter1[proxyB];
	if ((fA | fB) & PF_SENSOR) return false;
	const uint32_t bfA = W.b_flags[bodyA], bfB = W.b_flags[bodyB];
	if ((bfA | bfB) & BF_BULLET) return true;
	const bool includesNonDynamic = (bfA & BF_TYPE_MASK) != BT_DYNAMIC || (bfB & BF_TYPE_MASK) != BT_DYNAMIC;
	const bool neitherThick = ((fA | fB) & PF_THICK) == 0;
	return includesNonDynamic && neitherThick;
}

// The reference keeps its TOI-candidate contacts in the first m_toiCount slots of one array and removes a
// contact by moving the LAST candidate into its slot (b2ContactManager::RemoveFromContactArray, :688-714),
// destroying in (proxyLow, proxyHigh) order (FinishCollide :388-439). That slot order is the order in which
// b2World::FindMinToiContact re-synchronises sweeps, so it is mirrored here: ContactArrays::mgr is the slot,
// toiPos2c its inverse. Few candidates die per step; one lane replays the removals.
#define TOI_ORDER_SORT_MAX 2048
// Run by the last workgroup of k_collide (256 lanes; it was a launch of its own behind it). Everything the replay reads more
// than once is in LDS first - the keys (the ranking compares every pair), the slots of the dying contacts, the TAIL of the slot
// table (the only entries a removal can move) - and the one lane that replays the removals in order reads LDS only; its
// stores to the tables are fire and forget. (Round 4: with the keys fetched from memory inside the ranking loop and the
// replay chasing three dependent loads per removal, this tail was most of k_collide's 220 us on the 1 M-body field, where
// the bullets' fat AABBs make and break ~100 candidate contacts per step - not the evaluation of the manifolds.)
#define TOI_ORDER_SCRATCH_BYTES (TOI_ORDER_SORT_MAX * (8 + 4 * 4))
// More dying candidates than the LDS tables hold (a dense start: 450 bullets in a crowd break 5 000 candidate contacts in one
// step; until round 5 the surplus was flagged in Counters::overflow bit 4 and the slot table left inconsistent - harmless
// while nothing but the re-synchronisation order read it, wrong results once k_toi_first walks it): the removals are
// replayed in CHUNKS of TOI_ORDER_SORT_MAX in key order. A chunk is the replay below applied to the table as the chunks
// before left it - a tail contact that dies in a LATER chunk is moved like a survivor (its slot is read again when its turn
// comes), one that died in an earlier chunk is gone. What needs all n at once - the ranks - goes through memory: the keys and
// the contacts in key order in the pair buffer (idle between the pair updates), a contact's rank in DW::toiList (the TOI
// phase's list: idle during Collide). n * n / 256 comparisons per lane: milliseconds at n = 20 000, and only then.
__device__ __forceinline__ void toiOrderDestroy(const DW& W, void* scratch)
{
	DState* S = W.st;
	const int n = __hip_atomic_load(&S->c.nToiDestroy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (n == 0) return;
	const ContactArrays& C = W.ca[S->cur];
	uint64_t* s_key = (uint64_t*)scratch;
	int* s_sorted = (int*)(s_key + TOI_ORDER_SORT_MAX);   // dying contacts in key order
	int* s_slot = s_sorted + TOI_ORDER_SORT_MAX;          // ... their slots (kept up to date while earlier removals move them)
	int* s_tail = s_slot + TOI_ORDER_SORT_MAX;            // contact in slot base + t
	int* s_tailJ = s_tail + TOI_ORDER_SORT_MAX;           // ... its place in s_sorted if it is dying itself, else -1
	const bool chunked = n > TOI_ORDER_SORT_MAX;
	uint64_t* g_key = W.pairKey;                 // (chunked) keys of the dying contacts, list order
	int* g_sorted = (int*)(W.pairKey + n);       // (chunked) the dying contacts in key order
	int* g_rank = W.toiList;                     // (chunked) per contact: its rank among the dying
	if (chunked)
	{
		if ((size_t)n * 3 > (size_t)W.capPairs * 2 || n > W.capContacts)
		{
			// (no room for the tables: the host fails the step - b2hip_step_end)
			if (threadIdx.x == 0) { atomicOr(&S->c.overflow, 16); S->c.nToiDestroy = 0; }
			return;
		}
		for (int i = threadIdx.x; i < n; i += blockDim.x) g_key[i] = C.key[b2dLoadAgentI(&W.toiDestroyList[i])];
		__threadfence();
		__syncthreads();
		for (int i = threadIdx.x; i < n; i += blockDim.x)
		{
			const uint64_t key = g_key[i];
			int rank = 0;
			for (int j = 0; j < n; ++j) rank += g_key[j] < key ? 1 : 0; // (every lane the same word: one fetch per wave)
			const int ci = b2dLoadAgentI(&W.toiDestroyList[i]);
			g_sorted[rank] = ci;
			g_rank[ci] = rank;
		}
		__threadfence();
		__syncthreads();
	}
	int count = S->c.nToiOrder;
	for (int c0 = 0; c0 < n; c0 += TOI_ORDER_SORT_MAX)
	{
		const int m = n - c0 < TOI_ORDER_SORT_MAX ? n - c0 : TOI_ORDER_SORT_MAX;
		const int count0 = count;
		const int base = count0 - m > 0 ? count0 - m : 0;
		if (!chunked)
		{
			for (int i = threadIdx.x; i < m; i += blockDim.x) s_key[i] = C.key[b2dLoadAgentI(&W.toiDestroyList[i])];
			__syncthreads();
			// rank by key (keys are unique)
			for (int i = threadIdx.x; i < m; i += blockDim.x)
			{
				const uint64_t key = s_key[i];
				int rank = 0;
				for (int j = 0; j < m; ++j) rank += s_key[j] < key ? 1 : 0;
				const int ci = b2dLoadAgentI(&W.toiDestroyList[i]);
				s_sorted[rank] = ci;
				s_slot[rank] = C.mgr[ci];
			}
		}
		else
		{
			// (what lane 0 stored in the chunk before is read past this CU's L1: the lines may sit there from that chunk's loads)
			for (int i = threadIdx.x; i < m; i += blockDim.x)
			{
				const int ci = g_sorted[c0 + i];
				s_sorted[i] = ci;
				s_slot[i] = b2dLoadAgentI(&C.mgr[ci]);
			}
		}
		for (int t = threadIdx.x; t < count0 - base; t += blockDim.x) s_tail[t] = b2dLoadAgentI(&W.toiPos2c[base + t]);
		__syncthreads();
		for (int t = threadIdx.x; t < count0 - base; t += blockDim.x)
		{
			const int c = s_tail[t];
			int j = -1;
			// (k_collide has set CF_DESTROY on every dying contact and its stores have landed: this is the last workgroup)
			if (b2dLoadAgentI((const int*)&C.flags[c]) & (int)CF_DESTROY)
			{
				if (!chunked) { for (int q = 0; q < m; ++q) if (s_sorted[q] == c) { j = q; break; } }
				else { const int r = g_rank[c]; if (r >= c0 && r < c0 + m) j = r - c0; }
			}
			s_tailJ[t] = j;
		}
		__syncthreads();
		if (threadIdx.x == 0)
		{
			for (int k = 0; k < m; ++k)
			{
				const int ci = s_sorted[k];
				const int slot = s_slot[k];
				--count;
				const int last = s_tail[count - base], lastJ = s_tailJ[count - base];
				W.toiPos2c[slot] = last;
				C.mgr[last] = slot;
				if (slot >= base) { s_tail[slot - base] = last; s_tailJ[slot - base] = lastJ; }
				if (lastJ >= 0) s_slot[lastJ] = slot;
				C.mgr[ci] = -1;
			}
			__threadfence();
		}
		count = count0 - m; // (every lane keeps count: m removals)
		__syncthreads();
	}
	if (threadIdx.x == 0)
	{
		S->c.nToiOrder = count;
		S->c.nToiDestroy = 0;
	}
}

// What a lane of k_collide will run for contact `i`: lanes of one wave that run different things run them one after the other
// (the 1 M-body field of circles, boxes and n-gons: 223 us for 131 000 contacts, a tenth of the rate the all-boxes Tumbler gets).
// SURVEY section 2.1 asks for kernels "specialised by shape-pair type" (b2Contact.cpp:42-52: one evaluate function per pair of
// shape types): here a workgroup sorts the 256 contacts of its tile by this key in LDS, so that a wave evaluates one class -
// the contact array keeps its creation order, no list is built, no launch added.
//   0-5 polygon / polygon by the larger vertex count (3 .. 8), 6-11 polygon / circle by vertex count, 12 circle / circle,
//   13-18 edge or chain child / polygon by vertex count, 19 edge / circle, 60 sensor (GJK), 61 foreign (structure only),
//   62 not updated at all (both bodies asleep, or the fat AABBs have parted), 63 no contact here (the tile's tail)
__device__ __forceinline__ int collideClassKey(const DW& W, const ContactArrays& C, int i)
{
	const int4 ids = C.ids[i];
	const uint32_t flags = C.flags[i];
	if (flags & CF_FOREIGN) return 61;
	if (!(bodyActiveForContact(W.b_flags[ids.z]) || bodyActiveForContact(W.b_flags[ids.w]))) return 62;
	const float4 fA = W.p_fat[ids.x], fB = W.p_fat[ids.y];
	if (fB.x - fA.z > 0.0f || fB.y - fA.w > 0.0f || fA.x - fB.z > 0.0f || fA.y - fB.w > 0.0f) return 62;
	if (flags & CF_SENSOR) return 60;
	const ShapeRec* sA = W.shapes + W.p_shape[ids.x];
	const ShapeRec* sB = W.shapes + W.p_shape[ids.y];
	const int tA = sA->type == B2D_SHAPE_CHAIN ? B2D_SHAPE_EDGE : sA->type, tB = sB->type;
	const int cA = sA->count, cB = sB->count;
	auto bucket = [](int c) { return c < 3 ? 0 : (c > 8 ? 5 : c - 3); };
	if (tA == B2D_SHAPE_POLYGON && tB == B2D_SHAPE_POLYGON) return bucket(cA > cB ? cA : cB);
	if (tA == B2D_SHAPE_POLYGON && tB == B2D_SHAPE_CIRCLE) return 6 + bucket(cA);
	if (tA == B2D_SHAPE_CIRCLE && tB == B2D_SHAPE_CIRCLE) return 12;
	if (tA == B2D_SHAPE_EDGE && tB == B2D_SHAPE_POLYGON) return 13 + bucket(cB);
	if (tA == B2D_SHAPE_EDGE && tB == B2D_SHAPE_CIRCLE) return 19;
	return 59;
}

// STAGE: the two shape records of a lane's contact are copied into LDS with wide loads before the manifold is evaluated from
// them. A world whose bodies share a few shape records (a box scene: one) reads them through the caches at the kernel's full
// rate; a world where every body has a record of its own (the 1 M-body field: a million radii and n-gons) pays for every
// vertex the SAT loops touch with a load instruction whose 64 lanes hit 64 different lines - ~200 such instructions per
// wave against the 20 that fetch both records whole. The host picks the instantiation by the number of distinct records.
// (bits, not values: -0 against +0 and NaN payloads are differences too)
__device__ __forceinline__ bool b2dSameBits4(float4 a, float4 b)
{
	return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
		__float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

// ORDER: the last workgroup replays the dying TOI candidates (toiOrderDestroy) - no launch for it, but its 48 KB of LDS tables
// and the 150 registers of the evaluation leave three waves per SIMD. The kernel is bound by the chains of loads in front of
// the arithmetic (ids -> bodies, proxies, old manifold -> shapes), not by bandwidth or the arithmetic: with the replay in a
// launch of its own (k_toi_order_destroy) and the registers capped at 128 - four waves per SIMD, 15 values spilled - the
// 2.6 M contacts of the settled Tumbler take 260 us instead of 348 (the step 3.29 -> 3.20 ms; five waves with 56 spills:
// 3.25, six: 3.35). The host picks the form by the contact count (b2hip_host_phases.h: phaseCollide).
template <int STAGE, bool ORDER = true>
__global__ __launch_bounds__(256) void
__attribute__((amdgpu_waves_per_eu((!ORDER && !STAGE) ? 4 : 1, (!ORDER && !STAGE) ? 4 : 8)))
k_collide(DW W, int sortTile)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	int nDestroy = 0, nTouch = 0;
	__shared__ int s_bin[64];
	__shared__ int s_perm[256];
	// (152-byte records side by side: lanes read the same member 38 words apart - a two-way bank conflict at worst)
	// (... and, once the contacts are done, the scratch of the last workgroup's toiOrderDestroy)
	__shared__ __attribute__((aligned(16))) unsigned char s_raw[STAGE ? ((512 * sizeof(ShapeRec) > TOI_ORDER_SCRATCH_BYTES || !ORDER) ? 512 * sizeof(ShapeRec) : TOI_ORDER_SCRATCH_BYTES) : (ORDER ? TOI_ORDER_SCRATCH_BYTES : 16)];
	ShapeRec* const s_shape = (ShapeRec*)s_raw;
	// (sortTile bit 2) the shape records of the workgroup's FIRST contact wait in LDS: a world of boxes has one record, and the
	// separating-axis loops fetch its vertices and normals one by one - ~50 loads per contact, every one a round trip to the
	// vector L1 for a value all 64 lanes share, one after the other: that chain, not the bandwidth and not the arithmetic, was
	// this kernel's time on the 100 000-box Tumbler. A polygon pair whose two records are the staged ones reads LDS instead.
	__shared__ __attribute__((aligned(16))) ShapeRec s_uni[2];
	__shared__ int s_uniId[2];
	const bool uniOn = !STAGE && (sortTile & 4) != 0;
	sortTile &= 3;
	if (uniOn)
	{
		const int i0 = (int)(blockIdx.x * blockDim.x);
		int idA = 0, idB = 0;
		if (i0 < n)
		{
			const int4 ids0 = C.ids[i0];
			idA = W.p_shape[ids0.x];
			idB = W.p_shape[ids0.y];
		}
		static_assert(sizeof(ShapeRec) % 4 == 0, "ShapeRec is copied word by word");
		constexpr int RW = (int)(sizeof(ShapeRec) / 4);
		if (threadIdx.x < 2 * RW)
		{
			const int which = (int)threadIdx.x / RW, word = (int)threadIdx.x % RW;
			((uint32_t*)&s_uni[which])[word] = ((const uint32_t*)(W.shapes + (which ? idB : idA)))[word];
		}
		if (threadIdx.x == 0) { s_uniId[0] = idA; s_uniId[1] = idB; }
		__syncthreads();
	}
	for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x)
	{
		int i = base + (int)threadIdx.x;
		if (sortTile)
		{
			// the tile's contacts by class: a counting sort over 64 keys in LDS (which lane takes which contact of a class does
			// not matter: every contact is evaluated on its own)
			const int key = i < n ? collideClassKey(W, C, i) : 63;
			if (threadIdx.x < 64) s_bin[threadIdx.x] = 0;
			__syncthreads();
			const int within = atomicAdd(&s_bin[key], 1);
			__syncthreads();
			if (threadIdx.x < 64)
			{
				const int v = s_bin[threadIdx.x];
				int incl = v;
				for (int off = 1; off < 64; off <<= 1)
				{
					const int o = __shfl_up(incl, off);
					if ((int)threadIdx.x >= off) incl += o;
				}
				s_bin[threadIdx.x] = incl - v;
			}
			__syncthreads();
			s_perm[s_bin[key] + within] = (int)threadIdx.x;
			__syncthreads();
			i = base + s_perm[threadIdx.x];
		}
		if (i >= n) continue;
		int4 ids = C.ids[i];
		uint32_t flags = C.flags[i];
		if (flags & CF_FOREIGN)
		{
			// a spatially sharded world: the bodies are another rank's, which evaluates the manifold. Here the contact only keeps
			// existing - the structural half of b2ContactManager::Collide (:177-230), the same on every rank: the re-filter, the
			// "both bodies asleep" skip, the fat-AABB test (else destroy)
			int keepF = 1;
			if (flags & CF_FILTER)
			{
				const bool ok = bodiesShouldCollide(W, ids.w, ids.z) &&
					filterShouldCollide(W.p_filter0[ids.x], W.p_filter1[ids.x], W.p_filter0[ids.y], W.p_filter1[ids.y]);
				if (!ok) keepF = 0; else flags &= ~CF_FILTER;
			}
			if (keepF && (bodyActiveForContact(W.b_flags[ids.z]) || bodyActiveForContact(W.b_flags[ids.w])))
			{
				const float4 fA = W.p_fat[ids.x], fB = W.p_fat[ids.y];
				AABB boxA, boxB;
				boxA.lo = v2(fA.x, fA.y); boxA.hi = v2(fA.z, fA.w);
				boxB.lo = v2(fB.x, fB.y); boxB.hi = v2(fB.z, fB.w);
				if (!b2dAabbOverlap(boxA, boxB)) keepF = 0;
			}
			if (!keepF)
			{
				flags |= CF_DESTROY;
				++nDestroy;
				if (flags & CF_TOI_CANDIDATE) b2dStoreAgentI(&W.toiDestroyList[atomicAdd(&S->c.nToiDestroy, 1)], i);
			}
			C.flags[i] = flags;
			W.keepFlag[i] = keepF;
			continue;
		}
		// Everything the update of a live contact reads is fetched in TWO rounds - what hangs on the contact index with the
		// ids, what hangs on the ids right after - instead of level by level behind the tests that need it (flags, then fat
		// AABBs, then the old manifold and the shape indices, then shapes and transforms: five dependent round trips in front
		// of the arithmetic). A contact that turns out inactive or out of its fat AABBs has read a few rows for nothing.
		const int4 m3 = C.man3[i];
		const float4 oldImp = C.imp[i];
		const float4 o0 = C.man0[i], o1 = C.man1[i];
		const int proxyA = ids.x, proxyB = ids.y, bodyA = ids.z, bodyB = ids.w;
		uint32_t bfA = W.b_flags[bodyA], bfB = W.b_flags[bodyB];
		const float4 fatA = W.p_fat[proxyA], fatB = W.p_fat[proxyB];
		const int shapeA = W.p_shape[proxyA], shapeB = W.p_shape[proxyB];
		const Xf xfA = loadXf(W.b_xf, bodyA), xfB = loadXf(W.b_xf, bodyB);
		int keep = 1;

		if (flags & CF_FILTER)
		{
			// (with a user contact filter the host has asked it already: k_filter_list / CF_USER_REJECT)
			bool ok = bodiesShouldCollide(W, bodyB, bodyA) &&
				(W.userFilter ? (flags & CF_USER_REJECT) == 0
				              : filterShouldCollide(W.p_filter0[proxyA], W.p_filter1[proxyA], W.p_filter0[proxyB], W.p_filter1[proxyB]));
			flags &= ~CF_USER_REJECT;
			if (!ok)
			{
				keep = 0;
			}
			else
			{
				flags &= ~CF_FILTER;
			}
		}

		flags &= ~(CF_PRESOLVE | CF_VC_ONE_POINT);
		bool active = bodyActiveForContact(bfA) || bodyActiveForContact(bfB);
		if (keep && active)
		{
			AABB boxA, boxB;
			boxA.lo = v2(fatA.x, fatA.y); boxA.hi = v2(fatA.z, fatA.w);
			boxB.lo = v2(fatB.x, fatB.y); boxB.hi = v2(fatB.z, fatB.w);
			bool overlap = b2dAabbOverlap(boxA, boxB);
			if (!overlap)
			{
				keep = 0;
			}
			else if (flags & CF_FOREIGN)
			{
				// a spatially sharded world: the bodies are another rank's, which evaluates the manifold; here the contact only
				// keeps existing (the overlap test above is structure: every rank destroys the same contacts)
			}
			else
			{
				// b2Contact::UpdateImpl (b2Contact.cpp:173-298)
				const uint32_t oldId0 = (uint32_t)m3.x, oldId1 = (uint32_t)m3.y;
				const int oldCount = m3.w;
				flags |= CF_ENABLED;
				bool wasTouching = (flags & CF_TOUCHING) != 0;
				bool touching = false;
				bool sensor = (flags & CF_SENSOR) != 0;
				float4 o0s = make_float4(0, 0, 0, 0), o1s = o0s; // the old manifold's vectors, for PreSolve
				Manifold mf;
				mf.pointCount = 0;
				mf.type = m3.z;
				mf.id[0] = oldId0; // a sensor keeps whatever the manifold held (the reference leaves it untouched)
				mf.id[1] = oldId1;
				if (sensor)
				{
					// b2Contact::Update, sensor branch (b2Contact.cpp:193-202): touching = b2TestOverlap (b2Collision.cpp:233-252),
					// GJK distance with the shape radii below 10 epsilon; no manifold. Sensors never enter the solver.
					const GjkProxy pA = b2dProxy(W.shapes + shapeA);
					const GjkProxy pB = b2dProxy(W.shapes + shapeB);
					GjkCache cache;
					cache.count = 0;
					cache.metric = 0.0f;
					for (int k = 0; k < 3; ++k) cache.indexA[k] = cache.indexB[k] = 0;
					GjkOutput dist;
					b2dDistance(dist, cache, pA, xfA, pB, xfB, true);
					touching = dist.distance < 10.0f * B2D_EPSILON;
					mf.pointCount = 0;
				}
				else
				{
					const ShapeRec* sA = W.shapes + shapeA;
					const ShapeRec* sB = W.shapes + shapeB;
					if (STAGE)
					{
						static_assert(sizeof(ShapeRec) % 8 == 0, "ShapeRec is copied in 8-byte pieces");
						const float2* gA = (const float2*)sA;
						const float2* gB = (const float2*)sB;
						float2* lA = (float2*)&s_shape[2 * threadIdx.x];
						float2* lB = (float2*)&s_shape[2 * threadIdx.x + 1];
#pragma unroll
						for (int q = 0; q < (int)(sizeof(ShapeRec) / 8); ++q) { lA[q] = gA[q]; lB[q] = gB[q]; }
						sA = &s_shape[2 * threadIdx.x];
						sB = &s_shape[2 * threadIdx.x + 1];
					}
					// stale fields survive an early-out exactly like the reference's persistent manifold
					o0s = o0;
					o1s = o1;
					mf.localNormal = v2(o0.x, o0.y);
					mf.localPoint = v2(o0.z, o0.w);
					mf.p[0] = v2(o1.x, o1.y);
					mf.p[1] = v2(o1.z, o1.w);
					mf.id[0] = oldId0;
					mf.id[1] = oldId1;
					bool evaluated = false;
					if (uniOn)
					{
						const int u0 = s_uniId[0], u1 = s_uniId[1];
						const bool a0 = shapeA == u0, b0 = shapeB == u0;
						if ((a0 || shapeA == u1) && (b0 || shapeB == u1))
						{
							const ShapeRec* lA = &s_uni[a0 ? 0 : 1];
							const ShapeRec* lB = &s_uni[b0 ? 0 : 1];
							if (lA->type == B2D_SHAPE_POLYGON && lB->type == B2D_SHAPE_POLYGON && lA->count == 4 && lB->count == 4)
							{
								b2dCollidePolygons<4>(&mf, lA, xfA, lB, xfB);
								evaluated = true;
							}
						}
					}
					if (!evaluated) b2dEvaluate(&mf, sA, xfA, sB, xfB);
					touching = mf.pointCount > 0;
					float ni[2], ti[2];
					ni[0] = oldImp.x; ti[0] = oldImp.y; ni[1] = oldImp.z; ti[1] = oldImp.w;
					for (int k = 0; k < mf.pointCount; ++k)
					{
						mf.ni[k] = 0.0f;
						mf.ti[k] = 0.0f;
						uint32_t id2 = mf.id[k];
						if (oldCount > 0 && oldId0 == id2)
						{
							mf.ni[k] = oldImp.x;
							mf.ti[k] = oldImp.y;
						}
						else if (oldCount > 1 && oldId1 == id2)
						{
							mf.ni[k] = oldImp.z;
							mf.ti[k] = oldImp.w;
						}
						ni[k] = mf.ni[k];
						ti[k] = mf.ti[k];
					}
					if (touching != wasTouching)
					{
						// quirk kept: only fixture A's body is woken (b2ContactManager.cpp:472-485)
						W.b_wake[bodyA] = 1;
					}
					// (round 6) only what CHANGED is stored: five of six contacts of a dense pile are fat-AABB pairs that did not touch
					// before and do not touch now - their manifold comes out of b2dEvaluate bit for bit as it went in (the early-outs leave
					// the stale fields alone, as the reference's persistent manifold does), and storing it again was 64 of this kernel's
					// ~150 written bytes per contact. Compared as bits: what is in memory afterwards is the same either way.
					{
						const float4 n0 = make_float4(mf.localNormal.x, mf.localNormal.y, mf.localPoint.x, mf.localPoint.y);
						const float4 n1 = make_float4(mf.p[0].x, mf.p[0].y, mf.p[1].x, mf.p[1].y);
						const float4 nI = make_float4(ni[0], ti[0], ni[1], ti[1]);
						if (!b2dSameBits4(n0, o0)) C.man0[i] = n0;
						if (!b2dSameBits4(n1, o1)) C.man1[i] = n1;
						if (!b2dSameBits4(nI, oldImp)) C.imp[i] = nI;
					}
				}
				if (W.preSolveOn && !sensor && touching)
				{
					// what PreSolve is handed as the old manifold (the new one has just overwritten it above)
					W.pre_o0[i] = o0s;
					W.pre_o1[i] = o1s;
					W.pre_oimp[i] = oldImp;
					W.pre_o3[i] = m3;
					flags |= CF_PRESOLVE;
				}
				{
					const int4 n3 = make_int4((int)mf.id[0], (int)mf.id[1], mf.type, mf.pointCount);
					if (n3.x != m3.x || n3.y != m3.y || n3.z != m3.z || n3.w != m3.w) C.man3[i] = n3;
				}
				if (touching) flags |= CF_TOUCHING; else flags &= ~CF_TOUCHING;
			}
		}

		if (!keep)
		{
			if (W.eventsOn && (flags & CF_REPORTED))
			{
				// b2ContactManager::Destroy (b2ContactManager.cpp:104-107): a touching contact ends when it is destroyed
				const int e = atomicAdd(&S->c.nEvents, 1);
				if (e < W.capContacts)
				{
					W.evKey[e] = C.key[i];
					W.evInfo[e] = make_int4(proxyA, proxyB, 1, -1);
				}
			}
			flags |= CF_DESTROY;
			++nDestroy;
			if (flags & CF_TOI_CANDIDATE) b2dStoreAgentI(&W.toiDestroyList[atomicAdd(&S->c.nToiDestroy, 1)], i); // (read by the last workgroup)
			// b2Contact::Destroy (b2Contact.cpp:100-113): wake both bodies if the manifold had points
			int pc = C.man3[i].w;
			if (pc > 0 && (flags & (CF_SENSOR | CF_FOREIGN)) == 0) // (a foreign contact's point count is not maintained: its owner wakes the bodies)
			{
				W.b_wake[bodyA] = 1;
				W.b_wake[bodyB] = 1;
			}
		}
		else if ((flags & (CF_TOUCHING | CF_FOREIGN)) == CF_TOUCHING)
		{
			++nTouch;
		}
		C.flags[i] = flags;
		W.keepFlag[i] = keep;
	}
	// the census of the pass (destroyed, touching) travels with the arrival of the workgroups (b2d_world.h: b2dTreeArrive -
	// one atomic per WAVE on each of the two counters was 2 x 2 000 atomics on one word for the 128 000 contacts of the 1 M
	// field, served one after the other); the last workgroup to finish adds the totals, and:
	// the TOI candidates destroyed by this pass leave the manager's slot order
	{
		__shared__ int s_census[2];
		if (threadIdx.x == 0) { s_census[0] = 0; s_census[1] = 0; }
		__syncthreads();
		nDestroy = waveSumInt(nDestroy);
		nTouch = waveSumInt(nTouch);
		if (waveLane() == 0)
		{
			if (nDestroy) atomicAdd(&s_census[0], nDestroy);
			if (nTouch) atomicAdd(&s_census[1], nTouch);
		}
		__syncthreads();
		const bool fit = (unsigned)W.capContacts <= TREE_SUM_MAX;
		if (!fit && threadIdx.x == 0)
		{
			if (s_census[0]) atomicAdd(&S->c.nDestroy, s_census[0]);
			if (s_census[1]) atomicAdd(&S->c.nTouching, s_census[1]);
		}
		unsigned t0 = 0u, t1 = 0u;
		if (b2dLastBlockArrive(W, ARRIVE_COLLIDE, fit ? (unsigned)s_census[0] : 0u, fit ? (unsigned)s_census[1] : 0u, &t0, &t1))
		{
			if (threadIdx.x == 0)
			{
				if (t0) atomicAdd(&S->c.nDestroy, (int)t0);
				if (t1) atomicAdd(&S->c.nTouching, (int)t1);
			}
			if (ORDER) toiOrderDestroy(W, s_raw);
		}
	}
}

// toiOrderDestroy as a launch of its own, behind k_collide<.., false> (one workgroup; returns at once when no candidate died).
__global__ __launch_bounds__(256) void k_toi_order_destroy(DW W)
{
	__shared__ __attribute__((aligned(16))) unsigned char s_raw[TOI_ORDER_SCRATCH_BYTES];
	toiOrderDestroy(W, s_raw);
}

// b2World::CreateJoint with collideConnected == false flags the contacts between the two bodies for
// re-filtering at the next step (b2World.cpp:716-732).
// Contact events (the begin / end half of the b2ContactListener bridge): once per step, after the last
// phase, every contact whose touching state differs from what the host was last told yields one event and flips its
// CF_REPORTED bit; contacts destroyed while reported yield their end event in k_collide. The host sorts them by proxy-key
// pair, begins first (the order of b2ContactManager::Collide's deferred callbacks, b2ContactManager.cpp:420-438). One
// net event per contact and step: a begin + end inside one step (possible through TOI sub-steps) cancels out.
__global__ __launch_bounds__(256) void k_contact_events(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const uint32_t flags = C.flags[i];
		const bool touching = (flags & CF_TOUCHING) != 0, reported = (flags & CF_REPORTED) != 0;
		if (touching == reported || (flags & CF_FOREIGN) != 0) continue;
		C.flags[i] = flags ^ CF_REPORTED;
		const int e = atomicAdd(&S->c.nEvents, 1);
		if (e < W.capContacts)
		{
			const int4 ids = C.ids[i];
			W.evKey[e] = C.key[i];
			W.evInfo[e] = make_int4(ids.x, ids.y, touching ? 0 : 1, i);
		}
	}
}

// b2ContactListener::PreSolve records (b2Contact.cpp:283-297): between the scan of the keep flags and the compaction, so
// that the old manifolds saved by k_collide are still addressed by the contact's old index; the record carries the index
// the contact has after the compaction.
__global__ __launch_bounds__(256) void k_presolve_gather(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const bool compacting = S->c.nDestroy != 0;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		if (!W.keepFlag[i] || (C.flags[i] & CF_PRESOLVE) == 0) continue;
		C.flags[i] &= ~CF_PRESOLVE_OFF; // (set again by k_presolve_disable if this call switches the contact off)
		const int e = atomicAdd(&S->c.nPreSolve, 1);
		if (e >= W.capContacts) continue;
		const int4 ids = C.ids[i];
		PreSolveRec r;
		r.info = make_int4(compacting ? W.keepScan[i] : i, ids.x, ids.y, 0);
		r.key = C.key[i];
		r.pad = 0ull;
		r.o0 = W.pre_o0[i]; r.o1 = W.pre_o1[i]; r.oimp = W.pre_oimp[i]; r.o3 = W.pre_o3[i];
		r.n0 = C.man0[i]; r.n1 = C.man1[i]; r.nimp = C.imp[i]; r.n3 = C.man3[i];
		r.mat = C.mat[i];
		W.preRecs[e] = r;
	}
}

// b2Contact::SetEnabled(false) from PreSolve: the listed contacts sit out this step (the next Collide enables them again)
__global__ __launch_bounds__(256) void k_presolve_disable(DW W, const int* list, int count)
{
	b2dPhaseStamp(W);
	const ContactArrays& C = W.ca[W.st->cur];
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
	{
		const int j = list[k];
		if (j >= 0 && j < W.st->c.nContacts) C.flags[j] = (C.flags[j] & ~CF_ENABLED) | CF_PRESOLVE_OFF;
	}
}

// b2Contact::SetFriction / SetRestitution / SetTangentSpeed from PreSolve (b2Contact.h:129-160): the values stay with the
// contact until they are set again (the conveyor belt of Testbed/Tests/ConveyorBelt.h sets its speed in every PreSolve).
// list = count x {contact index, friction, restitution, tangent speed as bits}
__global__ __launch_bounds__(256) void k_presolve_material(DW W, const int* list, int count)
{
	b2dPhaseStamp(W);
	const ContactArrays& C = W.ca[W.st->cur];
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
	{
		const int j = list[4 * k];
		if (j < 0 || j >= W.st->c.nContacts) continue;
		float4 m = C.mat[j];
		m.x = __int_as_float(list[4 * k + 1]);
		m.y = __int_as_float(list[4 * k + 2]);
		m.z = __int_as_float(list[4 * k + 3]);
		C.mat[j] = m;
	}
}

// b2ContactListener::PostSolve records (b2Island::Report, b2Island.cpp:532-570): every contact constraint of the islands
// solved in this step = solid contacts with a non-static body that was in a solved island.
__global__ __launch_bounds__(256) void k_postsolve_gather(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const uint32_t flags = C.flags[i];
		if ((flags & (CF_ENABLED | CF_TOUCHING | CF_SENSOR | CF_DESTROY)) != (CF_ENABLED | CF_TOUCHING)) continue;
		const int4 ids = C.ids[i];
		const uint32_t fA = W.b_flags[ids.z], fB = W.b_flags[ids.w];
		const bool inIsland = ((fA & BF_TYPE_MASK) != BT_STATIC && (fA & BF_ISLAND)) || ((fB & BF_TYPE_MASK) != BT_STATIC && (fB & BF_ISLAND));
		if (!inIsland) continue;
		const int e = atomicAdd(&S->c.nPostSolve, 1);
		if (e >= W.capContacts) continue;
		PostSolveRec r;
		const int pc = C.man3[i].w;
		r.info = make_int4(i, ids.x, ids.y, (flags & CF_VC_ONE_POINT) != 0 && pc > 1 ? 1 : pc);
		r.key = C.key[i];
		r.pad = 0ull;
		r.imp = C.imp[i];
		W.postRecs[e] = r;
	}
}

// Contacts flagged for re-filtering, listed for the user's contact filter (asked on the host before Collide)
__global__ __launch_bounds__(256) void k_filter_list(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		if ((C.flags[i] & CF_FILTER) == 0) continue;
		const int e = atomicAdd(&S->c.nFilterList, 1);
		if (e < W.capContacts) W.filterList[e] = i;
	}
}

__global__ __launch_bounds__(256) void k_filter_reject(DW W, const int* list, int count)
{
	b2dPhaseStamp(W);
	const ContactArrays& C = W.ca[W.st->cur];
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
	{
		const int j = list[k];
		if (j >= 0 && j < W.st->c.nContacts) C.flags[j] |= CF_USER_REJECT;
	}
}

// Candidate pairs the user's contact filter refused: they are no first occurrences any more (nothing is created for them)
__global__ __launch_bounds__(256) void k_pairs_reject(DW W, const int* list, int count)
{
	b2dPhaseStamp(W);
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) W.pairFirst[list[k]] = 0;
}

// Per-step counters back to zero (one launch instead of a handful of memsets); `bar` = the resident solver's grid barrier.
__global__ void k_step_begin(DW W, int* bar)
{
	b2dPhaseStamp(W);
	Counters& c = W.st->c;
	const int t = threadIdx.x;
	if (t == 0)
	{
		c.nDestroy = 0;
		(&c.nDestroy)[1] = 0;
		c.nPairs = 0;
		(&c.nPairs)[1] = 0;
		c.overflow = 0;
		c.cellExtBits = 0;
		c.gridFresh = 0;
		W.st->dbgCensus[0] = 0;
		c.nEvents = 0;
		c.nToiList = 0;
		c.nNewToiCand = 0;
		c.nToiEvents = 0;
		c.nToiCalls = 0;
		c.toiBase = 0;
		c.toiOverflow = 0;
		c.nToiLog = 0;
		c.toiIncomplete = 0;
		c.toiUnsafe = 0;
		c.nToiGroups = 0;
		c.nToiMoved = 0;
		c.nToiNewPairs = 0;
		c.nToiChainCreated = 0;
		c.nPreSolve = 0;
		c.nPostSolve = 0;
		c.nFilterList = 0;
		c.spToiCreated = 0;
		c.spToiStraddle = 0;
	}
	if (t < 32) bar[t] = t == 6 ? W.testSpinMax : 0; // ([6]: B2HIP_TEST_SPIN_MAX - the spin limit of every wait between workgroups, 0 = the built-in ones: tests/test_gpu_recovery.py)
	// (the arrival trees are all zero between launches - whoever completes a word puts it back; a launch that died halfway
	// in a failed step must not leave the next step's close-outs without their last workgroup)
	for (int k = t; k < ARRIVE_SITES * TREE_WORDS; k += (int)blockDim.x) W.arriveTree[k] = 0ull;
}

// The wake requests Collide gathered (ConsumeAwakes, b2Contact::Destroy) are normally applied by the island build that
// follows (k_island_init). A call that continues an open step (sub-stepping) has no island build: applied here.
__global__ __launch_bounds__(256) void k_wake_apply(DW W)
{
	b2dPhaseStamp(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if (W.b_wake[i])
		{
			W.b_flags[i] |= BF_AWAKE;
			W.b_pos[i].w = 0.0f;
			W.b_wake[i] = 0;
		}
	}
}

// b2World::CreateJoint / DestroyJoint (b2World.cpp:716-732, 833-845): contacts between the two bodies of a joint that does
// not let them collide (or no longer keeps them from it) are filtered again by the next Collide. `pairs` = the body pairs of
// every joint created or destroyed since the last step, as sorted (low body << 32 | high body) keys: one launch whatever
// their number (a world built with 20 000 joints used to issue 20 000 launches in its first step).
__global__ __launch_bounds__(256) void k_flag_filter(DW W, const unsigned long long* pairs, int nPairs)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		const unsigned lo = (unsigned)(ids.z < ids.w ? ids.z : ids.w), hi = (unsigned)(ids.z < ids.w ? ids.w : ids.z);
		const unsigned long long key = ((unsigned long long)lo << 32) | hi;
		int a = 0, b = nPairs;
		while (a < b)
		{
			const int m = (a + b) >> 1;
			if (pairs[m] < key) a = m + 1; else b = m;
		}
		if (a < nPairs && pairs[a] == key) C.flags[i] |= CF_FILTER;
	}
}

// Stable compaction (creation order is preserved). keepScan = exclusive scan of keepFlag.
__global__ __launch_bounds__(256) void k_compact_contacts(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nDestroy == 0) return;
	const int n = S->c.nContacts;
	const ContactArrays& A = W.ca[S->cur];
	const ContactArrays& B = W.ca[1 - S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		if (W.keepFlag[i])
		{
			int j = W.keepScan[i];
			const uint32_t fl = A.flags[i];
			B.ids[j] = A.ids[i];
			B.key[j] = A.key[i];
			B.flags[j] = fl;
			if ((fl & CF_FOREIGN) == 0)
			{
				// (another rank's contact: only its place in the array is kept here - b2d_kernels_spatial.h)
				B.mat[j] = A.mat[i];
				B.man0[j] = A.man0[i];
				B.man1[j] = A.man1[i];
				B.imp[j] = A.imp[i];
				B.man3[j] = A.man3[i];
			}
			B.color[j] = A.color[i];
			const int m = A.mgr[i];
			B.mgr[j] = m;
			if (m >= 0) W.toiPos2c[m] = j;
		}
	}
	// the workgroup that finishes last switches the buffers (was a kernel of its own): everybody has read the count and
	// the live half by then
	// (no fence: the last workgroup reads nothing the others wrote - a fence per workgroup writes the L2 back 256 times)
	if (b2dLastBlockArrive(W, ARRIVE_COMPACT) && threadIdx.x == 0)
	{
		S->c.nContacts = W.keepScan[n];
		S->cur = 1 - S->cur;
	}
}

// ---- contact key hash set ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t hashKey(uint64_t k)
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdULL;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ULL;
	k ^= k >> 33;
	return (uint32_t)k;
}

// The part of the table a pair update uses: sized for the LIVE contacts (at most 25 % load), not for the array's capacity -
// the table is cleared every update, and the capacity of a world that once saw a burst of pairs (the 1 M-body field:
// 16 M slots, 128 MB) cost 120 us of clearing per step for 130 000 keys. Every kernel of one update (clear, build, the
// pair search) derives the same mask from Counters::nContacts, which only the update's own commit changes.
__device__ __forceinline__ uint32_t htLiveMask(const DW& W)
{
	const uint32_t want = 4u * (uint32_t)W.st->c.nContacts + 1024u;
	uint32_t m = 0xffffffffu >> __clz((int)(want | 1u)); // next power of two above `want`, minus one
	return m < W.htMask ? m : W.htMask;
}

__device__ __forceinline__ void htInsert(const DW& W, uint64_t key)
{
	const uint32_t mask = htLiveMask(W);
	uint32_t h = hashKey(key) & mask;
	for (;;)
	{
		unsigned long long old = atomicCAS((unsigned long long*)&W.ht_keys[h], 0ull, (unsigned long long)key);
		if (old == 0ull || old == key) return;
		h = (h + 1) & mask;
	}
}

__device__ __forceinline__ bool htContains(const DW& W, uint64_t key)
{
	const uint32_t mask = htLiveMask(W);
	uint32_t h = hashKey(key) & mask;
	for (;;)
	{
		uint64_t v = W.ht_keys[h];
		if (v == key) return true;
		if (v == 0) return false;
		h = (h + 1) & mask;
	}
}

__global__ __launch_bounds__(256) void k_ht_clear(DW W)
{
	b2dPhaseStamp(W);
	if (W.st->c.nMoves == 0) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, live = htLiveMask(W); i <= live; i += gridDim.x * blockDim.x)
	{
		W.ht_keys[i] = 0;
	}
}

__global__ __launch_bounds__(256) void k_ht_build(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nMoves == 0) return;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		htInsert(W, C.key[i] + 1ull); // +1 so that key 0 (proxies 0,0 never pair) cannot collide with "empty"
	}
}

#endif
