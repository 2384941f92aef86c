// b2d_kernels_broadphase.h - fat-AABB broad-phase on the device.
//
// Pair EXISTENCE in the reference is a pure function of the per-proxy fat AABBs and the move list
// (b2BroadPhase::UpdatePairs, b2BroadPhase.h:211-267): the dynamic tree is only an index. We keep
// the same per-proxy fat-AABB state machine (b2DynamicTree::MoveProxy, b2DynamicTree.cpp:130-174)
// and replace the tree by a hashed uniform grid rebuilt on the device, so the pair SET is identical;
// creation ORDER is then fixed by sorting on (proxyKeyLow, proxyKeyHigh) exactly like
// b2ContactManager::FinishFindNewContacts (b2ContactManager.cpp:366-386).
#ifndef B2D_KERNELS_BROADPHASE_H
#define B2D_KERNELS_BROADPHASE_H

#include "b2d_kernels_solve_large.h"

// b2ContactManager::SynchronizeFixtures (:315-364) + FinishSynchronizeFixtures (:441-452) +
// b2DynamicTree::MoveProxy, one lane per proxy. The proxies that left their fat AABB are collected per workgroup (a tile of
// SYNC_TILE proxies, the list in LDS) and take their places in the move buffer with ONE atomic on its counter: in a world
// where everything moves (the 1 M-body field: 600 000 moved proxies per step) one atomic per wave was 16 000 on the same
// word, ~10 ns each - most of the kernel's 215 us. (The order of the move buffer has no meaning: the pairs are sorted.)
#define SYNC_TILE 1024
__global__ __launch_bounds__(256) void k_sync_fixtures(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = W.nProxies;
	const int tid = (int)threadIdx.x;
	__shared__ int s_list[SYNC_TILE];
	__shared__ int s_cnt, s_base;
	// (a small world keeps one proxy per lane - 10 000 proxies in tiles of 1 024 would be ten workgroups on 256 CUs)
	const int perLane = n >= 262144 ? SYNC_TILE / 256 : 1, tile = 256 * perLane;
	if (blockIdx.x == 0 && tid == 0) S->c.gridFresh = 0; // (boxes move: whatever grid there is is stale until the next pair update)
	for (int base = blockIdx.x * tile; base < n; base += gridDim.x * tile)
	{
		if (tid == 0) s_cnt = 0;
		__syncthreads();
		for (int j = 0; j < perLane; ++j)
		{
			const int p = base + j * 256 + tid;
			if (p >= n) break;
			const int body = W.p_body[p];
			if (body < 0) continue;
			uint32_t f = W.b_flags[body];
			// If a body was not in an island then it did not move.
			if ((f & BF_ISLAND) == 0) continue;
			float4 m = W.b_mass[body];
			float4 p0 = W.b_pos0[body];
			Xf xf1 = b2dXfFromSweep(v2(p0.x, p0.y), p0.z, v2(m.z, m.w));
			Xf xf2 = loadXf(W.b_xf, body);
			const ShapeRec* shape = W.shapes + W.p_shape[p];
			AABB aabb = b2dAabbCombine(b2dShapeAABB(shape, xf1), b2dShapeAABB(shape, xf2));
			AABB fat = loadAabb(W.p_fat, p);
			if (b2dAabbContains(fat, aabb)) continue;
			V2 displacement = xf2.p - xf1.p;
			AABB b = aabb;
			b.lo = v2(b.lo.x - B2D_AABB_EXTENSION, b.lo.y - B2D_AABB_EXTENSION);
			b.hi = v2(b.hi.x + B2D_AABB_EXTENSION, b.hi.y + B2D_AABB_EXTENSION);
			V2 d = B2D_AABB_MULTIPLIER * displacement;
			if (d.x < 0.0f) b.lo.x += d.x; else b.hi.x += d.x;
			if (d.y < 0.0f) b.lo.y += d.y; else b.hi.y += d.y;
			W.p_fat[p] = make_float4(b.lo.x, b.lo.y, b.hi.x, b.hi.y);
			s_list[atomicAdd(&s_cnt, 1)] = p;
		}
		__syncthreads();
		const int cnt = s_cnt;
		if (tid == 0 && cnt > 0) s_base = atomicAdd(&S->c.nMoves, cnt);
		__syncthreads();
		if (cnt > 0)
		{
			const int at = s_base;
			for (int k = tid; k < cnt; k += 256)
			{
				if (at + k < W.capMoves) W.moveBuf[at + k] = s_list[k];
				else atomicOr(&S->c.overflow, 8);
			}
		}
		__syncthreads();
	}
}

// ---- hashed uniform grid over ALL proxies ----------------------------------------------------------
__device__ __forceinline__ uint32_t cellHash(int ix, int iy, uint32_t mask)
{
	// two odd multipliers, then an avalanche before the power-of-two mask: bodies laid out on a regular lattice (tile
	// maps, rows of vehicles) otherwise land in a small fraction of the buckets (measured: 10 000 cars on a 12 x 10 m
	// lattice, 3.1 ms in k_find_pairs_small with the plain xor of products)
	uint32_t h = (uint32_t)ix * 73856093u ^ (uint32_t)iy * 19349663u;
	h ^= h >> 15;
	h *= 0x2c1b3c6du;
	h ^= h >> 12;
	h *= 0x297a2d39u;
	h ^= h >> 15;
	return h & mask;
}

// The geometry of the hashed grid. gridLimit: the widest proxy that goes through the grid at all. DW::cellSize is sized from the
// fixtures as they were created; fat AABBs of fast bodies stretch (b2DynamicTree::MoveProxy adds twice the displacement), and
// a proxy wider than the limit takes the brute-force "large proxy" path: thousands of falling boxes did in the Tumbler. So
// the limit follows the widest proxy of this pair update (k_bp_clear reduces it into Counters::cellExtBits), capped at 4 x
// the static size - beyond that a proxy is an outlier (a ground box) and stays on the large path.
// gridCell: the cell is the limit, or - in dense scenes (DW::gridHalf) - half of it. Proxies are binned by the cell of their
// centre, so every partner of a box `a` has its centre inside `a` grown by half the limit (gridWindow); with a cell as wide
// as the limit that window is 3 x 3 cells = 9 limit^2 of candidates for every proxy, however small - 355 per moved proxy on
// the settled Tumbler, where the pair search runs at 12 G candidates per second whatever its form. Half cells make the
// window follow the proxy - 4-5 cells a side, (extent + 1.5 limit)^2: Tumbler 730 -> 590 us, Pyramid 316 237 -> 207 us - but
// cost a sparse scene more cells to look up than candidates to save (1 M field: 370 -> 500 us). The host switches on the
// candidates per moved proxy of the previous update (Counters::candRounds). The pair SET does not depend on any of this.
__device__ __forceinline__ float gridLimit(const DW& W)
{
	const float dyn = 1.05f * __uint_as_float(W.st->c.cellExtBits);
	return dyn > W.cellSize ? dyn : W.cellSize;
}
__device__ __forceinline__ float gridCell(const DW& W) { return W.gridHalf ? 0.5f * gridLimit(W) : gridLimit(W); }

__device__ __forceinline__ bool proxyIsLarge(const DW& W, float4 a)
{
	const float limit = gridLimit(W);
	return (a.z - a.x) > limit || (a.w - a.y) > limit;
}

__device__ __forceinline__ void proxyCell(const DW& W, float4 a, int* ix, int* iy)
{
	const float inv = 1.0f / gridCell(W);
	*ix = (int)floorf(0.5f * (a.x + a.z) * inv);
	*iy = (int)floorf(0.5f * (a.y + a.w) * inv);
}

// The cells that can hold a partner of `a`: [ix0, ix0 + nx) x [iy0, iy0 + ny). false: degenerate or absurdly large (NaN, a
// box of kilometres): the caller looks at everything.
#define GRID_WINDOW_MAX 36 // cells of the window of a grid-sized proxy: at most 6 x 6 (extent <= limit = 2 cells, + 1 either side, + rounding)
__device__ __forceinline__ bool gridWindow(const DW& W, float4 a, int* ix0, int* iy0, int* nx, int* ny)
{
	const float limit = gridLimit(W), inv = 1.0f / gridCell(W), half = 0.5f * limit;
	const float fx0 = floorf((a.x - half) * inv), fx1 = floorf((a.z + half) * inv);
	const float fy0 = floorf((a.y - half) * inv), fy1 = floorf((a.w + half) * inv);
	const float cells = (fx1 - fx0 + 1.0f) * (fy1 - fy0 + 1.0f);
	if (!(cells >= 1.0f && cells <= 4096.0f)) return false;
	*ix0 = (int)fx0;
	*iy0 = (int)fy0;
	*nx = (int)(fx1 - fx0) + 1;
	*ny = (int)(fy1 - fy0) + 1;
	return true;
}

// force = 1: rebuild the grid although the move buffer is empty (the TOI phase queries it and needs it to reflect
// every fat AABB as of now; the pair census of the finished pair update is left alone)
__global__ __launch_bounds__(256) void k_grid_clear(DW W, int force)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (force && S->c.gridFresh) return; // (this step's pair update built it, nothing has moved since)
	// always reset the pair census, also when nothing moved: the ordering / creation kernels that
	// follow key off nPairs and must see 0 then
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		S->c.nLargeProxies = 0;
		if (!force)
		{
			S->c.nPairs = 0;
			S->c.nNewContacts = 0;
		}
	}
	if (S->c.nMoves == 0 && !force) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= W.gridMask; i += gridDim.x * blockDim.x)
	{
		W.gridCount[i] = 0;
		W.gridCursor[i] = 0;
	}
}

__global__ __launch_bounds__(256) void k_grid_count(DW W, int force)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nMoves == 0 && !force) return;
	if (force && S->c.gridFresh) return;
	const int n = W.nProxies;
	for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x)
	{
		if (W.p_body[p] < 0) continue;
		float4 a = W.p_fat[p];
		if (proxyIsLarge(W, a))
		{
			int k = atomicAdd(&S->c.nLargeProxies, 1);
			W.largeProxies[k] = p;
		}
		else
		{
			int ix, iy;
			proxyCell(W, a, &ix, &iy);
			atomicAdd(&W.gridCount[cellHash(ix, iy, W.gridMask)], 1);
		}
	}
}

__global__ __launch_bounds__(256) void k_grid_fill(DW W, int force)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nMoves == 0 && !force) return;
	if (force && S->c.gridFresh) return;
	const int n = W.nProxies;
	for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x)
	{
		if (W.p_body[p] < 0) continue;
		float4 a = W.p_fat[p];
		if (proxyIsLarge(W, a)) continue;
		int ix, iy;
		proxyCell(W, a, &ix, &iy);
		uint32_t h = cellHash(ix, iy, W.gridMask);
		// (the box travels with the item: the pair search tests a candidate without a second, dependent load - its time is the
		// chain of loads per moved proxy, and cell by cell the boxes lie together)
		const int slot = W.gridStart[h] + atomicAdd(&W.gridCursor[h], 1);
		W.gridItems[slot] = p;
		W.gridFat[slot] = a;
	}
}

// b2ContactManager::AddPair filters (:237-312) evaluated before the pair is even stored: does (p, q) become a pair, and which
// (key = the two proxy keys, lower first; lo / hi = the proxies in that order)
__device__ __forceinline__ bool pairPasses(const DW& W, int p, int q, uint64_t* keyOut, int2* proxOut)
{
	const int bodyP = W.p_body[p], bodyQ = W.p_body[q];
	if (bodyP == bodyQ) return false;
	if (W.spatial)
	{
		// a spatially sharded world: a rank emits the pairs one of ITS bodies takes part in (a moved static proxy - a host edit
		// every rank made - is searched by every rank: its pair with another rank's body is that rank's to emit). The hash set
		// of existing contacts then only needs this rank's own contacts (k_bp_build).
		const bool mineP = (W.b_flags[bodyP] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[bodyP] == (uint8_t)W.shardRank;
		const bool mineQ = (W.b_flags[bodyQ] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[bodyQ] == (uint8_t)W.shardRank;
		if (!mineP && !mineQ) return false;
	}
	const int keyP = W.p_key[p], keyQ = W.p_key[q];
	const int lo = keyP < keyQ ? p : q;
	const int hi = keyP < keyQ ? q : p;
	const uint64_t key = ((uint64_t)(uint32_t)W.p_key[lo] << 32) | (uint32_t)W.p_key[hi];
	if (htContains(W, key + 1ull)) return false;
	// bodyB->ShouldCollide(bodyA) with A = lower proxy id
	if (!bodiesShouldCollide(W, W.p_body[hi], W.p_body[lo])) return false;
	// (a user contact filter replaces the built-in rule: it is asked on the host for every pair that gets this far)
	if (!W.userFilter && !filterShouldCollide(W.p_filter0[lo], W.p_filter1[lo], W.p_filter0[hi], W.p_filter1[hi])) return false;
	if (b2dContactSwap(W.shapes[W.p_shape[lo]].type, W.shapes[W.p_shape[hi]].type) < 0) return false;
	*keyOut = key;
	*proxOut = make_int2(lo, hi);
	return true;
}

// ... and stored at once, a place of the pair buffer per pair (the brute-force search of the large proxies, lists beyond a wave)
__device__ __forceinline__ void tryEmitPair(const DW& W, DState* S, int p, int q)
{
	uint64_t key;
	int2 prox;
	if (!pairPasses(W, p, q, &key, &prox)) return;
	int k = atomicAdd(&S->c.nPairs, 1);
	if (k < W.capPairs)
	{
		W.pairKey[k] = key;
		W.pairProxy[k] = prox;
	}
	else
	{
		atomicOr(&S->c.overflow, 2);
	}
}

// The pairs a WAVE of the search kernels has found wait in LDS (round 6) and take their places in the pair buffer PAIR_STAGE_CAP
// at a time; what is left when the kernel ends goes out with one atomic per workgroup. A place per pair as it turned up - the
// compiler combines the lanes of a wave, so: one returning atomic per flush of the hit list below - was ~100 000 atomics on
// ONE word in a step of the settled 100 000-box Tumbler (530 000 pairs: every box moves and finds each new neighbour from both
// sides), served one after the other at the memory side in ~4 ns each: the 370 us of k_find_pairs_window were that queue.
// The order of the pair buffer means nothing: it is sorted by key before anything reads it.
#define PAIR_STAGE_CAP 256
struct PairStage
{
	uint64_t* key; // [PAIR_STAGE_CAP] of this wave
	int2* prox;
	int n;
};
__device__ __forceinline__ void pairStageStore(const DW& W, DState* S, const PairStage& st, int lane, int base)
{
	for (int j = lane; j < st.n; j += 64)
	{
		const int k = base + j;
		if (k < W.capPairs)
		{
			W.pairKey[k] = st.key[j];
			W.pairProxy[k] = st.prox[j];
		}
	}
	if (lane == 0 && base + st.n > W.capPairs) atomicOr(&S->c.overflow, 2);
}
__device__ __forceinline__ void pairStageDrain(const DW& W, DState* S, PairStage& st, int lane)
{
	if (st.n == 0) return;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	int base = 0;
	if (lane == 0) base = atomicAdd(&S->c.nPairs, st.n);
	base = __builtin_amdgcn_readfirstlane(base);
	pairStageStore(W, S, st, lane, base);
	__builtin_amdgcn_wave_barrier(); // (the stage is written again)
	st.n = 0;
}
// (every wave of the workgroup, once, at the end of the kernel)
__device__ __forceinline__ void pairStageFinish(const DW& W, DState* S, PairStage& st, int lane)
{
	__shared__ int s_stageCount[16], s_stageBase;
	const int wv = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63u) >> 6);
	if (lane == 0) s_stageCount[wv] = st.n;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		int run = 0;
		for (int q = 0; q < nw; ++q) { const int c = s_stageCount[q]; s_stageCount[q] = run; run += c; }
		s_stageBase = run > 0 ? atomicAdd(&S->c.nPairs, run) : 0;
	}
	__syncthreads();
	pairStageStore(W, S, st, lane, s_stageBase + s_stageCount[wv]);
	st.n = 0;
}

// The candidates whose boxes overlap are put aside - (proxy, candidate) in a list per wave in LDS - and go through the filters
// (pairPasses: keys, hash probe of the existing contacts, body and fixture rules, shape types - a chain of a dozen gathers)
// 64 at a time with every lane busy, instead of where they turn up: a dense window is five or six rounds of 64 candidates
// with a few hits each, a sparse scene has a hit every second proxy - the chain ran once per round / proxy for those few
// lanes (Tumbler: pair update 1.50 -> 1.17 ms with the list per window; then per wave across proxies).
struct PairHits
{
	int2* list; // [64] of this wave
	int n;
};
__device__ __forceinline__ void pairHitsFlush(const DW& W, DState* S, PairHits& h, PairStage& st, int lane)
{
	if (h.n == 0) return;
	if (st.n + h.n > PAIR_STAGE_CAP) pairStageDrain(W, S, st, lane);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	uint64_t key = 0ull;
	int2 prox = make_int2(0, 0);
	bool ok = false;
	if (lane < h.n)
	{
		const int2 pq = h.list[lane];
		ok = pairPasses(W, pq.x, pq.y, &key, &prox);
	}
	const unsigned long long om = __ballot(ok);
	if (ok)
	{
		const int j = st.n + (int)__popcll(om & ((1ull << lane) - 1ull));
		st.key[j] = key;
		st.prox[j] = prox;
	}
	st.n += (int)__popcll(om);
	__builtin_amdgcn_wave_barrier(); // (the list is written again)
	h.n = 0;
}
__device__ __forceinline__ void pairHitsAdd(const DW& W, DState* S, PairHits& h, PairStage& st, int lane, bool hit, int p, int q)
{
	const unsigned long long hm = __ballot(hit);
	if (hm == 0ull) return;
	const int more = (int)__popcll(hm);
	if (h.n + more > 64) pairHitsFlush(W, S, h, st, lane);
	if (hit) h.list[h.n + (int)__popcll(hm & ((1ull << lane) - 1ull))] = make_int2(p, q);
	h.n += more;
}

// One WAVE per moved SMALL proxy. The candidates of its 3x3 cell neighbourhood are flattened into one
// index space so that 64 candidates are fetched and tested at once (the per-candidate chain
// item -> fat AABB -> filters -> hash probe is then paid once per wave, not once per candidate).
__global__ __launch_bounds__(256) void k_find_pairs_small(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int nm = S->c.nMoves < W.capMoves ? S->c.nMoves : W.capMoves;
	const int nLarge = S->c.nLargeProxies;
	const int lane = (int)(threadIdx.x & 63u);
	const int waveId = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nWaves = (int)((gridDim.x * blockDim.x) >> 6);
	int rounds = 0;
	__shared__ int2 s_hits[4][64];
	__shared__ uint64_t s_stageKey[4][PAIR_STAGE_CAP];
	__shared__ int2 s_stageProx[4][PAIR_STAGE_CAP];
	PairHits hits;
	hits.list = s_hits[threadIdx.x >> 6];
	hits.n = 0;
	PairStage stage;
	stage.key = s_stageKey[threadIdx.x >> 6];
	stage.prox = s_stageProx[threadIdx.x >> 6];
	stage.n = 0;
	// The search is a chain of dependent loads per moved proxy - proxy, its body and box, cell headers, items, and the large
	// proxies with their boxes - and the kernel's time is that chain times the proxies a wave goes through. So: the large
	// proxies (walls, the ground: a handful) are fetched ONCE per wave, lane t keeping the t-th; the proxy of the round after
	// next and the body and box of the next round are asked for while this round works.
	const int largeQ = lane < nLarge ? W.largeProxies[lane] : -1;
	const float4 largeFat = largeQ >= 0 ? W.p_fat[largeQ] : make_float4(0, 0, 0, 0);
	// (round 6: ... and the cell headers of the NEXT round's proxy - count and start of its nine cells - while this round
	// fetches and tests its candidates: a round is then one round trip, the candidates', not two in a row. The million
	// proxies of the field are 122 rounds for each of the 8 192 waves the chip holds.)
	// lanes 0..8 own one neighbour cell each; cells that hash to an already seen bucket are dropped
	auto cellHeaders = [&](const float4& f, int* cntOut, int* startOut)
	{
		int ix, iy;
		proxyCell(W, f, &ix, &iy);
		const uint32_t h = lane < 9 ? cellHash(ix + (lane % 3) - 1, iy + (lane / 3) - 1, W.gridMask) : 0xffffffffu;
		// (lane indices known at compile time: v_readlane - a scalar read of one lane - instead of a trip through the LDS
		// crossbar per value; the kernel issues ~50 of these per proxy and is bound by them in sparse scenes)
		bool dup = false;
#pragma unroll
		for (int j = 0; j < 8; ++j)
		{
			const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, j);
			if (j < lane && lane < 9 && hj == h) dup = true;
		}
		*cntOut = (lane < 9 && !dup) ? W.gridCount[h] : 0;
		*startOut = lane < 9 ? W.gridStart[h] : 0;
	};
	int pNext = waveId < nm ? W.moveBuf[waveId] : -1;
	int pAhead = waveId + nWaves < nm ? W.moveBuf[waveId + nWaves] : -1;
	int pAhead2 = waveId + 2 * nWaves < nm ? W.moveBuf[waveId + 2 * nWaves] : -1;
	int bodyNext = W.p_body[pNext < 0 ? 0 : pNext];
	float4 fatNext = W.p_fat[pNext < 0 ? 0 : pNext];
	int bodyAhead = W.p_body[pAhead < 0 ? 0 : pAhead];
	float4 fatAhead = W.p_fat[pAhead < 0 ? 0 : pAhead];
	int cntNext = 0, startNext = 0;
	cellHeaders(fatNext, &cntNext, &startNext);
	for (int k = waveId; k < nm; k += nWaves)
	{
		const int p = pNext;
		const int bodyOfP = bodyNext;
		const float4 a4 = fatNext;
		const int cnt = cntNext, start = startNext;
		pNext = pAhead;
		bodyNext = bodyAhead;
		fatNext = fatAhead;
		pAhead = pAhead2;
		pAhead2 = k + 3 * nWaves < nm ? W.moveBuf[k + 3 * nWaves] : -1;
		bodyAhead = W.p_body[pAhead < 0 ? 0 : pAhead];
		fatAhead = W.p_fat[pAhead < 0 ? 0 : pAhead];
		cellHeaders(fatNext, &cntNext, &startNext);
		if (p < 0 || bodyOfP < 0) continue;
		// (a spatially sharded world: every rank searches for the proxies ITS bodies moved; b2d_kernels_spatial.h, E2)
		if (W.spatial && (W.b_flags[bodyOfP] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[bodyOfP] != (uint8_t)W.shardRank) continue;
		if (proxyIsLarge(W, a4))
		{
			// (for k_find_pairs_large; the list is as long as the move buffer)
			if (lane == 0) W.largeMoves[atomicAdd(&S->c.nLargeMoves, 1)] = p;
			continue;
		}
		AABB a;
		a.lo = v2(a4.x, a4.y);
		a.hi = v2(a4.z, a4.w);
		int ecs[9], ccs[9], scs[9];
		int total = 0;
#pragma unroll
		for (int c = 0; c < 9; ++c)
		{
			ccs[c] = __builtin_amdgcn_readlane(cnt, c);
			scs[c] = __builtin_amdgcn_readlane(start, c);
			ecs[c] = total;
			total += ccs[c];
		}
		{ const int r = (total + 63) >> 6; rounds += r < 8 ? r : 8; } // (at most 8 per proxy enter the census: its 24-bit field in the arrival word holds 8 x capMoves - ADVICE round 4; the host only asks whether the mean is above ~2)
		for (int base = 0; base < total; base += 64)
		{
			const int idx = base + lane;
			const bool valid = idx < total;
			int t = -1;
#pragma unroll
			for (int c = 0; c < 9; ++c)
			{
				const int ec = ecs[c], cc = ccs[c], sc = scs[c];
				if (valid && idx >= ec && idx < ec + cc) t = sc + (idx - ec);
			}
			bool hit = false;
			int q = -1;
			if (t >= 0)
			{
				q = W.gridItems[t];
				const float4 fq = W.gridFat[t];
				AABB bq;
				bq.lo = v2(fq.x, fq.y);
				bq.hi = v2(fq.z, fq.w);
				hit = q != p && b2dAabbOverlap(a, bq);
			}
			pairHitsAdd(W, S, hits, stage, lane, hit, p, q);
		}
		{
			AABB bq;
			bq.lo = v2(largeFat.x, largeFat.y);
			bq.hi = v2(largeFat.z, largeFat.w);
			pairHitsAdd(W, S, hits, stage, lane, largeQ >= 0 && b2dAabbOverlap(a, bq), p, largeQ);
		}
		for (int t = 64 + lane; t < nLarge; t += 64) // (more than a wave holds: the rest as before)
		{
			const int q = W.largeProxies[t];
			if (b2dAabbOverlap(a, loadAabb(W.p_fat, q))) tryEmitPair(W, S, p, q);
		}
	}
	pairHitsFlush(W, S, hits, stage, lane);
	pairStageFinish(W, S, stage, lane);
	// (the candidate census for the host's choice of the cell: one atomic per wave on the 32 words of ONE line was 8 192
	// atomics at the end of the kernel, ~5 ns each - carried by the arrival of the workgroups instead, b2d_world.h)
	b2dBlockTreeAdd2(W, ARRIVE_PAIRS, &S->c.candRounds[0], lane == 0 ? rounds : 0, &S->c.candRounds[1], 0, (unsigned)W.capMoves <= (TREE_SUM_MAX >> 3));
}


// The same with half cells (DW::gridHalf): lane c owns cell c of the proxy's window (gridWindow).
// One WAVE per moved SMALL proxy. The candidates of its 3x3 cell neighbourhood are flattened into one
// index space so that 64 candidates are fetched and tested at once (the per-candidate chain
// item -> fat AABB -> filters -> hash probe is then paid once per wave, not once per candidate).
__global__ __launch_bounds__(256) void k_find_pairs_window(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int nm = S->c.nMoves < W.capMoves ? S->c.nMoves : W.capMoves;
	const int nLarge = S->c.nLargeProxies;
	const int lane = (int)(threadIdx.x & 63u);
	const int waveId = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nWaves = (int)((gridDim.x * blockDim.x) >> 6);
	int rounds = 0;
	__shared__ int2 s_hits[4][64];
	__shared__ uint64_t s_stageKey[4][PAIR_STAGE_CAP];
	__shared__ int2 s_stageProx[4][PAIR_STAGE_CAP];
	PairHits hits;
	hits.list = s_hits[threadIdx.x >> 6];
	hits.n = 0;
	PairStage stage;
	stage.key = s_stageKey[threadIdx.x >> 6];
	stage.prox = s_stageProx[threadIdx.x >> 6];
	stage.n = 0;
	// The search is a chain of dependent loads per moved proxy - proxy, its body and box, cell headers, items, and the large
	// proxies with their boxes - and the kernel's time is that chain times the proxies a wave goes through. So: the large
	// proxies (walls, the ground: a handful) are fetched ONCE per wave, lane t keeping the t-th; the proxy of the round after
	// next and the body and box of the next round are asked for while this round works.
	const int largeQ = lane < nLarge ? W.largeProxies[lane] : -1;
	const float4 largeFat = largeQ >= 0 ? W.p_fat[largeQ] : make_float4(0, 0, 0, 0);
	int pNext = waveId < nm ? W.moveBuf[waveId] : -1;
	int pAhead = waveId + nWaves < nm ? W.moveBuf[waveId + nWaves] : -1;
	int bodyNext = W.p_body[pNext < 0 ? 0 : pNext];
	float4 fatNext = W.p_fat[pNext < 0 ? 0 : pNext];
	for (int k = waveId; k < nm; k += nWaves)
	{
		const int p = pNext;
		const int bodyOfP = bodyNext;
		const float4 a4 = fatNext;
		pNext = pAhead;
		pAhead = k + 2 * nWaves < nm ? W.moveBuf[k + 2 * nWaves] : -1;
		bodyNext = W.p_body[pNext < 0 ? 0 : pNext];
		fatNext = W.p_fat[pNext < 0 ? 0 : pNext];
		if (p < 0 || bodyOfP < 0) continue;
		// (a spatially sharded world: every rank searches for the proxies ITS bodies moved; b2d_kernels_spatial.h, E2)
		if (W.spatial && (W.b_flags[bodyOfP] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[bodyOfP] != (uint8_t)W.shardRank) continue;
		if (proxyIsLarge(W, a4))
		{
			// (for k_find_pairs_large; the list is as long as the move buffer)
			if (lane == 0) W.largeMoves[atomicAdd(&S->c.nLargeMoves, 1)] = p;
			continue;
		}
		AABB a;
		a.lo = v2(a4.x, a4.y);
		a.hi = v2(a4.z, a4.w);
		// the window of this proxy: lane c owns cell c of it; cells that hash to a bucket already seen are dropped. Full cells:
		// always the 3 x 3 neighbourhood of the centre's cell; half cells: what gridWindow says (at most GRID_WINDOW_MAX)
		int wx0 = 0, wy0 = 0, wnx = 1, wny = 1;
		(void)gridWindow(W, a4, &wx0, &wy0, &wnx, &wny); // (a grid-sized proxy: always a window of a few cells)
		const int nCells = wnx * wny < GRID_WINDOW_MAX ? wnx * wny : GRID_WINDOW_MAX;
		uint32_t h = lane < nCells ? cellHash(wx0 + lane % wnx, wy0 + lane / wnx, W.gridMask) : 0xffffffffu;
		bool dup = false;
		// (j, c below: the same for all lanes - v_readlane, not a trip through the LDS crossbar; see k_find_pairs_small)
		for (int j = 0; j + 1 < nCells; ++j)
		{
			const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, j);
			if (j < lane && lane < nCells && hj == h) dup = true;
		}
		const int cnt = (lane < nCells && !dup) ? W.gridCount[h] : 0;
		const int start = lane < nCells ? W.gridStart[h] : 0;
		int incl = cnt;
		for (int off = 1; off < 64; off <<= 1)
		{
			int v = __shfl_up(incl, off);
			if (lane >= off) incl += v;
		}
		const int excl = incl - cnt;
		const int total = __shfl(incl, 63);
		{ const int r = (total + 63) >> 6; rounds += r < 8 ? r : 8; } // (at most 8 per proxy enter the census: its 24-bit field in the arrival word holds 8 x capMoves - ADVICE round 4; the host only asks whether the mean is above ~2)
		for (int base = 0; base < total; base += 64)
		{
			const int idx = base + lane;
			const bool valid = idx < total;
			int t = -1;
			{
				// which cell candidate idx falls into: the first cell whose inclusive count exceeds idx - a binary search over the
				// lanes' prefix sums, six steps whatever the window (<= 36 cells). The walk over all cells it replaces - three lane
				// reads and five operations per cell and round - was most of this kernel's time: 36 cells x 64 candidates a round.
				int lo = 0, hi = nCells - 1;
#pragma unroll
				for (int step = 0; step < 6; ++step)
				{
					const int mid = (lo + hi) >> 1;
					const int im = __shfl(incl, mid);
					if (im > idx) hi = mid; else lo = mid + 1;
				}
				const int c = lo < 63 ? lo : 63;
				const int ec = __shfl(excl, c), sc = __shfl(start, c);
				if (valid) t = sc + (idx - ec);
			}
			bool hit = false;
			int q = -1;
			if (t >= 0)
			{
				q = W.gridItems[t];
				const float4 fq = W.gridFat[t];
				AABB bq;
				bq.lo = v2(fq.x, fq.y);
				bq.hi = v2(fq.z, fq.w);
				hit = q != p && b2dAabbOverlap(a, bq);
			}
			pairHitsAdd(W, S, hits, stage, lane, hit, p, q);
		}
		{
			AABB bq;
			bq.lo = v2(largeFat.x, largeFat.y);
			bq.hi = v2(largeFat.z, largeFat.w);
			pairHitsAdd(W, S, hits, stage, lane, largeQ >= 0 && b2dAabbOverlap(a, bq), p, largeQ);
		}
		for (int t = 64 + lane; t < nLarge; t += 64) // (more than a wave holds: the rest as before)
		{
			const int q = W.largeProxies[t];
			if (b2dAabbOverlap(a, loadAabb(W.p_fat, q))) tryEmitPair(W, S, p, q);
		}
	}
	pairHitsFlush(W, S, hits, stage, lane);
	pairStageFinish(W, S, stage, lane);
	// (the candidate census for the host's choice of the cell: one atomic per wave on the 32 words of ONE line was 8 192
	// atomics at the end of the kernel, ~5 ns each - carried by the arrival of the workgroups instead, b2d_world.h)
	b2dBlockTreeAdd2(W, ARRIVE_PAIRS, &S->c.candRounds[0], lane == 0 ? rounds : 0, &S->c.candRounds[1], 0, (unsigned)W.capMoves <= (TREE_SUM_MAX >> 3));
}

// Moved LARGE proxies (listed by k_find_pairs_small): brute force over every proxy, a workgroup per (proxy, slice of 1024
// candidates) - the four walls of the Tumbler's container move every step, and one workgroup per wall walking 100 000
// proxies was 360 us of a 4.9 ms step with 252 CUs idle.
__global__ __launch_bounds__(256) void k_find_pairs_large(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int nLM = S->c.nLargeMoves;
	const int slices = (W.nProxies + 1023) / 1024;
	const long long units = (long long)nLM * slices;
	for (long long u = blockIdx.x; u < units; u += gridDim.x)
	{
		const int k = (int)(u / slices), slice = (int)(u - (long long)k * slices);
		const int p = W.largeMoves[k];
		const float4 a4 = W.p_fat[p];
		AABB a;
		a.lo = v2(a4.x, a4.y);
		a.hi = v2(a4.z, a4.w);
		// four candidates per lane: their loads are issued together (the emit path contains atomics, which the compiler
		// will not move loads across; one candidate per trip made this a chain of dependent round trips)
		const int q0 = slice * 1024 + (int)threadIdx.x;
		int body[4];
		float4 fat[4];
#pragma unroll
		for (int v = 0; v < 4; ++v)
		{
			const int q = q0 + v * 256;
			const bool in = q < W.nProxies;
			body[v] = in ? W.p_body[q] : -1;
			fat[v] = in ? W.p_fat[q] : make_float4(0, 0, 0, 0);
		}
#pragma unroll
		for (int v = 0; v < 4; ++v)
		{
			const int q = q0 + v * 256;
			if (q == p || body[v] < 0) continue;
			AABB b;
			b.lo = v2(fat[v].x, fat[v].y);
			b.hi = v2(fat[v].z, fat[v].w);
			if (!b2dAabbOverlap(a, b)) continue;
			tryEmitPair(W, S, p, q);
		}
	}
}

// ---- ordering of the new pairs ---------------------------------------------------------------------
// Small sets (<= COUNT_RANK_MAX): rank by counting, tiles of keys staged through LDS.
__global__ __launch_bounds__(256) void k_pairs_first(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nPairs < W.capPairs ? S->c.nPairs : W.capPairs;
	if (n == 0 || n > COUNT_RANK_MAX) return;
	__shared__ uint64_t tile[256];
	const int rounds = (n + 255) / 256;
	for (int base = blockIdx.x * 256; base < rounds * 256; base += gridDim.x * 256)
	{
		const int i = base + threadIdx.x;
		const uint64_t key = i < n ? W.pairKey[i] : 0;
		int first = 1;
		for (int t0 = 0; t0 < n; t0 += 256)
		{
			__syncthreads();
			// the tail of the last tile is padded with a key no pair can have: every tile is a full, unrolled 256-compare
			// loop (16 LDS reads in flight instead of one 64-cycle round trip per compare)
			tile[threadIdx.x] = t0 + threadIdx.x < n ? W.pairKey[t0 + threadIdx.x] : ~0ull;
			__syncthreads();
#pragma unroll 16
			for (int t = 0; t < 256; ++t)
			{
				if (tile[t] == key && t0 + t < i) first = 0;
			}
		}
		if (i < n) W.pairFirst[i] = first;
	}
}

__global__ __launch_bounds__(256) void k_pairs_rank(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nPairs < W.capPairs ? S->c.nPairs : W.capPairs;
	if (n == 0 || n > COUNT_RANK_MAX) return;
	__shared__ uint64_t tile[256];
	const int rounds = (n + 255) / 256;
	for (int base = blockIdx.x * 256; base < rounds * 256; base += gridDim.x * 256)
	{
		const int i = base + threadIdx.x;
		const uint64_t key = i < n ? W.pairKey[i] : 0;
		int rank = 0;
		for (int t0 = 0; t0 < n; t0 += 256)
		{
			__syncthreads();
			// only first occurrences count; the others and the tail padding carry a key that is never smaller
			const bool in = t0 + threadIdx.x < n && W.pairFirst[t0 + threadIdx.x] != 0;
			tile[threadIdx.x] = in ? W.pairKey[t0 + threadIdx.x] : ~0ull;
			__syncthreads();
#pragma unroll 16
			for (int t = 0; t < 256; ++t) rank += tile[t] < key ? 1 : 0;
		}
		if (i < n) W.pairRank[i] = rank;
		// one add per wave, not per pair (hundreds of same-address atomics were most of this kernel's time)
		{
			const unsigned long long firsts = __ballot(i < n && W.pairFirst[i] != 0);
			if (waveLane() == 0 && firsts) atomicAdd(&S->c.nNewContacts, __popcll(firsts));
		}
	}
}

// Large sets: after the radix sort the keys are ordered; first-of-run flags + scan give the ranks.
__global__ __launch_bounds__(256) void k_pairs_sorted_first(DW W, const uint64_t* keys, int* nOut)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nPairs < W.capPairs ? S->c.nPairs : W.capPairs;
	// (any count: the radix path is also what a world takes that has been sorting large sets lately, without looking at the
	// count first - findNewContactsGraph; until round 5 a set the counting path could have ranked was left alone here)
	if (blockIdx.x == 0 && threadIdx.x == 0) *nOut = n;
	if (n <= 0) return;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		W.pairFirst[i] = (i == 0 || keys[i - 1] != keys[i]) ? 1 : 0;
	}
}

__global__ void k_pairs_sorted_total(DW W, const int* n2)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (*n2 > 0) S->c.nNewContacts = W.pairRank[*n2];
}

// Creation is all or nothing: if the new contacts of this update do not fit the contact array (or the counting path was
// given a set it does not handle), nothing is created, the moves stay buffered and Counters::overflow tells the host, which
// grows the arrays and runs the pair update again - the contacts then appear in one piece, in proxy-key order.
__device__ __forceinline__ bool createBlocked(const DW& W, const DState* S, int smallPath)
{
	if (smallPath && S->c.nPairs > COUNT_RANK_MAX) return true;
	// (the pair buffer overflowed, or the radix passes were queued for fewer pairs than there are: the host runs the update again)
	if (!smallPath && (S->c.overflow & 2) != 0) return true;
	return S->c.nContacts + S->c.nNewContacts > W.capContacts;
}

// b2ContactManager::ConsumeCreate / OnContactCreate (:488-564) + b2Contact::b2Contact (b2Contact.cpp:125-159)
// smallPath: ranks came from the counting kernels, which only run for n <= COUNT_RANK_MAX; a larger set is
// left untouched (moves stay buffered) and the host finishes it with the radix path after its read-back.
__global__ __launch_bounds__(256) void k_create_contacts(DW W, const uint64_t* keys, const int2* proxies, int smallPath)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nPairs < W.capPairs ? S->c.nPairs : W.capPairs;
	if (createBlocked(W, S, smallPath)) return;
	const int base = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		if (!W.pairFirst[i]) continue;
		const int dst = base + W.pairRank[i];
		if (dst >= W.capContacts)
		{
			atomicOr(&S->c.overflow, 1);
			continue;
		}
		int2 pr = proxies[i];
		int pA = pr.x, pB = pr.y; // A = lower proxy id (b2ContactManager.cpp:254)
		if (b2dContactSwap(W.shapes[W.p_shape[pA]].type, W.shapes[W.p_shape[pB]].type) == 1)
		{
			int t = pA;
			pA = pB;
			pB = t;
		}
		const int bodyA = W.p_body[pA], bodyB = W.p_body[pB];
		const bool sensor = ((W.p_filter1[pA] | W.p_filter1[pB]) & PF_SENSOR) != 0;
		uint32_t flags = CF_ENABLED | (sensor ? CF_SENSOR : 0u);
		if (isToiCandidate(W, pA, pB, bodyA, bodyB))
		{
			flags |= CF_TOI_CANDIDATE;
			// (listed for k_toi_order_create: few or none of an update's new contacts are TOI candidates as a rule - none of the
			// Tumbler's 150 000 per step - and one workgroup ballot-ranking all of them to find that out was 94 us of its step)
			const int k = atomicAdd(&S->c.nNewToiCand, 1);
			if (k < TOI_NEW_LIST_MAX) W.toiNewList[k] = dst;
		}
		float2 mA = W.p_mat[pA], mB = W.p_mat[pB];
		// b2MixFriction / b2MixRestitution (b2Contact.h:40-50)
		float friction = b2dSqrt(mA.x * mB.x);
		float restitution = mA.y > mB.y ? mA.y : mB.y;
		C.ids[dst] = make_int4(pA, pB, bodyA, bodyB);
		C.key[dst] = keys[i];
		C.flags[dst] = flags;
		C.mat[dst] = make_float4(friction, restitution, 0.0f, 1.0f);
		C.man0[dst] = make_float4(0, 0, 0, 0);
		C.man1[dst] = make_float4(0, 0, 0, 0);
		C.imp[dst] = make_float4(0, 0, 0, 0);
		C.man3[dst] = make_int4(0, 0, 0, 0);
		C.color[dst] = -1;
		C.mgr[dst] = -1;
		if (!sensor)
		{
			// SetAwake(true) on both bodies (:525-529), applied by k_apply_wake
			W.b_wake[bodyA] = 1;
			W.b_wake[bodyB] = 1;
		}
	}
}

__global__ __launch_bounds__(256) void k_create_finish(DW W, int smallPath)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (createBlocked(W, S, smallPath)) return;
	// apply the wake requests of contact creation now (the next user of the flags is the next step)
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if (W.b_wake[i])
		{
			W.b_wake[i] = 0;
			// (most bodies that get a new contact are awake with their timer at zero: nothing to write, and no mark on their row -
			// a moving field makes contacts for a tenth of its bodies per step, spread over every tile of the read-back)
			const uint32_t f = W.b_flags[i];
			if ((f & BF_AWAKE) != 0 && W.b_pos[i].w == 0.0f) continue;
			W.b_flags[i] = f | BF_AWAKE;
			W.b_pos[i].w = 0.0f;
			W.b_rowDirty[i] = 1;
		}
	}
}

// b2ContactManager::AddToContactArray (:659-686): a new TOI candidate takes the slot after the last
// candidate, in creation order. One workgroup, block scan over the new contacts.
__global__ __launch_bounds__(1024) void k_toi_order_create(DW W, int smallPath)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const bool blocked = createBlocked(W, S, smallPath);
	const int nNew = blocked ? 0 : S->c.nNewContacts;
	const int nCand = blocked ? 0 : S->c.nNewToiCand; // (counted and listed by k_create_contacts)
	if (nNew > 0 && nCand > 0 && nCand <= TOI_NEW_LIST_MAX)
	{
		// the candidates among the new contacts take their slots in creation order = index order: rank by index (the list is in
		// the order of the atomics that filled it)
		__shared__ int s_idx[TOI_NEW_LIST_MAX];
		const ContactArrays& C = W.ca[S->cur];
		const int tid = (int)threadIdx.x;
		const int count0 = S->c.nToiOrder;
		const int mine = tid < nCand ? W.toiNewList[tid] : 0x7fffffff;
		s_idx[tid] = mine;
		__syncthreads();
		if (tid < nCand)
		{
			int rank = 0;
			for (int k = 0; k < nCand; ++k) rank += s_idx[k] < mine ? 1 : 0;
			const int slot = count0 + rank;
			C.mgr[mine] = slot;
			W.toiPos2c[slot] = mine;
		}
		__syncthreads();
		if (tid == 0) S->c.nToiOrder = count0 + nCand;
	}
	else if (nNew > 0 && nCand > 0)
	{
		const int base = S->c.nContacts;
		const int cap = W.capContacts;
		const ContactArrays& C = W.ca[S->cur];
		// 1024 new contacts per round: wave ballots rank the candidates inside a wave, 16 wave totals are summed by every lane
		__shared__ int s_wave[2][16];
		const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
		int count = S->c.nToiOrder;
		int buf = 0;
		for (int i0 = 0; i0 < nNew; i0 += 1024, buf ^= 1)
		{
			const int i = base + i0 + tid;
			const bool flag = i0 + tid < nNew && i < cap && (C.flags[i] & CF_TOI_CANDIDATE) != 0;
			const unsigned long long m = __ballot(flag);
			if (lane == 0) s_wave[buf][wave] = __popcll(m);
			__syncthreads();
			int before = 0, total = 0;
			for (int k = 0; k < 16; ++k)
			{
				const int v = s_wave[buf][k];
				if (k < wave) before += v;
				total += v;
			}
			if (flag)
			{
				const int slot = count + before + __popcll(m & ((1ull << lane) - 1ull));
				C.mgr[i] = slot;
				W.toiPos2c[slot] = i;
			}
			count += total;
			// (the other buffer is written in the next round: one barrier per round is enough)
		}
		// (every lane has read Counters::nToiOrder and nContacts before lane 0 changes them below)
		__syncthreads();
		if (tid == 0) S->c.nToiOrder = count;
	}
	// the commit of the pair update (was a kernel of its own): the new contacts join the array, the move buffer is reset
	// (b2BroadPhase::ResetBuffers) - or nothing happens and the host is told to grow the arrays
	if (threadIdx.x == 0)
	{
		if (blocked)
		{
			if (S->c.nContacts + S->c.nNewContacts > W.capContacts) atomicOr(&S->c.overflow, 1);
		}
		else
		{
			S->c.nContacts = S->c.nContacts + S->c.nNewContacts;
			S->c.nMoves = 0;
		}
		S->c.nNewToiCand = 0;
	}
}

// ---- end of step -----------------------------------------------------------------------------------
// The read-back (b2hip_body_state rows, then the counters) goes straight into the host's pinned buffer `out` - no staging
// array, no copy behind the kernel: the rows of a workgroup's 256 bodies are transposed through LDS and leave as contiguous
// 16-byte stores; the workgroup that finishes last appends DState and, last of all, the sequence number the host polls
// (b2hip.hip: awaitState). Nothing else runs in the step after this kernel: the counters are final.
// skipIfRedo: the first read-back of b2hip_step_end. If the counters say that the host will finish the pair update with the
// radix path and read back AGAIN (more new pairs than the counting path ranks, or a pair-buffer overflow: the test
// stepEndImpl makes), only the counters travel now - the rows would be overwritten by the second read-back anyway, and at
// 1 M bodies they are 40 MB of PCIe. Counters::rowsSkipped tells the host (which reads back in full before it returns).
// mode END_STEP_LAZY (b2hip_set_lazy_readback): the step's house-keeping and the counters, but the rows stay on the device
// until somebody asks for a body's state; mode END_STEP_ROWS is that later fetch: rows and counters, nothing else.
#define END_STEP_FULL 0
#define END_STEP_SKIP_IF_REDO 1
#define END_STEP_LAZY 2
#define END_STEP_ROWS 3
// The rows travel where they differ from what the host's buffer holds: `shadow` is the device's copy of it (rowMode 2: a row
// is stored - to both - only if it differs from its shadow; 1: all rows are stored and the shadow is written afresh; 0: no
// shadow, `out` receives every row). A world at rest sends nothing.
// mode END_STEP_EARLY: the rows only, no house-keeping, no counters, nothing published - launched behind SynchronizeFixtures on
// a stream of its own with the shadow as `out` (b2hip.hip: startEarlyRows), followed by a copy of the shadow to the host that
// runs while the pair update and the TOI phase do; the launch at the end of the step then sends what the TOI phase changed
// since. Whatever the early launch read while TOI events were already moving bodies is either final or differs from the
// final row and is sent again: host buffer and shadow always hold the same words.
#define END_STEP_EARLY 4
#define END_STEP_TILE_ROWS 16 // (more changed rows than this in a tile of 256: the tile leaves whole, as coalesced 16-byte stores)
// marks (with rowMode 2, behind an early launch of this step): 1 - only tiles with a body whose row was written since
// SynchronizeFixtures are looked at (DW::b_rowDirty, set by whoever writes such a row: contact creation's wake-ups, the TOI
// phase; the compare of a million untouched rows with their shadow was 90 us of config 5's step); 2 - every tile is compared
// as with 0 and a row that differs without a mark raises Counters::overflow bit 12 (B2HIP_ROW_MARKS_CHECK=1: the proof that
// the marks are complete). Every launch but the early one clears the marks it saw.
__global__ __launch_bounds__(256) void k_end_step(DW W, int clearForces, const int* bar, float* out, int seq, int mode, float* shadow, int rowMode, int marks)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = W.nBodies;
	const int tid = (int)threadIdx.x;
	__shared__ __attribute__((aligned(16))) float s_out[2560];
	__shared__ int s_chg[256];
	__shared__ int s_pack[256 * 11];
	__shared__ int s_last;
	// (ClearPostSolveTOI only when the step is complete; sub-stepping: also for what earlier calls of the step touched)
	const bool toiEvents = (S->c.nToiEvents != 0 || W.toiContinue != 0) && S->c.toiIncomplete == 0;
	bool skipRows = false;
	int stored = 0; // (this lane has stored towards the host: the workgroup's fence below)
	const bool storeRows = mode != END_STEP_LAZY, houseKeeping = mode != END_STEP_ROWS && mode != END_STEP_EARLY;
	if (mode == END_STEP_SKIP_IF_REDO)
	{
		const int ov = S->c.overflow, moves = S->c.nMoves;
		const bool pairOverflow = (ov & 2) != 0 || ((ov & 1) != 0 && moves != 0);
		skipRows = (moves != 0 || pairOverflow) && (S->c.nPairs > COUNT_RANK_MAX || pairOverflow);
	}
	for (int base = blockIdx.x * 256; base < n && !skipRows; base += gridDim.x * 256)
	{
		const int i = base + tid;
		int mark = 0;
		if (mode != END_STEP_EARLY)
		{
			if (i < n)
			{
				mark = W.b_rowDirty[i];
				if (mark) W.b_rowDirty[i] = 0;
			}
			if (marks == 1 && __syncthreads_or(mark) == 0) continue; // (uniform: nobody wrote a row of this tile since the early launch read it)
			if (marks == 1 && tid == 0) atomicAdd(&S->c.endBlocksDone, 1);
		}
		if (i < n)
		{
			uint32_t f = W.b_flags[i];
			if (marks == 2 && !mark && W.b_pos0[i].w != 0.0f) atomicOr(&S->c.overflow, 0x1000);
			if (clearForces && houseKeeping) W.b_force[i] = make_float4(0, 0, 0, 0);
			// b2ClearBodySolveTOIFlags (b2World.cpp:239-259, k_toi_clear): sweeps go back to alpha0 = 0 for the next step
			// (only where an event advanced the sweep: a 4-byte store into every 16-byte element is a read-modify-write of every
			// line at the memory - 37 us of this kernel's 140 on a million bodies, against 16 MB read)
			if (toiEvents && houseKeeping && W.b_pos0[i].w != 0.0f) W.b_pos0[i].w = 0.0f;
			float4 xf = W.b_xf[i], p = W.b_pos[i], v = W.b_vel[i];
			float* o = s_out + tid * 10; // (filled in every mode: cheaper than a second predicate around ten LDS stores)
			o[0] = xf.x;
			o[1] = xf.y;
			o[2] = p.z;
			o[3] = v.x;
			o[4] = v.y;
			o[5] = v.z;
			o[6] = p.x;
			o[7] = p.y;
			o[8] = __uint_as_float(f & 0x7fu);
			o[9] = p.w;
		}
		if (W.spatial && !W.spFullRows && houseKeeping && W.spOwnOut != nullptr)
		{
			// a spatially sharded world with the lean exchange: this rank answers for the bodies it OWNS - their rows go to the
			// host packed (id + row), the table of all rows stays on the device until somebody asks (as with the read-back on
			// demand). Ownership need not follow the body ids: 1 / N of the rows cross PCIe, not all. One atomic per tile takes
			// the tile's place in the list (not one per wave: see k_sync_fixtures); the tile's rows are packed in LDS and leave
			// as ONE contiguous run of words (eleven 4-byte stores per lane into host memory were 150 us for the 50 000 bodies
			// of a rank of config 4).
			const bool mine = i < n && (W.b_flags[i] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[i] == (uint8_t)W.shardRank;
			const unsigned long long m = __ballot(mine);
			if (waveLane() == 0) s_chg[tid >> 6] = __popcll(m);
			__syncthreads();
			if (tid == 0)
			{
				const int c0 = s_chg[0], c1 = s_chg[1], c2 = s_chg[2], c3 = s_chg[3];
				s_chg[0] = 0; s_chg[1] = c0; s_chg[2] = c0 + c1; s_chg[3] = c0 + c1 + c2;
				s_chg[4] = c0 + c1 + c2 + c3;
				s_chg[5] = s_chg[4] > 0 ? atomicAdd(&S->c.spOwnRows, s_chg[4]) : 0;
			}
			__syncthreads();
			if (mine)
			{
				int* q = s_pack + (s_chg[tid >> 6] + __popcll(m & ((1ull << waveLane()) - 1ull))) * 11;
				const float* o = s_out + tid * 10;
				q[0] = i;
				for (int c = 0; c < 10; ++c) q[1 + c] = __float_as_int(o[c]);
			}
			__syncthreads();
			{
				const int total = s_chg[4], at = s_chg[5];
				const int fit = at + total <= W.spOwnCap ? total : (W.spOwnCap > at ? W.spOwnCap - at : 0);
				int* dstp = W.spOwnOut + (size_t)at * 11;
				for (int q = tid; q < fit * 11; q += 256) dstp[q] = s_pack[q];
			}
			__syncthreads(); // (s_chg serves the row comparison next)
		}
		if (!storeRows) continue; // (uniform over the workgroup: every lane leaves the tile before its barriers)
		const int cnt = (n - base < 256 ? n - base : 256) * 10; // floats of this tile; base * 40 bytes is 16-byte aligned
		float* dst = out + (size_t)base * 10;
		float* sh = shadow + (size_t)base * 10;
		bool whole = true;
		if (rowMode == 2)
		{
			// does this lane's row differ from what the host holds? Its own ten words (still in its part of s_out) against its
			// shadow row, 8 bytes at a time - no pass through LDS, no barrier but the one that counts the tile's changed rows
			bool changed = false;
			if (i < n)
			{
				const float2* o2 = (const float2*)(s_out + tid * 10);
				const float2* h2 = (const float2*)(shadow + (size_t)i * 10);
#pragma unroll
				for (int c = 0; c < 5; ++c)
				{
					const float2 a = o2[c], b = h2[c];
					changed = changed || __float_as_uint(a.x) != __float_as_uint(b.x) || __float_as_uint(a.y) != __float_as_uint(b.y);
				}
				if (marks == 2 && changed && !mark) atomicOr(&S->c.overflow, 0x1000);
			}
			const int nChanged = __syncthreads_count(changed ? 1 : 0);
			if (nChanged == 0) continue; // (uniform)
			whole = nChanged > END_STEP_TILE_ROWS;
			if (!whole)
			{
				if (changed)
				{
					// a few rows of the tile: each on its own (40 bytes, 8-byte aligned)
					const float2* o2 = (const float2*)(s_out + tid * 10);
					float2* d2 = (float2*)(out + (size_t)i * 10);
					float2* h2 = (float2*)(shadow + (size_t)i * 10);
					for (int c = 0; c < 5; ++c) { const float2 v = o2[c]; d2[c] = v; h2[c] = v; }
					stored = 1;
				}
				continue; // (uniform; every lane read its own part of s_out only)
			}
		}
		else __syncthreads();
		// the whole tile, as it lies in LDS: contiguous 16-byte stores (the barrier above made every lane's row visible)
		stored = 1;
		for (int q = tid; q < cnt / 4; q += 256)
		{
			const float4 v = ((const float4*)s_out)[q];
			((float4*)dst)[q] = v;
			if (rowMode != 0) ((float4*)sh)[q] = v;
		}
		for (int q = (cnt / 4) * 4 + tid; q < cnt; q += 256)
		{
			dst[q] = s_out[q];
			if (rowMode != 0) sh[q] = s_out[q];
		}
		__syncthreads();
	}
	if (mode == END_STEP_EARLY) return; // (the rows were all: the launch at the end of the step publishes)
	// ---- the last workgroup: counters, then the sequence number ---------------------------------------------------------
	// (every wave's stores have left; ONE system-scope fence per workgroup - a fence writes the L2 back - then arrive)
	// (... and only for a workgroup that HAS stored towards the host: behind an early launch most workgroups find no marked
	// row, and two thousand write-backs of the L2s in a row were 60 us of a kernel that had 4 MB to read)
	if (W.spatial && !W.spFullRows) stored = 1; // (the packed rows of a sharded world)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	const int anyStored = __syncthreads_or(stored);
	if (tid == 0)
	{
		if (anyStored) __threadfence_system();
		// (two-level arrival: 2 048 atomics on one word were 60 us of a million-body step's last kernel; b2d_world.h)
		unsigned t0 = 0u, t1 = 0u;
		s_last = b2dTreeArrive(W.arriveTree + (size_t)ARRIVE_END_STEP * TREE_WORDS, 0u, 0u, &t0, &t1) ? 1 : 0;
	}
	__syncthreads();
	if (!s_last) return;
	// (the solver's phase stamps travel with the counters)
	if (bar != nullptr && tid < 6) S->stamps[tid] = bar[8 + tid];
	if (tid == 0)
	{
		if (houseKeeping) S->phaseClock[13] = wall_clock64(); // end of the step, read-back included
		S->c.rowsSkipped = skipRows ? 1 : (storeRows ? 0 : 2);
	}
	__threadfence();
	__syncthreads();
	// (16 bytes per lane, as b2dPublishCensus does)
	const float4* src = (const float4*)S;
	float4* tail = (float4*)(out + B2D_STATE_TAIL(n));
	for (int k = tid; k < (int)(offsetof(DState, pubSeq) / 16); k += 256) tail[k] = b2dLoadAgent4(&src[k]);
	__threadfence_system();
	__syncthreads();
	if (tid == 0)
	{
		S->c.spOwnRows = 0; // (the packed rows of a spatially sharded world: counted afresh by the next read-back)
		S->c.endBlocksDone = 0;
		__hip_atomic_store(&((DState*)tail)->pubSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

#endif
