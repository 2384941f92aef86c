// b2d_kernels_island.h - island construction on the device.
//
// The reference finds islands with a serial DFS on the user thread (b2World.cpp:1207-1371). Here:
//  1. connected components by lock-free union-find over the touching-contact graph (membership is
//     order independent, so this is exact),
//  2. components are classified: SMALL (max(bodies, contacts) <= 128, no joints) or LARGE,
//  3. every SMALL island gets one lane that replays the reference's DFS from the reference's seed,
//     which yields the reference's body order and constraint order for that island, plus the
//     dependency level of every constraint (so the solver can run independent constraints of one
//     level in parallel and still be bit-identical to the sequential sweep),
//  4. LARGE islands only need membership (they are solved by graph colouring).
#ifndef B2D_KERNELS_ISLAND_H
#define B2D_KERNELS_ISLAND_H

#include "b2d_kernels_collide.h"
#include "b2d_wave.h"

#define ROOT_NONE 0
#define ROOT_SMALL 1
#define ROOT_LARGE 2
#define ROOT_FREE 4     // one body, no contact, no joint: stepped on the spot by k_island_classify (freeBodyStep)
#define ROOT_REMOTE 3   // solved by another rank of a sharded world (b2d_kernels_shard.h): in an island, but not on our lists

// Which rank solves the island rooted at `root` (the member of lowest ufPriority: the same body on every rank)
__device__ __forceinline__ int shardHashOwner(int root, int shardCount)
{
	return (int)((((uint32_t)root * 2654435761u) >> 12) % (uint32_t)shardCount);
}

__device__ __forceinline__ bool contactSolid(uint32_t flags)
{
	// b2World.cpp:1261-1273: enabled, touching, not a sensor
	// (CF_FOREIGN: a spatially sharded world - the contact's bodies are another rank's, its touching bit is not maintained here)
	return (flags & (CF_ENABLED | CF_TOUCHING | CF_SENSOR | CF_DESTROY | CF_FOREIGN)) == (CF_ENABLED | CF_TOUCHING);
}

__device__ __forceinline__ int ufFind(int* parent, int i)
{
	// find with path halving: every visited node is re-pointed at its grandparent. Any ancestor is a valid
	// parent at any time (links only ever go from a root to a body of lower ufPriority), so this is safe under concurrency
	// and keeps chains short when ten thousand bodies collapse into one component.
	int r = i;
	for (;;)
	{
		int p = __hip_atomic_load(&parent[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (p == r) break;
		int g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (g != p) __hip_atomic_store(&parent[r], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		r = p;
	}
	return r;
}

// Read-only find for the flatten pass. It must NOT halve paths: a concurrent halving store parent[i] = grandparent
// could land after the owner's final parent[i] = root and leave a non-root there (the body would then look like
// it belongs to no island). Without halving the only writer of parent[i] in that pass is lane i itself.
__device__ __forceinline__ int ufFindReadOnly(const int* parent, int i)
{
	int r = i;
	for (;;)
	{
		const int p = __hip_atomic_load(&parent[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (p == r) return r;
		r = p;
	}
}

// Link by a fixed pseudo-random priority (a bijective hash of the body id): the root of a component is its member of
// lowest priority whatever the interleaving, so labels are deterministic, and the expected tree depth is logarithmic.
// Linking by the id itself made the pyramid's trees as deep as the pyramid is high (every box links under a box of the
// row below, which was created earlier): 58 us of pointer chasing for 22k unions.
__device__ __forceinline__ uint32_t ufPriority(int i) { return (uint32_t)i * 2654435761u; }

__device__ __forceinline__ void ufUnionFrom(int* parent, int a, int b, int pa, int pb);

__device__ __forceinline__ void ufUnion(int* parent, int a, int b)
{
	ufUnionFrom(parent, a, b, __hip_atomic_load(&parent[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
		__hip_atomic_load(&parent[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// (pa, pb = parent[a], parent[b] as the caller has already read them - together with whatever else it had to fetch)
__device__ __forceinline__ void ufUnionFrom(int* parent, int a, int b, int pa, int pb)
{
	// Both walks in lockstep, ONE round of loads per level: a walk is a chain of dependent loads past the L2 (a dozen levels
	// when ten thousand bodies collapse into one component), and one walk after the other, with parent and grandparent
	// fetched in two rounds per level, made k_island_union 44 us on the 10 011-box pyramid. Path splitting: every body on
	// the way is re-pointed at its grandparent (any ancestor is a valid parent at any time, see ufFind).
	for (;;)
	{
		for (;;)
		{
			if (pa == pb || pa == b || pb == a) return; // (the walks met: same component)
			if (pa == a && pb == b) break;               // two roots
			const int ga = __hip_atomic_load(&parent[pa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const int gb = __hip_atomic_load(&parent[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (pa != a && ga != pa) __hip_atomic_store(&parent[a], ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (pb != b && gb != pb) __hip_atomic_store(&parent[b], gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			a = pa; pa = ga;
			b = pb; pb = gb;
		}
		// hang the root of higher priority value under the other
		const bool swap = ufPriority(a) > ufPriority(b);
		const int lo = swap ? b : a, hi = swap ? a : b;
		const int old = atomicCAS(&parent[hi], hi, lo);
		if (old == hi) return;
		// somebody else linked `hi` meanwhile: go on from where it points now
		if (swap) pa = old; else pb = old;
	}
}

// Called by block 0 of k_island_init (one launch less than a kernel of its own).
__device__ __forceinline__ void colorCheckBegin(const DW& W)
{
	if (threadIdx.x <= MAX_COLORS)
	{
		W.colorCount[colorSlot(threadIdx.x)] = 0;
		W.colorCursor[colorSlot(threadIdx.x)] = 0;
	}
	if (threadIdx.x == 0)
	{
		// colour compaction visits one class per step, from the highest down to 1, then starts over
		const int prev = W.st->c.nColors < MAX_COLORS ? W.st->c.nColors : MAX_COLORS;
		int t = W.st->c.compactCursor - 1;
		if (t < 1 || t >= prev) t = prev - 1;
		if (t < 0) t = 0;
		// a colouring that a whole round over its classes could not improve (a settled pile) is visited on every 8th step
		// only: the visit reads every constraint of the class (k_color_check lists them, k_color_small tries them: ~15 us
		// of the 10 011-box pyramid's step)
		W.st->c.compactTick += 1;
		// (the outcome of the last visit - whether or not k_color_small had anything to run for)
		if (W.st->c.compactClass > 0) W.st->c.compactIdle = W.st->c.compactMoved > 0 ? 0 : W.st->c.compactIdle + 1;
		W.st->c.compactMoved = 0;
		const bool idle = W.st->c.compactIdle >= prev && prev > 1;
		if (idle && (W.st->c.compactTick & 7) != 0)
		{
			W.st->c.compactClass = 0;
		}
		else
		{
			W.st->c.compactCursor = t;
			W.st->c.compactClass = t;
		}
		W.st->c.nCompact = 0;
		W.st->c.needRecolor = 0;
		W.st->c.nColors = 0;
		W.st->c.nUncolored = 0;
		W.st->c.nUncolList = 0;
	}
}

// Home block (+ 1) of a body this step: its own, or the one a neighbour offered it in k_island_edges (a body that joins a
// partitioned island is adopted by the block next to it; k_block_census makes that permanent). 0 = none.
// An offer of block `blk` (1 .. MAX_BLOCKS) to `body`: the offers of a body's neighbours are combined with atomicMax, and
// the winner should not be "the highest block id" - newcomers of a whole region then pile into one block (Pyramid 316 shedding
// boxes: a block gained 200 rows in a step and the partition was remade every fifth step) - so an offer carries a hash of
// (block, body) above the block id: which neighbour's block wins is fixed but spread evenly. Decode with adoptBlock().
__device__ __forceinline__ int adoptOffer(int blk, int body)
{
	uint32_t h = (uint32_t)blk * 2654435761u ^ (uint32_t)body * 40503u;
	h ^= h >> 15;
	h *= 0x2c1b3c6du;
	h ^= h >> 12;
	return (int)(((h & 0xfffffu) << 11) | (uint32_t)blk); // (20 bits of order, 11 bits of block id: positive)
}
__device__ __forceinline__ int adoptBlock(int offer) { return offer & 0x7ff; }

// A body of a large island that NO neighbour could offer a block - it landed on other newcomers, further from any block
// than k_block_adopt hands blocks on - takes one by its own id, as the newcomers that touch only the ground have since round 3
// (k_island_edges): its constraints are cut constraints wherever its neighbours live, a handful of hand-overs through
// memory. Until round 5 such a body's constraints were "orphans" and the whole island was partitioned again for them -
// radix passes, a colour check of every contact, a dozen claim / resolve rounds with read-backs, ~1 ms: the settled 50 086-box
// pyramid, whose rim sheds boxes onto a debris field all the time, did that every second step (`B2HIP_TRACE_PARTITION=1`).
__device__ __forceinline__ int effBlk(const DW& W, int body)
{
	const int b = W.b_blk1[body];
	if (b) return b;
	const int o = adoptBlock(W.b_adopt[body]);
	if (o || W.noOwnIdBlocks > 0) return o;
	if ((W.b_flags[body] & BF_LARGE) == 0) return 0;
	// (plain islands only - no joints, no hubs: what k_solve_blocks takes. Islands with joints or hubs go sweep by sweep through
	// k_blocks_sweep, which sweeps such newcomers' constraints in order with the hub rows - DW::serialOrphans - and never made
	// a partition for them; handing them blocks there instead ran MultithreadDemo's growing pile into the sweep kernel's spin
	// limit, not looked into)
	if ((W.nJoints != 0 || W.st->c.maxDegree > HUB_DEGREE) && W.noOwnIdBlocks >= 0) return 0; // (noOwnIdBlocks -1, B2HIP_OWN_ID_BLOCKS_ALL=1: everywhere - the experiment)
	const int nb = W.st->c.nBlocks < MAX_BLOCKS ? W.st->c.nBlocks : MAX_BLOCKS;
	return nb > 0 ? ownIdBlock(body, nb) : 0;
}

__global__ __launch_bounds__(256) void k_island_init(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (blockIdx.x == 0) colorCheckBegin(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MAX_BLOCKS + 1; i += gridDim.x * blockDim.x)
	{
		W.blkRows[(size_t)i * BLK_SLOT] = 0;
		W.blkCursor[(size_t)i * BLK_SLOT] = 0;
		W.blkBodyCount[(size_t)i * BLK_SLOT] = 0;
		W.blkBodyCursor[(size_t)i * BLK_SLOT] = 0;
	}
	const int n = W.nBodies;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		W.parent[i] = i;
		// (rootSeed / rootBodies start with the body's OWN contribution, below: a body that stays its own root - 97 % of the
		// bodies of the 1 M field - then sends no atomic at all in k_island_flatten; only members add to their root)
		W.rootContacts[i] = 0;
		W.rootJoints[i] = 0;
		W.rootIsland[i] = ROOT_NONE;
		W.deg[i] = 0;
		W.adjCursor[i] = 0;
		W.b_slot[i] = -1;
		W.b_island[i] = -1;
		for (int k = 0; k < ROOT_PEN_SLOTS; ++k) W.rootPen[(size_t)k * W.nBodies + i] = 0;
		W.rootDone[i] = 0;
		W.rootSleepMin[i] = 0x7f7fffffu; // b2_maxFloat
		W.rootJointCursor[i] = 0;
		W.rootJointOkay[i] = 1;
		W.bodyClaim[i] = 0;
		W.bodyColorMask[i] = 0;
		W.bodyActive[i] = 0;
		W.bodyRest[i] = 0;
		W.b_adopt[i] = 0;
		for (int q = 0; q < 3; ++q) W.b_adoptStage[(size_t)q * W.nBodies + i] = 0;
		uint32_t f = W.b_flags[i] & ~(BF_ISLAND | BF_LARGE);
		// ConsumeAwakes / b2Contact::Destroy wake-ups gathered by collide: SetAwake(true) also
		// resets the sleep timer of bodies that are already awake (b2Body.h:699-703).
		if (W.b_wake[i])
		{
			f |= BF_AWAKE;
			W.b_pos[i].w = 0.0f;
			W.b_wake[i] = 0;
		}
		W.b_flags[i] = f;
		// (a spatially sharded world: another rank's body is in none of OUR islands - b2d_kernels_spatial.h)
		const bool member = (f & BF_TYPE_MASK) != BT_STATIC && (f & BF_ACTIVE) != 0 && !(W.spatial && W.b_owner[i] != (uint8_t)W.shardRank);
		W.rootBodies[i] = member ? 1 : 0;
		// seeds are taken in m_nonStaticBodies order (b2World.cpp:1207-1221): first awake, active body
		W.rootSeed[i] = member && (f & BF_AWAKE) != 0 ? W.b_order[i] : 0x7fffffff;
	}
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x) W.joints[j].islandFlag = 0;
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		S->c.nSIslands = S->c.nSBodies = S->c.nSContacts = S->c.nSW = S->c.nChunks = 0;
		S->c.nLIslands = S->c.nLBodies = S->c.nLContacts = 0;
		S->c.nColors = 0;
		S->c.nIslands = 0;
		S->c.posItersLarge = 0;
		S->c.maxSmallW = 0;
		S->c.maxDegree = 0;
		S->c.maxDegreePlain = 0;
		W.hubMeta[0] = 0ull; // (the primary hub of this step: k_island_flatten)
		S->c.hubEpoch += 1;
		S->c.chunkW = SMALL_ISLAND_MAX_W;
		S->c.partitionAge += 1;
		if (S->c.partitionCooldown > 0) S->c.partitionCooldown -= 1;
		S->c.nBigIslands = 0;
		for (int r = 0; r < SHARD_MAX_RANKS; ++r) S->c.shardBodies[r] = S->c.shardContacts[r] = S->c.shardJoints[r] = 0;
		S->c.shardCursor[0] = S->c.shardCursor[1] = S->c.shardCursor[2] = 0;
		S->c.nRemoteIslands = 0;
		S->c.nSmallJointed = 0;
		S->c.nFreeIslands = 0;
		S->c.nSerialOrphans = 0;
		S->c.hubRounds = 0;
		S->c.hubSerialChunks = 0;
		S->c.nOrphanRows = 0;
		S->c.blkMaxRows = 0;
		S->c.blkMaxBodies = 0;
		S->c.nCutRows = 0;
		S->c.colorMaskLo = S->c.colorMaskHi = 0u;
	}
}

// The SOLID contacts of a tile of `rounds` x 256 contacts, gathered into LDS (round 6). The island build's three passes over the
// contacts do their work for the solid ones only - one contact in seven of the settled 100 000-box Tumbler (2.6 M fat-AABB
// pairs, 370 000 of them touching) - and a lane per contact leaves their waves with nine lanes of 64 walking the
// union-find trees (chains of dependent loads past the L2) while the rest sit idle: as many waves in flight as for 2.6 M
// walks. Here a workgroup reads the flags of its tile with `rounds` independent loads per lane, lists the solid contacts in
// LDS (one LDS atomic per wave and round; which lane gets which contact does not matter to any of the three) and works
// through the list with full waves. `rounds` is the host's (b2hip_host_phases.h: 8 for millions of contacts, 1 - a lane
// per contact as before, the list then only squeezes the gaps out of a tile - where that would leave workgroups without work).
#define SOLID_TILE_ROUNDS_MAX 8
struct SolidTile
{
	int n;
	int list[SOLID_TILE_ROUNDS_MAX * 256];
};
template <bool CLEAR_ISLAND_BIT>
__device__ __forceinline__ int solidTileGather(const ContactArrays& C, int n, int tileBase, int rounds, SolidTile* t)
{
	if (threadIdx.x == 0) t->n = 0;
	__syncthreads();
	uint32_t fl[SOLID_TILE_ROUNDS_MAX];
#pragma unroll
	for (int k = 0; k < SOLID_TILE_ROUNDS_MAX; ++k)
	{
		const int i = tileBase + k * 256 + (int)threadIdx.x;
		fl[k] = (k < rounds && i < n) ? C.flags[i] : (uint32_t)CF_DESTROY;
	}
#pragma unroll
	for (int k = 0; k < SOLID_TILE_ROUNDS_MAX; ++k)
	{
		if (k >= rounds) break;
		const int i = tileBase + k * 256 + (int)threadIdx.x;
		// (b2World::Solve clears the island flag of every contact, b2World.cpp:1191-1194: stored where it was set)
		if (CLEAR_ISLAND_BIT && (fl[k] & CF_ISLAND) != 0 && i < n) C.flags[i] = fl[k] & ~CF_ISLAND;
		const bool solid = contactSolid(fl[k]);
		const unsigned long long m = __ballot(solid);
		int base = 0;
		if (waveLane() == 0 && m) base = atomicAdd(&t->n, __popcll(m));
		base = __shfl(base, 0);
		if (solid) t->list[base + __popcll(m & ((1ull << waveLane()) - 1ull))] = i;
	}
	__syncthreads();
	return t->n;
}

__global__ __launch_bounds__(256) void k_island_union(DW W, int rounds)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	__shared__ SolidTile s_tile;
	const int tile = rounds * 256;
	for (int base = blockIdx.x * tile; base < n; base += gridDim.x * tile)
	{
		const int m = solidTileGather<true>(C, n, base, rounds, &s_tile);
		for (int j0 = 0; j0 < m; j0 += 256)
		{
			const int j = j0 + (int)threadIdx.x;
			const bool live = j < m;
			const int i = live ? s_tile.list[j] : 0;
			int4 ids = live ? C.ids[i] : make_int4(0, 0, 0, 0);
			// (the first level of both walks is fetched together with the body flags: one round trip less in front of the walks)
			const uint32_t bfA = W.b_flags[ids.z], bfB = W.b_flags[ids.w];
			const int pa = __hip_atomic_load(&W.parent[ids.z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const int pb = __hip_atomic_load(&W.parent[ids.w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const bool nsA = live && (bfA & BF_TYPE_MASK) != BT_STATIC;
			const bool nsB = live && (bfB & BF_TYPE_MASK) != BT_STATIC;
			// (the count a body's atomic returns is the contact's place in the body's adjacency segment: k_island_edges fills the
			// segments with plain stores - for every body, large islands included: k_color_masks walks them. Neighbouring lanes
			// with the same body share one atomic: b2d_wave.h, waveRunAlloc. Not what this kernel waits for, though: issuing all
			// 750 000 of the settled Tumbler's a second time on a scratch array changed nothing - docs/KERNEL_NOTES.md.)
			int2 slot = make_int2(-1, -1);
			const int sa = waveRunAlloc(W.deg, ids.z, nsA), sb = waveRunAlloc(W.deg, ids.w, nsB);
			if (nsA) slot.x = sa;
			if (nsB) slot.y = sb;
			if (nsA && nsB) ufUnionFrom(W.parent, ids.z, ids.w, pa, pb);
			if (live) W.adjSlot[i] = slot;
		}
		__syncthreads(); // (the list is the next tile's)
	}
	// joints connect bodies too (b2World.cpp:1292-1318)
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x)
	{
		const RevoluteJoint& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		// (a joint to an inactive body is skipped by the traversal: b2World.cpp:1303-1307)
		if (((W.b_flags[jn.bodyA] & W.b_flags[jn.bodyB]) & BF_ACTIVE) == 0) continue;
		bool nsA = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC;
		bool nsB = (W.b_flags[jn.bodyB] & BF_TYPE_MASK) != BT_STATIC;
		if (W.spatial && ((nsA && W.b_owner[jn.bodyA] != (uint8_t)W.shardRank) || (nsB && W.b_owner[jn.bodyB] != (uint8_t)W.shardRank))) continue;
		if (nsA && nsB) ufUnion(W.parent, jn.bodyA, jn.bodyB);
	}
}

__global__ __launch_bounds__(256) void k_island_flatten(DW W)
{
	b2dPhaseStamp(W);
	const int n = W.nBodies;
	// census of the largest contact degree (k_island_union counted deg[]) - and WHOSE it is when it makes a hub: the primary
	// hub, whose constraints k_sweep_end takes as one fixed point (ties: the higher body id) - and the busiest body below the
	// hub threshold: what a colouring cannot go under (the host's measure of how far the colours in use have crept,
	// b2hip_host_phases.h). Kept per lane over the loop and offered ONCE per workgroup: an offer is a load past the L2 of a
	// word every workgroup looks at (two such loads per wave and round were 40 us of this kernel on a million bodies).
	int dg = 0, who = -1, dgPlain = 0;
	for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x)
	{
		const int i = base + threadIdx.x;
		bool valid = i < n;
		uint32_t f = valid ? W.b_flags[i] : 0u;
		valid = valid && (f & BF_TYPE_MASK) != BT_STATIC && (f & BF_ACTIVE) != 0 && !(W.spatial && W.b_owner[i] != (uint8_t)W.shardRank);
		int r = 0;
		if (valid)
		{
			r = ufFindReadOnly(W.parent, i);
			__hip_atomic_store(&W.parent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const int d = W.deg[i];
			if (d > dg || (d == dg && i > who)) { dg = d; who = i; }
			if (d <= HUB_DEGREE && d > dgPlain) dgPlain = d;
		}
		// (a root has counted itself and offered its own seed in k_island_init: only the other members add to it)
		const bool other = valid && r != i;
		waveAtomicAddInt(W.rootBodies, r, 1, other);
		waveAtomicMinInt(W.rootSeed, r, other ? W.b_order[i] : 0, other && (f & BF_AWAKE) != 0);
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		const int o = __shfl_xor(dg, off), ow = __shfl_xor(who, off), op = __shfl_xor(dgPlain, off);
		if (o > dg || (o == dg && ow > who)) { dg = o; who = ow; }
		dgPlain = op > dgPlain ? op : dgPlain;
	}
	__shared__ int s_dg[4], s_who[4], s_plain[4];
	if (waveLane() == 0) { s_dg[threadIdx.x >> 6] = dg; s_who[threadIdx.x >> 6] = who; s_plain[threadIdx.x >> 6] = dgPlain; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		for (int k = 1; k < 4; ++k)
		{
			if (s_dg[k] > dg || (s_dg[k] == dg && s_who[k] > who)) { dg = s_dg[k]; who = s_who[k]; }
			dgPlain = s_plain[k] > dgPlain ? s_plain[k] : dgPlain;
		}
		if (dgPlain > 0) atomicMaxIfAbove(&W.st->c.maxDegreePlain, dgPlain);
		if (dg > 0)
		{
			atomicMaxIfAbove(&W.st->c.maxDegree, dg);
			if (dg > HUB_DEGREE)
			{
				const unsigned long long key = ((unsigned long long)(uint32_t)dg << 32) | (unsigned long long)(uint32_t)who;
				if (key > __hip_atomic_load(&W.hubMeta[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&W.hubMeta[0], key);
			}
		}
	}
}

__global__ __launch_bounds__(256) void k_island_count(DW W, int rounds)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	__shared__ BlockHot s_hot; // (the workgroup's count for the first root it meets: b2d_wave.h)
	__shared__ SolidTile s_tile;
	blockHotInit(&s_hot);
	const int tile = rounds * 256;
	for (int base = blockIdx.x * tile; base < n; base += gridDim.x * tile)
	{
		const int m = solidTileGather<false>(C, n, base, rounds, &s_tile);
		for (int j0 = 0; j0 < m; j0 += 256)
		{
			const int j = j0 + (int)threadIdx.x;
			bool valid = j < m;
			int root = 0;
			if (valid)
			{
				int4 ids = C.ids[s_tile.list[j]];
				int b = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC ? ids.z : ids.w;
				valid = (W.b_flags[b] & BF_TYPE_MASK) != BT_STATIC;
				if (valid) root = W.parent[b];
			}
			blockHotAddInt(&s_hot, W.rootContacts, root, 1, valid);
		}
		__syncthreads(); // (the list is the next tile's)
	}
	blockHotFlush(&s_hot, W.rootContacts);
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x)
	{
		const RevoluteJoint& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		if (((W.b_flags[jn.bodyA] & W.b_flags[jn.bodyB]) & BF_ACTIVE) == 0) continue;
		int b = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC ? jn.bodyA : jn.bodyB;
		if ((W.b_flags[b] & BF_TYPE_MASK) == BT_STATIC) continue;
		atomicAdd(&W.rootJoints[W.parent[b]], 1);
	}
}

// An island of ONE body that touches nothing and hangs on no joint - most bodies of a sparse world (97 % of the islands of
// the 1 M-body field, the boxes in flight in the Tumbler): b2Island::Solve (b2Island.cpp:184-396) reduces to integrating
// the body and its sleep timer, done here with the arithmetic k_solve_small performs for such a body (same helpers, same
// order), instead of sending it through the island traversal, the chunk tables and a lane of the LDS solver.
__device__ __forceinline__ void freeBodyStep(const DW& W, const StepParams& sp, int body)
{
	const uint32_t bflags = W.b_flags[body];
	const float4 pos = W.b_pos[body], vel = W.b_vel[body], massv = W.b_mass[body];
	float sleepTime = pos.w;
	W.b_pos0[body] = make_float4(pos.x, pos.y, pos.z, 0.0f);
	V2 v = v2(vel.x, vel.y);
	float w = vel.z;
	if ((bflags & BF_TYPE_MASK) == BT_DYNAMIC)
	{
		const float4 damp = W.b_damp[body], force = W.b_force[body];
		b2dIntegrateVelocity(&v, &w, sp.dt, sp.gravity, damp.z, massv.x, massv.y, v2(force.x, force.y), force.z, damp.x, damp.y);
	}
	V2 c = v2(pos.x, pos.y);
	float a = pos.z;
	b2dIntegratePosition(&c, &a, &v, &w, sp.dt);
	// (b2dXfFromSweep with the rotation expanded in place: a call would give this streaming kernel a stack)
	Xf xf;
	xf.q = b2dRotInline(a);
	xf.p = c - b2dMulRV(xf.q, v2(massv.z, massv.w));
	W.b_xf[body] = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
	uint32_t f = bflags | BF_ISLAND | BF_AWAKE;
	if (sp.allowSleep)
	{
		const float linTolSqr = B2D_LINEAR_SLEEP_TOL * B2D_LINEAR_SLEEP_TOL;
		const float angTolSqr = B2D_ANGULAR_SLEEP_TOL * B2D_ANGULAR_SLEEP_TOL;
		if ((bflags & BF_AUTOSLEEP) == 0 || w * w > angTolSqr || b2dDot(v, v) > linTolSqr) sleepTime = 0.0f;
		else sleepTime += sp.dt;
		// (no constraint: the first position iteration finds the island solved - if there is one)
		if (sleepTime >= B2D_TIME_TO_SLEEP && sp.posIters > 0)
		{
			f &= ~BF_AWAKE;
			sleepTime = 0.0f;
			v = v2(0, 0);
			w = 0.0f;
			W.b_force[body] = make_float4(0, 0, 0, 0);
		}
	}
	W.b_flags[body] = f;
	W.b_pos[body] = make_float4(c.x, c.y, a, sleepTime);
	W.b_vel[body] = make_float4(v.x, v.y, w, 0.0f);
}

__global__ __launch_bounds__(256) void k_island_classify(DW W, int forceLarge, StepParams sp)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = W.nBodies;
	int nIslands = 0, nFree = 0, maxW = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		int4 in = make_int4(0, 0, 0, 0);
		uint32_t f = W.b_flags[i];
		if ((f & BF_TYPE_MASK) != BT_STATIC && (f & BF_ACTIVE) && W.parent[i] == i)
		{
			bool solved = W.rootSeed[i] != 0x7fffffff;
			if (solved)
			{
				++nIslands;
				int nb = W.rootBodies[i], nc = W.rootContacts[i], nj = W.rootJoints[i];
				int w = nb > nc ? nb : nc;
				if (w < 1) w = 1;
				bool mine = true, big = false;
				const bool isFree = nb == 1 && nc == 0 && nj == 0 && forceLarge != 2 && !W.noFreeBodies;
				if (isFree)
				{
					// (in a sharded world every rank steps these itself: nothing to exchange)
					freeBodyStep(W, sp, i);
					W.rootIsland[i] = ROOT_FREE;
					++nFree;
				}
				else
				if (W.shardCount > 1 && !W.spatial) // (island-owner sharding; spatial ownership has sorted the islands out already)
				{
					// islands are dealt over the ranks: the big ones one by one (k_shard_big), the others by a hash of their root
					big = nb > SHARD_BIG_BODIES && forceLarge != 2;
					if (big)
					{
						const int k = atomicAdd(&S->c.nBigIslands, 1);
						if (k < SHARD_BIG_MAX) W.bigRoots[k] = i; else big = false;
					}
					if (!big)
					{
						const int owner = shardHashOwner(i, W.shardCount);
						mine = owner == W.shardRank;
						// (every rank keeps every rank's census: the slabs of the exchange are sized from it)
						atomicAdd(&S->c.shardBodies[owner], nb);
						if (nc) atomicAdd(&S->c.shardContacts[owner], nc);
						if (nj) atomicAdd(&S->c.shardJoints[owner], nj);
					}
				}
				if (isFree) // (not a load of the word just stored: the wave would wait for the store to come back)
				{
				}
				else if (big)
				{
					W.rootIsland[i] = ROOT_LARGE; // k_shard_big keeps it, or hands it to another rank
				}
				else if (!mine)
				{
					W.rootIsland[i] = ROOT_REMOTE;
					atomicAdd(&S->c.nRemoteIslands, 1);
				}
				else if ((nj <= SMALL_ISLAND_MAX_JOINTS && w <= W.smallMaxW && forceLarge == 0) || forceLarge == 2)
				{
					W.rootIsland[i] = ROOT_SMALL;
					if (nj > 0) atomicAdd(&S->c.nSmallJointed, 1);
					in = make_int4(nb, nc, w, 1);
					maxW = w > maxW ? w : maxW; // (offered once per workgroup, below)
				}
				else
				{
					W.rootIsland[i] = ROOT_LARGE;
					int k = atomicAdd(&S->c.nLIslands, 1);
					W.li_roots[k] = i;
				}
			}
		}
		W.rootScanIn[i] = in;
	}
	// (the census travels with the arrival of the workgroups: even one add per workgroup and counter - 4 096 atomics on two
	// words - was 120 of this kernel's 205 us on the 1 M field; b2d_world.h: b2dTreeArrive)
	blockAtomicMaxIfAbove(&S->c.maxSmallW, maxW);
	b2dBlockTreeAdd2(W, ARRIVE_CLASSIFY, &S->c.nIslands, nIslands, &S->c.nFreeIslands, nFree, (unsigned)W.nBodies <= TREE_SUM_MAX);
}

__global__ __launch_bounds__(256) void k_island_assign(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = W.nBodies;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		uint32_t f = W.b_flags[i];
		if ((f & BF_TYPE_MASK) == BT_STATIC || (f & BF_ACTIVE) == 0) continue;
		int r = W.parent[i];
		int tier = W.rootIsland[r];
		if (tier == ROOT_NONE || tier == ROOT_FREE) continue;
		// the DFS makes every visited body awake without touching its sleep timer (b2World.cpp:1243-1244)
		f |= BF_ISLAND | BF_AWAKE;
		if (tier == ROOT_LARGE)
		{
			f |= BF_LARGE;
			int k = atomicAdd(&S->c.nLBodies, 1);
			W.li_bodies[k] = i;
		}
		W.b_flags[i] = f;
		if (r == i && tier == ROOT_SMALL)
		{
			int4 sc = W.rootScanOut[i];
			int idx = sc.w;
			W.si_root[idx] = i;
			W.si_bodyStart[idx] = sc.x;
			W.si_contactStart[idx] = sc.y;
			W.si_wStart[idx] = sc.z;
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		int4 tot = W.rootScanOut[n];
		S->c.nSBodies = tot.x;
		S->c.nSContacts = tot.y;
		S->c.nSW = tot.z;
		S->c.nSIslands = tot.w;
		W.si_bodyStart[tot.w] = tot.x;
		W.si_contactStart[tot.w] = tot.y;
		W.si_wStart[tot.w] = tot.z;
		// A chunk takes the islands whose weight prefix falls into one window of chunkW; it can overshoot the window
		// by at most one island, so chunkW = lanes - largest island packs the workgroup as full as possible (the more
		// islands a chunk holds, the more lanes each dependency level keeps busy).
		const int lanes = (S->c.maxSmallW <= TINY_ISLAND_MAX_W && !W.bigChunks) ? TINY_CHUNK_LANES : SMALL_CHUNK_LANES;
		int chunkW = lanes - S->c.maxSmallW;
		if (chunkW < lanes / 2) chunkW = lanes / 2;
		S->c.chunkW = chunkW;
		S->c.chunkLanes = lanes;
		S->c.nChunks = tot.w > 0 ? (tot.z - 1) / chunkW + 1 : 0;
	}
}

// Adjacency (small islands) and the large-island contact list. `pub`: this is the last kernel of the island build (no large
// islands lately, no joints): its last workgroup publishes the census for the host (b2dPublishCensus).
__global__ __launch_bounds__(256) void k_island_edges(DW W, DState* pub, int rounds)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	__shared__ SolidTile s_tile;
	const int tile = rounds * 256;
	for (int base = blockIdx.x * tile; base < n; base += gridDim.x * tile)
	{
	const int m = solidTileGather<false>(C, n, base, rounds, &s_tile);
	for (int j0 = 0; j0 < m; j0 += 256)
	{
		const int j = j0 + (int)threadIdx.x;
		const int i = j < m ? s_tile.list[j] : 0;
		int4 ids = make_int4(0, 0, 0, 0);
		bool nsA = false, nsB = false;
		int tier = ROOT_NONE;
		if (j < m)
		{
			ids = C.ids[i];
			nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
			nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
			if (nsA || nsB) tier = W.rootIsland[W.parent[nsA ? ids.z : ids.w]];
		}
		if (nsA || nsB)
		{
			// every body's solid contacts, grouped by body (the places were handed out by k_island_union's degree count): the DFS
			// of the small islands and the colour masks of all bodies read them
			const int2 slot = W.adjSlot[i];
			if (nsA) W.adj[W.adjStart[ids.z] + slot.x] = i;
			if (nsB) W.adj[W.adjStart[ids.w] + slot.y] = i;
		}
		// the large islands' contact list: one cursor for the whole world - a slot per workgroup and iteration, not per wave
		// (the settled Tumbler: 32 000 atomics on that word, 200 us; b2d_wave.h)
		const int k = blockAlloc(&S->c.nLContacts, tier == ROOT_LARGE);
		if (tier == ROOT_LARGE)
		{
			W.li_contacts[k] = i;
			// a body without a home block is offered the block of its neighbour (if several do: the one whose offer ranks highest, adoptOffer)
			if (nsA != nsB)
			{
				// A body without a home block that touches only static bodies gets no offer from anybody (a box the pile shed and
				// that landed on the ground): its constraints would be orphans and the whole partition would be remade for it. It
				// takes a block by its own id - the constraint with the static body is interior wherever the body lives - as an
				// offer of the lowest rank, so that any neighbour's offer still wins.
				const int D = nsA ? ids.z : ids.w;
				const int nb = S->c.nBlocks < MAX_BLOCKS ? S->c.nBlocks : MAX_BLOCKS;
				if (nb > 0 && W.b_blk1[D] == 0)
				{
					const int offer = ownIdBlock(D, nb); // (rank bits 0)
					atomicMax(&W.b_adopt[D], offer);
					for (int q = 0; q < 3; ++q) atomicMax(&W.b_adoptStage[(size_t)q * W.nBodies + D], offer);
				}
			}
			if (nsA && nsB)
			{
				// (the offers also go to the three stage buffers of k_block_adopt, which hands blocks on to bodies further away)
				const int ba = W.b_blk1[ids.z], bb = W.b_blk1[ids.w];
				if (ba == 0 && bb != 0)
				{
					const int offer = adoptOffer(bb, ids.z);
					atomicMax(&W.b_adopt[ids.z], offer);
					for (int q = 0; q < 3; ++q) atomicMax(&W.b_adoptStage[(size_t)q * W.nBodies + ids.z], offer);
				}
				if (bb == 0 && ba != 0)
				{
					const int offer = adoptOffer(ba, ids.w);
					atomicMax(&W.b_adopt[ids.w], offer);
					for (int q = 0; q < 3; ++q) atomicMax(&W.b_adoptStage[(size_t)q * W.nBodies + ids.w], offer);
				}
			}
		}
	}
	__syncthreads(); // (the list is the next tile's)
	}
	// (the counters the census consists of are atomics' results: nothing of this workgroup's plain stores is read there)
	if (pub != nullptr && b2dLastBlockArrive(W, ARRIVE_EDGES)) b2dPublishCensus(W, pub);
}

// The census published by a launch of its own (the island build ends with another kernel than the two that can do it).
__global__ __launch_bounds__(256) void k_publish_census(DW W, DState* pub)
{
	b2dPhaseStamp(W);
	b2dPublishCensus(W, pub);
}

// A body that joins a partitioned island without touching a body that has a home block - it landed on other newcomers -
// gets no offer from k_island_edges; its constraints would be orphans and the island would have to be partitioned again
// (every step, while a pile is growing). Stage `stage` (0, 1, 2) hands blocks on by one more contact: it reads what stage
// - 1 knew (stage buffer `stage`; buffer 0 holds k_island_edges' offers) and adds its offers to every later buffer and to
// b_adopt, so the outcome does not depend on which lane ran first.
__global__ __launch_bounds__(256) void k_block_adopt(DW W, int stage)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLContacts;
	const ContactArrays& C = W.ca[S->cur];
	const int* known = W.b_adoptStage + (size_t)stage * W.nBodies;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[W.li_contacts[k]];
		if ((W.b_flags[ids.z] & BF_TYPE_MASK) == BT_STATIC || (W.b_flags[ids.w] & BF_TYPE_MASK) == BT_STATIC) continue;
		// (what each end knows so far: its home block, or the block of the best offer it has had up to the stage before)
		const int ba = W.b_blk1[ids.z] ? W.b_blk1[ids.z] : adoptBlock(known[ids.z]);
		const int bb = W.b_blk1[ids.w] ? W.b_blk1[ids.w] : adoptBlock(known[ids.w]);
		if (W.b_blk1[ids.z] == 0 && bb != 0 && bb != ba)
		{
			const int offer = adoptOffer(bb, ids.z);
			atomicMax(&W.b_adopt[ids.z], offer);
			for (int q = stage + 1; q < 3; ++q) atomicMax(&W.b_adoptStage[(size_t)q * W.nBodies + ids.z], offer);
		}
		if (W.b_blk1[ids.w] == 0 && ba != 0 && ba != bb)
		{
			const int offer = adoptOffer(ba, ids.w);
			atomicMax(&W.b_adopt[ids.w], offer);
			for (int q = stage + 1; q < 3; ++q) atomicMax(&W.b_adoptStage[(size_t)q * W.nBodies + ids.w], offer);
		}
	}
}

// One lane per small island: the reference's DFS (b2World.cpp:1223-1319), restricted to what can
// change the outcome: static bodies are never pushed (they would be popped and skipped without
// changing the relative order of anything else) and non-solid contacts are not in the adjacency.
__global__ __launch_bounds__(64) void k_island_dfs(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int nS = S->c.nSIslands;
	const ContactArrays& C = W.ca[S->cur];
	if (blockIdx.x == 0 && threadIdx.x == 0) S->gapClock[1] = wall_clock64();
	for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nS; idx += gridDim.x * blockDim.x)
	{
		{
			// chunk table of the solver (was a kernel of its own): chunkFirst[c] = first small island whose weight prefix falls
			// into window c of chunkW
			const int chunkW = S->c.chunkW;
			const int c = W.si_wStart[idx] / chunkW;
			if (idx == 0 || W.si_wStart[idx - 1] / chunkW != c) W.chunkFirst[c] = idx;
			// The census counted windows up to the END of the last island; a window in which no island STARTS has no chunk
			// (its chunkFirst entry would be stale). Only the trailing window can be empty (islands are at most one window wide).
			if (idx == nS - 1) S->c.nChunks = c + 1;
		}
		const int root = W.si_root[idx];
		const int seed = W.orderBody[W.rootSeed[root]];
		const int bStart = W.si_bodyStart[idx];
		const int cStart = W.si_contactStart[idx];
		int* stack = W.si_stack + bStart;
		int sp = 0, nb = 0, nc = 0, nj = 0;
		stack[sp++] = seed;
		W.b_slot[seed] = -2;
		while (sp > 0)
		{
			const int b = stack[--sp];
			const int slot = bStart + nb;
			W.si_bodies[slot] = b;
			W.b_slot[b] = slot;
			W.b_island[b] = idx;
			W.si_lastLevel[slot] = 0;
			++nb;
			const int s = W.adjStart[b];
			const int e = s + W.deg[b];
			// the contact list is newest first (b2ContactManager.cpp:531-553) == descending index
			for (int k = s + 1; k < e; ++k)
			{
				int v = W.adj[k];
				int j = k - 1;
				while (j >= s && W.adj[j] < v)
				{
					W.adj[j + 1] = W.adj[j];
					--j;
				}
				W.adj[j + 1] = v;
			}
			for (int k = s; k < e; ++k)
			{
				const int ci = W.adj[k];
				uint32_t cf = C.flags[ci];
				if (cf & CF_ISLAND) continue;
				W.si_contacts[cStart + nc] = ci;
				++nc;
				C.flags[ci] = cf | CF_ISLAND;
				int4 ids = C.ids[ci];
				const int other = ids.z == b ? ids.w : ids.z;
				if ((W.b_flags[other] & BF_TYPE_MASK) == BT_STATIC) continue;
				if (W.b_slot[other] != -1) continue;
				stack[sp++] = other;
				W.b_slot[other] = -2;
			}
			// joints of b, newest first (b2World.cpp:1292-1318)
			for (int k = W.jadjStart[b]; k < W.jadjStart[b + 1]; ++k)
			{
				const int ji = W.jadj[k];
				RevoluteJoint& jn = W.joints[ji];
				if (jn.islandFlag) continue;
				const int other = jn.bodyA == b ? jn.bodyB : jn.bodyA;
				if ((W.b_flags[other] & BF_ACTIVE) == 0) continue;
				W.lj_list[W.rootJointStart[root] + nj] = ji;
				++nj;
				jn.islandFlag = 1;
				if ((W.b_flags[other] & BF_TYPE_MASK) == BT_STATIC) continue;
				if (W.b_slot[other] != -1) continue;
				stack[sp++] = other;
				W.b_slot[other] = -2;
			}
		}
		// dependency levels of the sequential constraint sweep
		int maxLevel = 0;
		for (int j = 0; j < nc; ++j)
		{
			int4 ids = C.ids[W.si_contacts[cStart + j]];
			int sa = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC ? W.b_slot[ids.z] : -1;
			int sb = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC ? W.b_slot[ids.w] : -1;
			int la = sa >= 0 ? W.si_lastLevel[sa] : 0;
			int lb = sb >= 0 ? W.si_lastLevel[sb] : 0;
			int level = 1 + (la > lb ? la : lb);
			W.si_level[cStart + j] = level;
			if (sa >= 0) W.si_lastLevel[sa] = level;
			if (sb >= 0) W.si_lastLevel[sb] = level;
			if (level > maxLevel) maxLevel = level;
		}
		W.si_maxLevel[idx] = maxLevel;
	}
}

// Joint lists of the LARGE (coloured) islands: grouped per root, ascending joint id inside an island.
__global__ __launch_bounds__(256) void k_joints_fill(DW W)
{
	b2dPhaseStamp(W);
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x)
	{
		const RevoluteJoint& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		if (((W.b_flags[jn.bodyA] & W.b_flags[jn.bodyB]) & BF_ACTIVE) == 0) continue;
		int b = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC ? jn.bodyA : jn.bodyB;
		if ((W.b_flags[b] & BF_TYPE_MASK) == BT_STATIC) continue;
		const int root = W.parent[b];
		if (W.rootIsland[root] != ROOT_LARGE) continue;
		W.lj_list[W.rootJointStart[root] + atomicAdd(&W.rootJointCursor[root], 1)] = j;
		// the bodies of a large island's joint are touched by the workgroup that walks the joints behind the colours of a sweep:
		// in a fused launch (k_rest_hub) their last rest row hands them on to it (REST_SERIAL_BIT = bit HUB_COLOR of DW::bodyRest,
		// wiped by k_island_init)
		{
			int ids[4] = { jn.bodyA, jn.bodyB, -1, -1 };
			if (jn.type == B2D_JOINT_GEAR) { const GearRec& g = W.gears[jn.enableLimit]; ids[2] = g.bodyC; ids[3] = g.bodyD; }
			for (int q = 0; q < 4; ++q)
				if (ids[q] >= 0 && (W.b_flags[ids[q]] & BF_TYPE_MASK) != BT_STATIC) atomicOr((unsigned long long*)&W.bodyRest[ids[q]], 1ull << HUB_COLOR);
		}
	}
}

__global__ __launch_bounds__(64) void k_joints_sort(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLIslands;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int root = W.li_roots[k];
		const int s = W.rootJointStart[root], e = s + W.rootJoints[root];
		for (int a = s + 1; a < e; ++a)
		{
			int v = W.lj_list[a];
			int b = a - 1;
			while (b >= s && W.lj_list[b] > v)
			{
				W.lj_list[b + 1] = W.lj_list[b];
				--b;
			}
			W.lj_list[b + 1] = v;
		}
	}
}

#endif
