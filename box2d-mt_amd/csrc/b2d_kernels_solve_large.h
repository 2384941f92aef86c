// b2d_kernels_solve_large.h - the island solver for LARGE islands (one pyramid, the tumbler pile).
//
// A large island cannot be walked by one lane, so its constraints are coloured (no two constraints
// of a colour share a non-static body) and each colour is one data-parallel phase; phases are
// separate launches on the world's stream (a kernel boundary is the cheapest grid-wide barrier on
// MI355X). Within a Gauss-Seidel sweep this visits constraints in a different order than the
// reference's DFS order: the iterates differ at the level the reference itself differs between two
// valid orders, which is why poses of large islands are compared with a tolerance, while contact
// counts and island membership stay exact.
//
// Constraint rows are field-major in HBM (lc[field * cap + row]) and rows are ordered by colour,
// so a wave reads every field with fully coalesced 256-byte requests.
#ifndef B2D_KERNELS_SOLVE_LARGE_H
#define B2D_KERNELS_SOLVE_LARGE_H

#include "b2d_kernels_solve_small.h"

#define LC_WORDS ((int)(sizeof(ContactConstraint) / 4))
#define LC_VEL_WORDS 36   // words [0, 36) = velocity part incl. pointCount
#define LC_IMP_FIRST 10   // normalImpulse[2], tangentImpulse[2] = words 10..13
#define LC_MASS_FIRST 28  // invMassA, invMassB, invIA, invIB
#define LC_POS_FIRST 36

// The constraint is moved between registers and its field-major HBM row through a word image
// (memcpy keeps this free of aliasing games; it compiles to plain register moves).
__device__ __forceinline__ void lcStore(const DW& W, int row, const ContactConstraint& cc, int first, int last)
{
	uint32_t words[LC_WORDS];
	memcpy(words, &cc, sizeof(ContactConstraint));
	uint32_t* dst = (uint32_t*)W.lc;
#pragma unroll
	for (int f = 0; f < LC_WORDS; ++f)
	{
		if (f >= first && f < last) dst[(size_t)f * W.capContacts + row] = words[f];
	}
}

__device__ __forceinline__ void lcLoad(const DW& W, int row, ContactConstraint& cc, int first, int last)
{
	uint32_t words[LC_WORDS];
	memcpy(words, &cc, sizeof(ContactConstraint));
	const uint32_t* src = (const uint32_t*)W.lc;
#pragma unroll
	for (int f = 0; f < LC_WORDS; ++f)
	{
		if (f >= first && f < last) words[f] = src[(size_t)f * W.capContacts + row];
	}
	memcpy(&cc, words, sizeof(ContactConstraint));
}

// ---- bodies --------------------------------------------------------------------------------------
// `sortJoints`: ... and every large island's joint list put in ascending joint order (k_joints_sort's loop, one lane per
// island: a launch of its own in the chain before round 6).
__global__ __launch_bounds__(256) void k_large_integrate(DW W, StepParams sp, int sortJoints)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (sortJoints)
	{
		const int nI = S->c.nLIslands;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nI; k += gridDim.x * blockDim.x)
		{
			const int root = W.li_roots[k];
			const int s0 = W.rootJointStart[root], e = s0 + W.rootJoints[root];
			for (int a = s0 + 1; a < e; ++a)
			{
				const int v = W.lj_list[a];
				int b = a - 1;
				while (b >= s0 && W.lj_list[b] > v) { W.lj_list[b + 1] = W.lj_list[b]; --b; }
				W.lj_list[b + 1] = v;
			}
		}
	}
	const int n = S->c.nLBodies;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int body = W.li_bodies[k];
		float4 pos = W.b_pos[body];
		W.b_pos0[body] = make_float4(pos.x, pos.y, pos.z, 0.0f);
		uint32_t f = W.b_flags[body];
		if ((f & BF_TYPE_MASK) == BT_DYNAMIC)
		{
			float4 vel = W.b_vel[body];
			float4 m = W.b_mass[body], damp = W.b_damp[body], force = W.b_force[body];
			V2 v = v2(vel.x, vel.y);
			float w = vel.z;
			b2dIntegrateVelocity(&v, &w, sp.dt, sp.gravity, damp.z, m.x, m.y, v2(force.x, force.y), force.z, damp.x, damp.y);
			W.b_vel[body] = make_float4(v.x, v.y, w, 0.0f);
		}
	}
}

// ---- colouring (Jones-Plassmann by claims; deterministic: priorities are a bijection of the contact index)
__device__ __forceinline__ uint32_t colorPriority(int contactIndex)
{
	return ((uint32_t)contactIndex + 1u) * 2654435761u;
}

// Colour classes of the block partition: a constraint between bodies of two blocks (a CUT constraint) takes its colour
// from [CUT_COLOR_BASE, HUB_COLOR), every other one from [0, CUT_COLOR_BASE): a sweep in colour order then visits, on
// every body, the constraints inside its block first and the ones that cross a block boundary last.
#define COLOR_INTERIOR_BITS ((1ull << CUT_COLOR_BASE) - 1ull)
__device__ __forceinline__ bool constraintIsCut(const DW& W, int bodyA, bool nsA, int bodyB, bool nsB)
{
	return nsA && nsB && effBlk(W, bodyA) != effBlk(W, bodyB);
}
// Colours a constraint may not KEEP from one step to the next. On top of colorClassMask: an interior constraint in the upper
// range is a leftover - it was a cut constraint until its two bodies came to share a block (a new partition: the 50 086-box
// pyramid makes one every few steps) - and would go on being handed over through memory for no reason: after 84 partitions
// 105 000 of that pile's 135 000 constraints sat in the upper range although 15 000 were cut (k_solve_blocks 549 us).
// It gives the colour up and takes the lowest free one again (in the rare case that this is an upper colour once more - both
// bodies holding 32 others - it does so every step: correct, merely busy).
__device__ __forceinline__ uint64_t colorStaleMask(bool cut)
{
	const uint64_t lower = (1ull << CUT_COLOR_BASE) - 1ull, hub = 1ull << HUB_COLOR;
	return (cut ? lower : ~lower) | hub;
}
// Colours a constraint of this class may NOT take. A cut constraint must sit in the upper range (k_solve_blocks hands the
// bodies of upper-range constraints over through memory, and only those). An interior constraint takes the lowest free
// colour, which lies in the lower range unless one of its bodies already holds 32 colours; it may spill into the upper
// range then (it is handed over through memory like a cut constraint: correct, merely slower).
__device__ __forceinline__ uint64_t colorClassMask(bool cut)
{
	return (cut ? COLOR_INTERIOR_BITS : 0ull) | (1ull << HUB_COLOR);
}
// (tests/test_gpu_recovery.py: B2HIP_TEST_MAX_COLORS = N leaves the colours [0, N) of each range - a pile then runs out of
// colours on its busiest bodies, as a world with 64 different ones on two bodies would)
__device__ __forceinline__ uint64_t colorTestMask(const DW& W)
{
	if (W.testMaxColors <= 0 || W.testMaxColors >= CUT_COLOR_BASE) return 0ull;
	const uint64_t keep = (1ull << W.testMaxColors) - 1ull;
	return ~(keep | (keep << CUT_COLOR_BASE));
}
__device__ __forceinline__ void noteColorUsed(DState* S, int color)
{
	if (color < 32) atomicOr(&S->c.colorMaskLo, 1u << color);
	else atomicOr(&S->c.colorMaskHi, 1u << (color - 32));
}

// A constraint that is not coloured but swept in order by k_large_hub after the coloured ones of every sweep: a hub's
// (more constraints on one body than colours can separate), or - in islands that are swept block-wise between joint walks
// and hub sweeps - one whose body has no home block yet (a newcomer that landed on other newcomers: the handful of such
// constraints a step brings is cheaper swept in order than a new partition of the whole island).
__device__ __forceinline__ bool rowIsHubs(const DW& W, bool nsA, int bodyA, bool nsB, int bodyB)
{
	return (nsA && W.deg[bodyA] > HUB_DEGREE) || (nsB && W.deg[bodyB] > HUB_DEGREE);
}
__device__ __forceinline__ bool rowIsSerial(const DW& W, bool nsA, int bodyA, bool nsB, int bodyB)
{
	if (rowIsHubs(W, nsA, bodyA, nsB, bodyB)) return true;
	if (!W.serialOrphans) return false;
	return (nsA && effBlk(W, bodyA) == 0) || (nsB && effBlk(W, bodyB) == 0);
}

__global__ __launch_bounds__(256) void k_color_begin(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLContacts;
	for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x)
	{
		if (W.li_color[s] != HUB_COLOR) W.li_color[s] = -1; // hub constraints are never coloured (k_color_check marked them)
	}
	if (blockIdx.x == 0 && threadIdx.x <= MAX_COLORS)
	{
		if (threadIdx.x != HUB_COLOR) W.colorCount[colorSlot(threadIdx.x)] = 0;
		W.colorCursor[colorSlot(threadIdx.x)] = 0;
	}
	// a full recolour forgets every stored colour (also those of non-touching contacts)
	{
		const ContactArrays& C = W.ca[S->cur];
		const int nAll = S->c.nContacts;
		for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nAll; i += gridDim.x * blockDim.x) C.color[i] = -1;
	}
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		W.bodyColorMask[i] = 0;
		W.bodyClaim[i] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		S->c.nUncolored = n - W.colorCount[colorSlot(HUB_COLOR)];
		S->c.colorRounds = 0;
		S->c.nColors = 0;
		S->c.colorMaskLo = S->c.colorMaskHi = 0u;
	}
}

// Colours persist with the contact (ContactArrays::color). A step can reuse them if every constraint of
// the large islands already has one and no two constraints on a body share a colour; otherwise the host
// runs the colouring rounds again. Also builds the per-colour census for the reuse case.
__global__ __launch_bounds__(256) void k_color_check(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	// (round 6) a world WITHOUT a partition - the settled Tumbler, whose island is beyond what the block solvers take - has no
	// home blocks: effBlk is 0 for every body (b_blk1 and b_adopt were wiped when the partition was dissolved), no constraint
	// is cut, every one is an "orphan". Said once here instead of found out by three or four gathers per body and constraint.
	const bool noPart = S->c.nBlocks == 0;
	auto isCut = [&](int bodyA, bool nsA, int bodyB, bool nsB) -> bool { return noPart ? false : constraintIsCut(W, bodyA, nsA, bodyB, nsB); };
	int bad = 0;
	int uncolored = 0;
	int maxColor = 0;
	int orphanRows = 0, cutRows = 0;
	uint32_t usedLo = 0u, usedHi = 0u;
	__shared__ int s_blkRows[MAX_BLOCKS];
	__shared__ int s_sums[2];
	__shared__ int s_compact[2]; // [0] rows of the compaction class in this round, [1] their first place in the list
	__shared__ int s_uncol[2];   // the same for the rows without a colour
	for (int i = threadIdx.x; i < MAX_BLOCKS; i += blockDim.x) s_blkRows[i] = 0;
	if (threadIdx.x < 2) { s_sums[threadIdx.x] = 0; s_compact[threadIdx.x] = 0; s_uncol[threadIdx.x] = 0; }
	__syncthreads();
	// (1) every touching contact that owns a colour - large island or not, asleep or not - reserves it on its bodies.
	// A contact that stopped touching gives its colour back: reservations of idle neighbours (a pyramid box has two of
	// them) would push new constraints to ever higher colours, and the depth of a sweep is the number of colours.
	const int nAll = S->c.nContacts;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nAll; i += gridDim.x * blockDim.x)
	{
		// (colour and flags together - one round trip, not two in a row, for the one contact in seven of a dense pile that owns
		// a colour; and without a partition nothing is cut: the answer needs neither the ids nor the bodies)
		const int col = C.color[i];
		const uint32_t cfl = C.flags[i];
		if (col < 0 || col >= MAX_COLORS) continue;
		if ((cfl & (CF_TOUCHING | CF_SENSOR)) != CF_TOUCHING)
		{
			C.color[i] = -1;
			continue;
		}
		const unsigned long long bit = 1ull << col;
		if (noPart)
		{
			if (bit & colorStaleMask(false)) C.color[i] = -1;
			continue;
		}
		int4 ids = C.ids[i];
		const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC, nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
		if (bit & colorStaleMask(isCut(ids.z, nsA, ids.w, nsB)))
		{
			// the constraint changed class (a body moved to another block, a new partition): its colour is void
			C.color[i] = -1;
			continue;
		}
		// (the reservation on the bodies - DW::bodyColorMask - is made body by body, by k_color_masks behind this kernel: two
		// returning 64-bit atomics per touching contact here were 800 000 on the 50 086-box pyramid and 740 000 on the Tumbler,
		// served at the memory side's ~8 per ns - most of this kernel's 115 - 145 us)
	}
	// (2) census of the large-island constraints by colour
	const int n = S->c.nLContacts;
	for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x)
	{
		const int s = base + threadIdx.x;
		const bool valid = s < n;
		int col = -1;
		int compactRank = -1, uncolRank = -1;
		if (valid)
		{
			const int ci = W.li_contacts[s];
			// (what hangs on the contact in one round trip, what hangs on its bodies in the next: the degrees are fetched beside
			// the body flags, not behind them)
			const int4 ids = C.ids[ci];
			const int colStored = C.color[ci];
			const uint32_t bfA = W.b_flags[ids.z], bfB = W.b_flags[ids.w];
			const int dgA = W.deg[ids.z], dgB = W.deg[ids.w];
			const bool nsA = (bfA & BF_TYPE_MASK) != BT_STATIC, nsB = (bfB & BF_TYPE_MASK) != BT_STATIC;
			const bool isHubs = (nsA && dgA > HUB_DEGREE) || (nsB && dgB > HUB_DEGREE); // (rowIsHubs)
			const bool hubA = isHubs || (W.serialOrphans && rowIsSerial(W, nsA, ids.z, nsB, ids.w)), hubB = false; // (swept in order: see rowIsSerial)
			if (hubA && !isHubs) atomicAdd(&S->c.nSerialOrphans, 1);
			{
				// block census: the row belongs to the home block of its first non-static body (counted in LDS: ten
				// thousand rows adding to a few dozen words of memory serialise in L2)
				const int blkA = (nsA && !noPart) ? effBlk(W, ids.z) : 0, blkB = (nsB && !noPart) ? effBlk(W, ids.w) : 0;
				// (a hub's constraints belong to no block: k_large_hub sweeps them, in a segment of their own behind the blocks' rows)
				const int owner = nsA ? blkA : blkB;
				if (!hubA && !hubB)
				{
					if (owner > 0 && owner <= MAX_BLOCKS) atomicAdd(&s_blkRows[owner - 1], 1);
					if ((nsA && blkA == 0) || (nsB && blkB == 0)) ++orphanRows;
					if (nsA && nsB && blkA != blkB) ++cutRows;
				}
			}
			col = colStored;
			if (col >= 0 && col < MAX_COLORS && ((1ull << col) & colorStaleMask(isCut(ids.z, nsA, ids.w, nsB)))) col = -1; // (voided above)
			if (hubA || hubB)
			{
				// a hub constraint owns no colour (and reserves none from the next step on)
				col = HUB_COLOR;
				C.color[ci] = -1;
			}
			else if (col == HUB_COLOR) col = -1;
			if (col < 0 || col >= MAX_COLORS)
			{
				++uncolored;
				col = -1;
				// short list for the in-kernel incremental colouring (k_color_small): places taken below, one atomic per workgroup
				// (one per row was a few thousand returning atomics on one word in every step of the settled Tumbler - the contacts that
				// began to touch - each served in its turn at the memory side: 30 us of its step)
				uncolRank = atomicAdd(&s_uncol[0], 1);
			}
			else if (col != HUB_COLOR)
			{
				if (col + 1 > maxColor) maxColor = col + 1;
				if (col < 32) usedLo |= 1u << col; else usedHi |= 1u << (col - 32);
				// candidates of this step's compaction class (k_color_small moves them down if a lower colour is free): listed below
				compactRank = (col == S->c.compactClass && col > 0) ? atomicAdd(&s_compact[0], 1) : -1;
			}
			W.li_color[s] = col;
		}
		(void)blockKeyedAlloc65(W.colorCount, col < 0 ? 0 : col, valid && col >= 0, false, true);
		// The class under compaction takes its places in the list with ONE atomic per workgroup and round. (One per row - the
		// value is needed - was a queue of 7 000 returning atomics on one word for Pyramid 316, 18 000 for the Tumbler: each
		// workgroup's round waited 60 - 85 us in it. The list holds COLOR_SMALL_MAX rows; once the count is past that the
		// class is not compacted this step - k_color_small - and only "more than that" matters: the adds stop.)
		__syncthreads();
		if (threadIdx.x == 0)
		{
			const int cnt = s_compact[0];
			s_compact[1] = COLOR_SMALL_MAX;
			if (cnt > 0 && __hip_atomic_load(&S->c.nCompact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= COLOR_SMALL_MAX) s_compact[1] = atomicAdd(&S->c.nCompact, cnt);
			s_compact[0] = 0;
			const int cu = s_uncol[0];
			s_uncol[1] = cu > 0 ? atomicAdd(&S->c.nUncolList, cu) : 0;
			s_uncol[0] = 0;
		}
		__syncthreads();
		if (uncolRank >= 0)
		{
			const int u = s_uncol[1] + uncolRank;
			if (u < COLOR_SMALL_MAX) W.uncolList[u] = s;
		}
		if (compactRank >= 0)
		{
			const int u = s_compact[1] + compactRank;
			if (u < COLOR_SMALL_MAX) W.compactList[u] = s;
		}
	}
	// needRecolor bit0: two constraints on one body share a colour -> colour everything again;
	// nUncolored: constraints without a colour yet -> incremental rounds on top of the existing masks
	if (bad) atomicOr(&S->c.needRecolor, 1);
	// (nUncolored = nUncolList - the same rows, counted once, by the workgroups' list allocations above; k_color_masks, the
	// kernel behind this one, copies it. A sum per wave on top was 1 500 more atomics on the line of Counters that every
	// workgroup of this kernel already queues at: a clock around the sections showed 13 - 45 us per workgroup spent there)
	(void)uncolored;
	// wave-combined, and only what would CHANGE the word (the maximum and the masks only grow while this kernel runs; a load
	// past the L2 first): one atomic per wave on each of the three words was 16 000 atomics, served one after the other, for
	// the 350 000 constraints of the settled Tumbler - most of this kernel's 209 us
	for (int off = 32; off > 0; off >>= 1)
	{
		usedLo |= (uint32_t)__shfl_xor((int)usedLo, off);
		usedHi |= (uint32_t)__shfl_xor((int)usedHi, off);
		const int om = __shfl_xor(maxColor, off);
		maxColor = om > maxColor ? om : maxColor;
	}
	orphanRows = waveSumInt(orphanRows);
	cutRows = waveSumInt(cutRows);
	if (waveLane() == 0)
	{
		if (maxColor) atomicMaxIfAbove(&S->c.nColors, maxColor);
		if (usedLo & ~__hip_atomic_load(&S->c.colorMaskLo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&S->c.colorMaskLo, usedLo);
		if (usedHi & ~__hip_atomic_load(&S->c.colorMaskHi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&S->c.colorMaskHi, usedHi);
		if (orphanRows) atomicAdd(&s_sums[0], orphanRows);
		if (cutRows) atomicAdd(&s_sums[1], cutRows);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < MAX_BLOCKS; i += blockDim.x)
	{
		const int v = s_blkRows[i];
		if (v) atomicAdd(&W.blkRows[(size_t)i * BLK_SLOT], v);
	}
	if (threadIdx.x == 0)
	{
		// (carried by the arrival of the workgroups: b2d_world.h)
		if ((unsigned)W.capContacts <= TREE_SUM_MAX)
		{
			unsigned t0 = 0u, t1 = 0u;
			if (b2dTreeArrive(W.arriveTree + (size_t)ARRIVE_COLOR_CHECK * TREE_WORDS, (unsigned)s_sums[0], (unsigned)s_sums[1], &t0, &t1))
			{
				if (t0) atomicAdd(&S->c.nOrphanRows, (int)t0);
				if (t1) atomicAdd(&S->c.nCutRows, (int)t1);
			}
		}
		else
		{
			if (s_sums[0]) atomicAdd(&S->c.nOrphanRows, s_sums[0]);
			if (s_sums[1]) atomicAdd(&S->c.nCutRows, s_sums[1]);
		}
	}
	// the home bodies of every block, counted for k_block_census; an adoption (b_adopt: a neighbour's block, offered by
	// k_island_edges / k_block_adopt) becomes the body's own block (small worlds: k_block_census does it itself)
	if (S->c.nLBodies > CENSUS_WG_MAX_BODIES)
	{
		const int nLB = S->c.nLBodies;
		const int nb = S->c.nBlocks < MAX_BLOCKS ? S->c.nBlocks : MAX_BLOCKS;
		for (int base = blockIdx.x * blockDim.x; base < nLB; base += gridDim.x * blockDim.x)
		{
			const int k = base + threadIdx.x;
			int e = 0;
			if (k < nLB)
			{
				const int body = W.li_bodies[k];
				e = effBlk(W, body);
				if (e > 0 && e <= nb) W.b_blk1[body] = e; else e = 0;
			}
			if (e > 0) atomicAdd(&W.blkBodyCount[(size_t)(e - 1) * BLK_SLOT], 1);
		}
	}
}

// The colours reserved on every body: the OR of the colours its solid contacts own (k_color_check has just voided what may not
// be kept), read from the body's adjacency segment and stored with a plain store. Two contacts of one body with the same
// colour -> Counters::needRecolor, as the returning atomics of round 4's k_color_check found them. (A touching contact that
// a PreSolve switched off for this step is not solid: it keeps its colour but reserves nothing this step; should a new
// contact take that colour meanwhile, the clash is found here next step and the island is coloured afresh.)
__global__ __launch_bounds__(256) void k_color_masks(DW W)
{
	b2dPhaseStamp(W);
	if (blockIdx.x == 0 && threadIdx.x == 0) W.st->c.nUncolored = W.st->c.nUncolList; // (k_color_check's count of the rows without a colour)
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	int bad = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if ((W.b_flags[i] & BF_TYPE_MASK) == BT_STATIC) continue;
		const int d = W.deg[i];
		unsigned long long mask = 0ull;
		// (a hub's constraints own no colour: k_color_check keeps them colourless)
		if (d > 0 && d <= HUB_DEGREE)
		{
			const int s = W.adjStart[i];
			for (int k = 0; k < d; ++k)
			{
				const int col = C.color[W.adj[s + k]];
				if (col < 0 || col >= MAX_COLORS || col == HUB_COLOR) continue;
				const unsigned long long bit = 1ull << col;
				if (mask & bit) bad = 1;
				mask |= bit;
			}
		}
		W.bodyColorMask[i] = mask;
	}
	if (bad) atomicOr(&S->c.needRecolor, 1);
}

__global__ __launch_bounds__(256) void k_color_claim(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nUncolored == 0) return;
	const int n = S->c.nLContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x)
	{
		if (W.li_color[s] >= 0) continue;
		const int ci = W.li_contacts[s];
		int4 ids = C.ids[ci];
		uint32_t pr = colorPriority(ci);
		if ((W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC) atomicMax(&W.bodyClaim[ids.z], pr);
		if ((W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC) atomicMax(&W.bodyClaim[ids.w], pr);
	}
}

__global__ __launch_bounds__(256) void k_color_resolve(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nUncolored == 0) return;
	const int n = S->c.nLContacts;
	const ContactArrays& C = W.ca[S->cur];
	int colored = 0;
	for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x)
	{
		if (W.li_color[s] >= 0) continue;
		const int ci = W.li_contacts[s];
		int4 ids = C.ids[ci];
		uint32_t pr = colorPriority(ci);
		const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
		const bool nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
		bool win = true;
		if (nsA && __hip_atomic_load(&W.bodyClaim[ids.z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != pr) win = false;
		if (nsB && __hip_atomic_load(&W.bodyClaim[ids.w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != pr) win = false;
		if (!win) continue;
		// this lane is the only winner on both bodies this round: plain read-modify-write is safe
		uint64_t used = colorClassMask(constraintIsCut(W, ids.z, nsA, ids.w, nsB)) | colorTestMask(W);
		if (nsA) used |= W.bodyColorMask[ids.z];
		if (nsB) used |= W.bodyColorMask[ids.w];
		int color = used == ~0ull ? MAX_COLORS - 1 : __ffsll((long long)~used) - 1;
		if (used == ~0ull) atomicOr(&S->c.overflow, 4);
		uint64_t bit = 1ull << color;
		if (nsA) { W.bodyColorMask[ids.z] |= bit; W.bodyClaim[ids.z] = 0; }
		if (nsB) { W.bodyColorMask[ids.w] |= bit; W.bodyClaim[ids.w] = 0; }
		W.li_color[s] = color;
		C.color[ci] = color;
		atomicAdd(&W.colorCount[colorSlot(color)], 1);
		atomicMax(&S->c.nColors, color + 1);
		noteColorUsed(S, color);
		++colored;
	}
	if (colored) atomicSub(&S->c.nUncolored, colored);
}

// Incremental colouring without a host round trip: the few constraints that have no colour yet (new contacts of a
// settled island) are listed by k_color_check; one workgroup runs the Jones-Plassmann rounds over that list only.
// Same claims, same priorities, same "lowest free colour" rule as k_color_claim / k_color_resolve, so the colours are
// the ones the grid-wide rounds would hand out.
// `queuedAhead`: launched behind the census before the host has read it (it runs while the host sizes the solver launches):
// only if the partition is certain to stay (b2dPartitionSettled), else the host launches it again in its usual place.
// (round 6) ... or if no partition CAN be made this step: the large islands hold more constraints than the block solvers take
// (`aheadMinRows` > 0: the host's threshold for staying in that state, b2hip_host_phases.h: blocksTooBig) and no partition
// exists. Such a world runs launch per colour, where the host needs the final colour count: the kernel publishes the state
// behind its work (`pub`, like the census) and the host polls instead of copying and synchronising - on the settled 100 000-box
// Tumbler the early return + the host's second launch + the copy were ~60 us of device idle per step.
__host__ __device__ inline bool b2dColorAheadNoPartition(const Counters& c, int aheadMinRows)
{
	return aheadMinRows > 0 && c.nBlocks == 0 && c.nLContacts >= aheadMinRows;
}
__device__ __forceinline__ void colorSmallBody(const DW& W);
__global__ __launch_bounds__(1024) void k_color_small(DW W, int queuedAhead, int aheadMinRows, DState* pub, int pubSeq)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (threadIdx.x == 0) S->gapClock[2] = wall_clock64();
	bool run = true, publish = pub != nullptr;
	if (queuedAhead)
	{
		const bool noPart = b2dColorAheadNoPartition(S->c, aheadMinRows);
		run = b2dPartitionSettled(S->c) || noPart;
		publish = publish && noPart; // (the host evaluates the same function on the same counters: it knows what to wait for)
	}
	if (S->c.nUncolList > COLOR_SMALL_MAX || S->c.needRecolor) run = false;
	if (run) colorSmallBody(W);
	if (publish)
	{
		__syncthreads();
		// the census by colour as it stands now (the host sizes the colour launches and picks the rest colours from it)
		if (run && threadIdx.x < MAX_COLORS) S->c.colorRows[threadIdx.x] = threadIdx.x == HUB_COLOR ? 0 : __hip_atomic_load(&W.colorCount[colorSlot(threadIdx.x)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (workgroup scope like the adds above: this XCD's L2 holds them)
		__syncthreads();
		b2dPublishCensus(W, pub, pubSeq); // (a buffer and a count of its own: b2d_world.h)
	}
}
// (round 6: ONE workgroup runs this - its atomics and its looks at the masks and claims carry workgroup scope: they are served
// by this XCD's L2 instead of going past it, ~0.3 us a round trip instead of 1 - 2, and a round is eight of them in a row)
__device__ __forceinline__ void colorSmallBody(const DW& W)
{
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	__shared__ int s_left, s_colored, s_maxColor, s_n;
	if (threadIdx.x == 0)
	{
		s_n = S->c.nUncolList;
		s_maxColor = 0;
	}
	__syncthreads();
	// Colour compaction: a constraint of this step's class gives its colour up when a lower one is free on both bodies,
	// and takes the lowest free colour in the rounds below. Constraints of one class never share a body, so each
	// decision reads masks nobody else is changing: deterministic. Over the steps the colouring becomes greedy-minimal
	// (colour <= (degA - 1) + (degB - 1)), whatever the order in which the contacts appeared.
	{
		// all of the class or none of it: a class that does not fit the list (together with the new constraints) would be
		// cut at an arbitrary member - which ones were listed depends on the order of the atomics in k_color_check
		const int nc = S->c.nCompact + S->c.nUncolList <= COLOR_SMALL_MAX ? S->c.nCompact : 0;
		for (int k = threadIdx.x; k < nc; k += blockDim.x)
		{
			const int s = W.compactList[k];
			const int c = W.li_color[s];
			if (c <= 0 || c >= MAX_COLORS) continue;
			const int ci = W.li_contacts[s];
			const int4 ids = C.ids[ci];
			const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
			const bool nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
			uint64_t used = colorClassMask(constraintIsCut(W, ids.z, nsA, ids.w, nsB)) | colorTestMask(W);
			if (nsA) used |= __hip_atomic_load((unsigned long long*)&W.bodyColorMask[ids.z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (nsB) used |= __hip_atomic_load((unsigned long long*)&W.bodyColorMask[ids.w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (used == ~0ull || __ffsll((long long)~used) - 1 >= c) continue;
			const int slot = atomicAdd(&s_n, 1);
			if (slot >= COLOR_SMALL_MAX) continue;
			const unsigned long long bit = 1ull << c;
			if (nsA) (void)__hip_atomic_fetch_and((unsigned long long*)&W.bodyColorMask[ids.z], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (nsB) (void)__hip_atomic_fetch_and((unsigned long long*)&W.bodyColorMask[ids.w], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			W.li_color[s] = -1;
			C.color[ci] = -1;
			(void)__hip_atomic_fetch_add(&W.colorCount[colorSlot(c)], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			W.uncolList[slot] = s;
		}
	}
	__syncthreads();
	const int n = s_n < COLOR_SMALL_MAX ? s_n : COLOR_SMALL_MAX;
	if (threadIdx.x == 0)
	{
		s_left = n;
		// (the next step's colorCheckBegin slows the visits down once a whole round of classes has moved nothing)
		if (s_n > S->c.nUncolList) S->c.compactMoved = s_n - S->c.nUncolList;
	}
	// what a round needs of a constraint, read once (a round used to start with five dependent loads per constraint and pass:
	// 200 us of the Tumbler's step, where a thousand contacts are new every step)
	constexpr int PER = COLOR_SMALL_MAX / 1024;
	int it_s[PER], it_ci[PER], it_a[PER], it_b[PER];
	uint32_t it_pr[PER];
	uint64_t it_class[PER];
	bool it_open[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j)
	{
		const int k = (int)threadIdx.x + j * 1024;
		it_open[j] = false;
		it_s[j] = it_ci[j] = 0; it_a[j] = it_b[j] = -1; it_pr[j] = 0u; it_class[j] = 0ull;
		if (k < n)
		{
			const int s = W.uncolList[k];
			it_s[j] = s;
			it_open[j] = W.li_color[s] < 0;
			const int ci = W.li_contacts[s];
			it_ci[j] = ci;
			const int4 ids = C.ids[ci];
			const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
			const bool nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
			it_a[j] = nsA ? ids.z : -1;
			it_b[j] = nsB ? ids.w : -1;
			it_pr[j] = colorPriority(ci);
			it_class[j] = colorClassMask(constraintIsCut(W, ids.z, nsA, ids.w, nsB)) | colorTestMask(W);
		}
	}
	__syncthreads();
	const int maxRounds = W.testColorRounds > 0 ? W.testColorRounds : 4 * MAX_COLORS; // (B2HIP_TEST_COLOR_ROUNDS: tests of the fall-back to the grid-wide rounds)
	for (int round = 0; round < maxRounds && s_left > 0; ++round)
	{
		if (threadIdx.x == 0) s_colored = 0;
#pragma unroll
		for (int j = 0; j < PER; ++j)
		{
			if (!it_open[j]) continue;
			if (it_a[j] >= 0) (void)__hip_atomic_fetch_max(&W.bodyClaim[it_a[j]], it_pr[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (it_b[j] >= 0) (void)__hip_atomic_fetch_max(&W.bodyClaim[it_b[j]], it_pr[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		__syncthreads();
		int colored = 0;
#pragma unroll
		for (int j = 0; j < PER; ++j)
		{
			if (!it_open[j]) continue;
			const int A = it_a[j], B = it_b[j];
			const uint32_t pr = it_pr[j];
			bool win = true;
			if (A >= 0 && __hip_atomic_load(&W.bodyClaim[A], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != pr) win = false;
			if (B >= 0 && __hip_atomic_load(&W.bodyClaim[B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != pr) win = false;
			if (!win) continue;
			uint64_t used = it_class[j];
			if (A >= 0) used |= __hip_atomic_load((unsigned long long*)&W.bodyColorMask[A], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (B >= 0) used |= __hip_atomic_load((unsigned long long*)&W.bodyColorMask[B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const int color = used == ~0ull ? MAX_COLORS - 1 : __ffsll((long long)~used) - 1;
			if (used == ~0ull) atomicOr(&S->c.overflow, 4);
			const unsigned long long bit = 1ull << color;
			if (A >= 0) { (void)__hip_atomic_fetch_or((unsigned long long*)&W.bodyColorMask[A], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_store(&W.bodyClaim[A], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
			if (B >= 0) { (void)__hip_atomic_fetch_or((unsigned long long*)&W.bodyColorMask[B], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_store(&W.bodyClaim[B], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
			W.li_color[it_s[j]] = color;
			C.color[it_ci[j]] = color;
			(void)__hip_atomic_fetch_add(&W.colorCount[colorSlot(color)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			atomicMax(&s_maxColor, color + 1);
			noteColorUsed(S, color);
			it_open[j] = false;
			++colored;
		}
		if (colored) atomicAdd(&s_colored, colored);
		__syncthreads();
		if (threadIdx.x == 0) s_left -= s_colored;
		__syncthreads();
	}
	if (threadIdx.x == 0)
	{
		int nc = S->c.nColors > s_maxColor ? S->c.nColors : s_maxColor;
		if (nc > MAX_COLORS) nc = MAX_COLORS;
		while (nc > 0 && __hip_atomic_load(&W.colorCount[colorSlot(nc - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) --nc;
		S->c.nColors = nc;
		S->c.nUncolored = s_left;
		S->c.colorRounds += 1;
	}
}

__global__ void k_color_scan(DW W)
{
	b2dPhaseStamp(W);
	// (the usual case - at most MAX_COLORS groups: one wave, a shuffle scan; the serial walk below was 64 dependent loads, ~14 us
	// in the chain of launches before the solver)
	if (W.st->c.nColors <= MAX_COLORS)
	{
		if (blockIdx.x == 0 && threadIdx.x < 64)
		{
			const int c = (int)threadIdx.x;
			const int cnt = c < MAX_COLORS ? W.colorCount[colorSlot(c)] : 0;
			int incl = cnt;
			for (int off = 1; off < 64; off <<= 1)
			{
				const int o = __shfl_up(incl, off);
				if (c >= off) incl += o;
			}
			W.colorStart[c] = incl - cnt;
			if (c == 63) W.colorStart[64] = incl;
			if (c == HUB_COLOR) W.st->c.nHubRows = cnt;
		}
		return;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		// all MAX_COLORS groups: the regular colours [0, nColors) first, the hub group (HUB_COLOR) last
		// (exact-order mode uses one group per dependency level, possibly thousands)
		const int nc = W.st->c.nColors > MAX_COLORS ? W.st->c.nColors : MAX_COLORS;
		int run = 0;
		for (int c = 0; c <= nc; ++c)
		{
			W.colorStart[c] = run;
			if (c < nc) run += W.colorCount[colorSlot(c)];
		}
		W.st->c.nHubRows = W.colorCount[colorSlot(HUB_COLOR)];
	}
}

// ---- exact-order mode (validation / bit-exact large islands) ------------------------------------
// Every island was walked by the DFS lane (k_island_dfs), so each constraint has its dependency
// level in the reference's sequential order. Using the LEVEL as the colour makes the phased solver
// below visit constraints in an order that commutes exactly with the reference's sweep, for any
// island size (at the price of as many phases as the deepest dependency chain).
__global__ __launch_bounds__(256) void k_exact_begin(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nSContacts;
	for (int c = blockIdx.x * blockDim.x + threadIdx.x; c <= n; c += gridDim.x * blockDim.x)
	{
		W.colorCount[colorSlot(c)] = 0;
		W.colorCursor[colorSlot(c)] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) S->c.nColors = 0;
}

__global__ __launch_bounds__(256) void k_exact_convert(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int nC = S->c.nSContacts, nB = S->c.nSBodies, nI = S->c.nSIslands;
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nC; j += gridDim.x * blockDim.x)
	{
		W.li_contacts[j] = W.si_contacts[j];
		const int color = W.si_level[j] - 1;
		W.li_color[j] = color;
		atomicAdd(&W.colorCount[colorSlot(color)], 1);
		atomicMax(&S->c.nColors, color + 1);
	}
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nB; k += gridDim.x * blockDim.x)
	{
		const int body = W.si_bodies[k];
		W.li_bodies[k] = body;
		W.b_flags[body] |= BF_LARGE;
	}
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nI; i += gridDim.x * blockDim.x)
	{
		W.li_roots[i] = W.si_root[i];
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		S->c.nLContacts = nC;
		S->c.nLBodies = nB;
		S->c.nLIslands = nI;
	}
}

// `restFirst`: the colours from this one up are swept by k_large_rest this step - their bits go into DW::bodyRest (64: none)
__global__ __launch_bounds__(256) void k_color_fill(DW W, int restFirst)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x)
	{
		const int s = base + threadIdx.x;
		const bool valid = s < n && W.li_color[s] >= 0;
		const int color = valid ? W.li_color[s] : 0;
		int ci = 0;
		int4 ids = make_int4(0, 0, 0, 0);
		bool nsA = false, nsB = false;
		if (valid)
		{
			ci = W.li_contacts[s];
			ids = C.ids[ci];
			nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
			nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
		}
		int p = 0;
		if (W.blockSort)
		{
			// k_solve_blocks: rows grouped by the home block of their first non-static body (the workgroup sorts its rows by
			// colour itself, in LDS); blkRowStart comes from the census of k_block_census
			int owner = valid ? effBlk(W, nsA ? ids.z : ids.w) - 1 : 0;
			if (valid && color == HUB_COLOR) owner = S->c.nBlocks < MAX_BLOCKS ? S->c.nBlocks : MAX_BLOCKS; // the segment behind the last block
			const bool placed = valid && owner >= 0 && owner <= MAX_BLOCKS;
			if (!placed) owner = 0;
			const int slot = waveKeyedAllocOnce(W.blkCursor, owner, placed, 11, BLK_SLOT); // (owner <= MAX_BLOCKS = 1024)
			p = W.blkRowStart[owner] + slot;
			if (placed) W.rowColor[p] = color;
			else if (valid) p = -1;
		}
		else
		{
			// (exact-order mode has one group per dependency level, possibly thousands: keys beyond the LDS histogram)
			const int slot = S->c.nColors > MAX_COLORS ? waveKeyedAlloc(W.colorCursor, colorSlot(color), valid) : blockKeyedAlloc65(W.colorCursor, color, valid, true, true);
			p = W.colorStart[color] + slot;
		}
		if (valid && p >= 0)
		{
			W.li_sorted[p] = s;
			// (every contact's row and every row's colour: the body-centred warm start gathers a body's rows in colour order)
			W.hubRowOf[ci] = p;
			if (!W.blockSort) W.rowColor[p] = color;
			if (color == HUB_COLOR)
			{
				// the lowest contact index among the primary hub's constraints with this partner (k_hub_flag: a partner's later
				// constraints with the hub - a box in a corner touches two walls - cannot join the one fixed point of k_sweep_end)
				if (W.hubWide)
				{
					const int hubP = (int)(uint32_t)(W.hubMeta[0] & 0xffffffffull);
					const bool pA = nsA && ids.z == hubP, pB = nsB && ids.w == hubP;
					if (W.hubMeta[0] != 0ull && pA != pB && (pA ? nsB : nsA))
						atomicMax(&W.hubFirst[pA ? ids.w : ids.z], ((unsigned long long)(uint32_t)S->c.hubEpoch << 32) | (unsigned long long)(0xffffffffu - (uint32_t)ci));
				}
			}
			W.li_ref[p] = make_int4(ci, nsA ? ids.z : -(ids.z + 1), nsB ? ids.w : -(ids.w + 1), W.parent[nsA ? ids.z : ids.w]);
			// a hub row's bodies are touched by the workgroup that sweeps the hub rows behind the colours: in a fused launch
			// (k_rest_hub) their last rest row hands them on to it (REST_SERIAL_BIT = this bit)
			if (color == HUB_COLOR && restFirst < MAX_COLORS)
			{
				if (nsA) atomicOr((unsigned long long*)&W.bodyRest[ids.z], 1ull << HUB_COLOR);
				if (nsB) atomicOr((unsigned long long*)&W.bodyRest[ids.w], 1ull << HUB_COLOR);
			}
			if (color >= restFirst && color != HUB_COLOR && S->c.nColors <= MAX_COLORS)
			{
				if (nsA) atomicOr((unsigned long long*)&W.bodyRest[ids.z], 1ull << color);
				if (nsB) atomicOr((unsigned long long*)&W.bodyRest[ids.w], 1ull << color);
			}
			// the upper-range colours of every body (its hand-overs through memory in k_solve_blocks, in this order)
			if (color >= CUT_COLOR_BASE && color != HUB_COLOR && S->c.nColors <= MAX_COLORS)
			{
				if (nsA) atomicOr((unsigned long long*)&W.bodyActive[ids.z], 1ull << color);
				if (nsB) atomicOr((unsigned long long*)&W.bodyActive[ids.w], 1ull << color);
			}
		}
	}
	// the home bodies of the blocks take their slots (blkBodyStart comes from k_block_census; which body gets which slot of
	// its block does not matter to the arithmetic: a slot is an LDS address)
	if (W.blockSort && S->c.nLBodies > CENSUS_WG_MAX_BODIES)
	{
		const int nLB = S->c.nLBodies;
		const int nb = S->c.nBlocks < MAX_BLOCKS ? S->c.nBlocks : MAX_BLOCKS;
		for (int base = blockIdx.x * blockDim.x; base < nLB; base += gridDim.x * blockDim.x)
		{
			const int k = base + threadIdx.x;
			int body = 0, e = 0;
			if (k < nLB)
			{
				body = W.li_bodies[k];
				e = W.b_blk1[body];
				if (e <= 0 || e > nb) e = 0;
			}
			if (e > 0)
			{
				// (one atomic per lane, all in flight together: the bodies of a wave belong to many blocks, and handing the slots out
				// block after block - waveKeyedAlloc - is a round trip per block)
				// (a line per block - BLK_SLOT - like the row counters: 50 000 bodies of the 50 086-box pyramid adding to 130 words on
				// FOUR lines queued behind one another there, 5 ns each: 64 us of this kernel's 76, and as much of k_color_check's
				// count, found with a clock around the loop)
				const int slot = atomicAdd(&W.blkBodyCursor[(size_t)(e - 1) * BLK_SLOT], 1);
				W.blkBodies[W.blkBodyStart[e - 1] + slot] = body;
				W.b_slot[body] = slot;
			}
		}
	}
}

// ---- constraints -----------------------------------------------------------------------------------
struct LargeRef
{
	int ci, bodyA, bodyB, root;
	bool nsA, nsB;
};

__device__ __forceinline__ LargeRef largeRef(const DW& W, const ContactArrays& C, int row)
{
	(void)C;
	const int4 q = W.li_ref[row];
	LargeRef r;
	r.ci = q.x;
	r.nsA = q.y >= 0;
	r.nsB = q.z >= 0;
	r.bodyA = r.nsA ? q.y : -(q.y + 1);
	r.bodyB = r.nsB ? q.z : -(q.z + 1);
	r.root = q.w;
	return r;
}

// ---- hub bodies ---------------------------------------------------------------------------------------
// The constraints of a body with a very large contact degree are inherently serial in a Gauss-Seidel sweep (each one
// reads the velocity the previous one wrote). They are visited in contact-index order by ONE wave after the coloured
// constraints of every sweep: the lanes fetch 64 rows and 64 partner bodies in parallel, then the hub-dependent solves
// run one lane at a time with the hub's state handed on through wave shuffles.
__global__ __launch_bounds__(256) void k_hub_flag(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		int f = 0;
		if ((C.flags[i] & (CF_ENABLED | CF_TOUCHING | CF_SENSOR | CF_DESTROY)) == (CF_ENABLED | CF_TOUCHING))
		{
			const int4 ids = C.ids[i];
			const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC, nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
			const int b = nsA ? ids.z : ids.w;
			if ((nsA || nsB) && W.rootIsland[W.parent[b]] == ROOT_LARGE)
			{
				// a hub's constraints first (in contact order), then the others that are swept in order: two counts in one word
				// (the scan that follows ranks both)
				if (W.hubWide)
				{
					// first the constraints of the PRIMARY hub that one fixed point can take (k_sweep_end): the partner is no hub
					// itself and this is its first constraint with the hub; then everything else that is swept in order
					const unsigned long long meta = W.hubMeta[0];
					const int hubP = (int)(uint32_t)(meta & 0xffffffffull);
					const bool pA = nsA && ids.z == hubP, pB = nsB && ids.w == hubP;
					bool wide = meta != 0ull && pA != pB;
					if (wide)
					{
						const int other = pA ? ids.w : ids.z;
						const bool otherNs = pA ? nsB : nsA;
						if (otherNs && W.deg[other] > HUB_DEGREE) wide = false;
						if (otherNs && W.hubFirst[other] != (((unsigned long long)(uint32_t)S->c.hubEpoch << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i))) wide = false;
					}
					if (wide) f = 1;
					else if (rowIsSerial(W, nsA, ids.z, nsB, ids.w)) f = 1 << 20;
				}
				else
				if (rowIsHubs(W, nsA, ids.z, nsB, ids.w)) f = 1;
				else if (rowIsSerial(W, nsA, ids.z, nsB, ids.w)) f = 1 << 20;
				// a constraint that found NO colour free on its two bodies (Counters::overflow bit 2; it sits in the hub group, its
				// stored colour says so - a hub's own constraints store none) is swept in order with the hub rows this step (round 6:
				// the step used to fail with "more than 64 constraint colours on one body") and asks for a colour again next step
				if (f == 0 && C.color[i] == HUB_COLOR) f = 1 << 20;
			}
		}
		W.keepFlag[i] = f;
	}
}

__global__ __launch_bounds__(256) void k_hub_fill(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int f = W.keepFlag[i];
		if (f == 1) W.hubList[W.keepScan[i] & 0xfffff] = W.hubRowOf[i];
		else if (f) W.hubList[(W.keepScan[n] & 0xfffff) + (W.keepScan[i] >> 20)] = W.hubRowOf[i];
	}
	// (k_color_scan sets this for the launch-per-colour layout; with rows grouped by block it is not run)
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		S->c.nHubRows = W.colorCount[colorSlot(HUB_COLOR)];
		S->c.nHubWide = W.hubWide ? (W.keepScan[n] & 0xfffff) : 0;
	}
}

// The ORDER in which the primary hub meets its partners (round 6): by the partner's highest colour, ascending (ties: contact
// index, the order k_hub_fill left). A hub row comes last on its partner - behind the partner's highest colour - so in a sweep
// the partners become ready for the hub in exactly this order: the rows of the first colours' partners can be swept while the
// last colours of the sweep (the REST rows, in the same launch: k_rest_hub) are still on their way. A fixed rule over the
// colouring: it does not depend on which colours a step sweeps as launches and which as rest rows.
// One workgroup: a stable counting sort of hubList[0, n) by 64 keys, through DW::keepFlag (free behind k_hub_fill).
// `allHubs`: the list's first segment holds the rows of ALL hubs (B2HIP_HUB_WIDE=0 with B2HIP_HUB_ORDER=1: the comparison
// runs of tests/test_gpu_sweep_end.py); the partner is the body of lower degree then.
#define HUB_ORDER_MAX 65536
__global__ __launch_bounds__(1024) void k_hub_order(DW W, int allHubs)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = allHubs ? (W.keepScan[S->c.nContacts] & 0xfffff) : S->c.nHubWide;
	if (n <= 1 || n > HUB_ORDER_MAX) return; // (uniform; beyond the limit the rows keep contact order: any order is a valid sweep)
	__shared__ int s_cnt[64], s_start[64], s_run[64];
	__shared__ int s_wcnt[16][64];
	const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
	if (t < 64) { s_cnt[t] = 0; s_run[t] = 0; }
	for (int i = t; i < 16 * 64; i += 1024) (&s_wcnt[0][0])[i] = 0;
	__syncthreads();
	auto keyOf = [&](int row) -> int
	{
		const int4 q = W.li_ref[row];
		const bool nsA = q.y >= 0, nsB = q.z >= 0;
		const int a = nsA ? q.y : -(q.y + 1), b = nsB ? q.z : -(q.z + 1);
		// the partner: the body that is not the hub (of two hubs: the one of lower degree)
		int partner = -1;
		if (nsA && nsB) partner = W.deg[a] <= W.deg[b] ? a : b;
		else if (nsA || nsB) partner = -1; // (the hub against a static body: ready at once)
		if (partner < 0) return 0;
		const unsigned long long m = W.bodyColorMask[partner] & ~(1ull << HUB_COLOR);
		return m == 0ull ? 0 : 64 - __clzll((long long)m); // 1 + the highest colour, <= 63
	};
	for (int k = t; k < n; k += 1024) atomicAdd(&s_cnt[keyOf(W.hubList[k])], 1);
	__syncthreads();
	if (t == 0)
	{
		int run = 0;
		for (int b = 0; b < 64; ++b) { s_start[b] = run; run += s_cnt[b]; }
	}
	__syncthreads();
	for (int base = 0; base < n; base += 1024) // (uniform)
	{
		const int k = base + t;
		const bool have = k < n;
		const int row = have ? W.hubList[k] : 0;
		const int key = have ? keyOf(row) : -1;
		// my rank among the lanes of this wave with my key, and the wave's count per key
		int rankW = 0;
		unsigned long long todo = __ballot(have);
		while (todo != 0ull)
		{
			const int src = __ffsll((long long)todo) - 1;
			const int k0 = __shfl(key, src);
			const unsigned long long m = __ballot(have && key == k0);
			if (have && key == k0) rankW = __popcll(m & ((1ull << lane) - 1ull));
			if (lane == src) s_wcnt[wave][k0] = __popcll(m);
			todo &= ~m;
		}
		__syncthreads();
		if (have)
		{
			int before = 0;
			for (int w2 = 0; w2 < wave; ++w2) before += s_wcnt[w2][key];
			W.keepFlag[s_start[key] + s_run[key] + before + rankW] = row;
		}
		__syncthreads();
		if (t < 64)
		{
			int tot = 0;
			for (int w2 = 0; w2 < 16; ++w2) { tot += s_wcnt[w2][t]; s_wcnt[w2][t] = 0; }
			s_run[t] += tot;
		}
		__syncthreads();
	}
	__syncthreads();
	// The constraints change PLACES among the rows they hold, so that the k-th of the new order sits in the k-th lowest of
	// those rows: the fixed point's lanes read consecutive rows again (field-major rows: 36 coalesced loads per wave - through
	// a permuted list every one of them touched a dozen lines, +10 us per pass of 512). Nothing has read the rows' content
	// yet (k_large_init fills them behind this kernel); what names a row - li_ref, hubRowOf - moves with the constraint.
	// keepFlag[0, n): the rows in the new order; keepScan[0, n) (free like keepFlag): their references, saved before any is overwritten.
	int4* const refs = (int4*)W.keepScan; // (n <= HUB_ORDER_MAX int4 = 1 MB: capContacts ints hold that from 262 144 contacts on)
	const bool room = (size_t)W.capContacts * sizeof(int) >= (size_t)n * sizeof(int4) + 64;
	if (!room)
	{
		for (int k = t; k < n; k += 1024) W.hubList[k] = W.keepFlag[k];
		return;
	}
	for (int k = t; k < n; k += 1024) refs[k] = W.li_ref[W.keepFlag[k]];
	__syncthreads();
	// the rows of the segment in ascending order: hubList still holds them in contact order, which is ascending in the row
	// only by accident (k_color_fill hands the places out by atomics) - rank them by a count over the list
	__shared__ int s_lo, s_hi;
	if (t == 0) { s_lo = 0x7fffffff; s_hi = -1; }
	__syncthreads();
	for (int k = t; k < n; k += 1024) { atomicMin(&s_lo, W.hubList[k]); atomicMax(&s_hi, W.hubList[k]); }
	__syncthreads();
	const int lo = s_lo, span = s_hi - s_lo + 1;
	// (a bitmap over [lo, hi] in keepFlag behind the list; the rows of one colour segment: span is n plus the handful of
	// leftover rows that share the segment)
	int* const bits = W.keepFlag + n;
	if ((size_t)n + (size_t)span + 1024 > (size_t)W.capContacts)
	{
		for (int k = t; k < n; k += 1024) W.hubList[k] = W.keepFlag[k];
		return;
	}
	for (int i = t; i < span; i += 1024) bits[i] = 0;
	__syncthreads();
	for (int k = t; k < n; k += 1024) bits[W.hubList[k] - lo] = 1;
	__syncthreads();
	__shared__ int s_base, s_wsum[16];
	if (t == 0) s_base = 0;
	__syncthreads();
	for (int b0 = 0; b0 < span; b0 += 1024) // (uniform)
	{
		const int i = b0 + t;
		const int f = i < span ? bits[i] : 0;
		const unsigned long long m = __ballot(f != 0);
		if (lane == 0) s_wsum[wave] = __popcll(m);
		__syncthreads();
		int before = s_base;
		for (int w2 = 0; w2 < wave; ++w2) before += s_wsum[w2];
		const int rank = before + __popcll(m & ((1ull << lane) - 1ull));
		if (f)
		{
			// the rank-th lowest row takes the rank-th constraint of the new order
			const int row = lo + i;
			const int4 q = refs[rank];
			W.li_ref[row] = q;
			W.hubRowOf[q.x] = row;
			W.hubList[rank] = row;
		}
		__syncthreads();
		if (t == 0) { int tot = 0; for (int w2 = 0; w2 < 16; ++w2) tot += s_wsum[w2]; s_base += tot; }
		__syncthreads();
	}
}

// ---- the hub list in ONE single-workgroup launch (round 6) -------------------------------------------------------------------------
// k_hub_flag (a pass over EVERY contact) + an exclusive scan of 2.6 M flags (three launches) + k_hub_fill + k_hub_order were
// 56 us of the Tumbler's step for a list of ~950 rows that already lie together: the rows of the hub group are one segment
// of the row array (k_color_fill put them there, in the order its atomics happened to run). This kernel sorts that segment in
// place by (class, partner's highest colour, contact index) - a rank sort: every row counts the keys below its own, the keys
// in LDS - and writes the list, the counts and what names a row (li_ref, hubRowOf) for the new places: hubList[k] = first + k.
//   class 0: what the fixed point of k_sweep_end takes (`wide`: the primary hub's constraint with a partner that is no hub and
//            meets the hub here for the first time) or, without it, every constraint of a hub; class 1: everything else that
//            is swept in order. The same predicates as k_hub_flag's, the same lists in the same order as k_hub_fill + k_hub_order.
//   `ordered`: class 0 by the partner's highest colour (k_hub_order's rule), else by contact index alone.
// Lists longer than HUB_BUILD_LDS rows keep their keys in memory (slow: n^2 / 1024 looks per lane - the host sends worlds whose
// last step had that many hub rows through the four launches instead).
#define HUB_BUILD_LDS 4096
__global__ __launch_bounds__(1024) void k_hub_build(DW W, int wideMode, int ordered, int blockSort)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int n = W.colorCount[colorSlot(HUB_COLOR)];
	int first = 0;
	if (blockSort) { const int nb = S->c.nBlocks < MAX_BLOCKS ? S->c.nBlocks : MAX_BLOCKS; first = W.blkRowStart[nb]; }
	else first = W.colorStart[HUB_COLOR];
	const int t = (int)threadIdx.x;
	__shared__ __attribute__((aligned(16))) unsigned long long s_key[HUB_BUILD_LDS + 8];
	__shared__ int s_wide;
	if (t == 0) s_wide = 0;
	if (n <= 0 || (size_t)n * 4 > (size_t)W.capContacts)
	{
		// (nothing to list - or a hub with more than a quarter of all contacts, beyond the scratch: the rows stay where they are)
		if (t == 0) { S->c.nHubRows = n > 0 ? n : 0; S->c.nHubWide = 0; }
		for (int k = t; k < n; k += 1024) W.hubList[k] = first + k;
		return;
	}
	const bool inLds = n <= HUB_BUILD_LDS;
	unsigned long long* const keys = inLds ? s_key : (unsigned long long*)W.keepScan;
	int4* const refs = (int4*)W.keepFlag;
	const unsigned long long meta = W.hubMeta[0];
	const int hubP = (int)(uint32_t)(meta & 0xffffffffull);
	__syncthreads();
	for (int j = t; j < n; j += 1024)
	{
		const int4 q = W.li_ref[first + j];
		const int ci = q.x;
		const bool nsA = q.y >= 0, nsB = q.z >= 0;
		const int a = nsA ? q.y : -(q.y + 1), b = nsB ? q.z : -(q.z + 1);
		int cls = 1;
		int partner = -1;
		if (wideMode)
		{
			const bool pA = nsA && a == hubP, pB = nsB && b == hubP;
			bool wide = meta != 0ull && pA != pB;
			if (wide)
			{
				const int other = pA ? b : a;
				const bool otherNs = pA ? nsB : nsA;
				if (otherNs && W.deg[other] > HUB_DEGREE) wide = false;
				if (otherNs && W.hubFirst[other] != (((unsigned long long)(uint32_t)S->c.hubEpoch << 32) | (unsigned long long)(0xffffffffu - (uint32_t)ci))) wide = false;
				if (wide && otherNs) partner = other;
			}
			cls = wide ? 0 : 1;
		}
		else
		{
			cls = rowIsHubs(W, nsA, a, nsB, b) ? 0 : 1;
			if (cls == 0 && nsA && nsB) partner = W.deg[a] <= W.deg[b] ? a : b;
		}
		int hi = 0;
		if (ordered && cls == 0 && partner >= 0)
		{
			const unsigned long long m = W.bodyColorMask[partner] & ~(1ull << HUB_COLOR);
			hi = m == 0ull ? 0 : 64 - __clzll((long long)m);
		}
		if (cls == 0) atomicAdd(&s_wide, 1);
		keys[j] = ((unsigned long long)cls << 62) | ((unsigned long long)hi << 40) | (unsigned long long)(uint32_t)ci;
		refs[j] = q;
	}
	// (the keys behind the last one never count: the loop below reads them eight at a time without asking where the list ends)
	if (inLds && t < 8) s_key[n + t] = ~0ull;
	__threadfence_block();
	__syncthreads();
	for (int j = t; j < n; j += 1024)
	{
		const unsigned long long mine = keys[j];
		int rank = 0;
		if (inLds)
		{
			// every lane reads the same words (broadcasts), eight keys per trip in four 16-byte reads that are in flight together: one
			// key per trip was a chain of ~950 LDS round trips, 40 us for the Tumbler's list
			typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
			const u64x2* k2 = (const u64x2*)s_key;
			for (int k = 0; k < n; k += 8)
			{
				const u64x2 v0 = k2[(k >> 1) + 0], v1 = k2[(k >> 1) + 1], v2 = k2[(k >> 1) + 2], v3 = k2[(k >> 1) + 3];
				rank += (v0.x < mine ? 1 : 0) + (v0.y < mine ? 1 : 0) + (v1.x < mine ? 1 : 0) + (v1.y < mine ? 1 : 0)
					+ (v2.x < mine ? 1 : 0) + (v2.y < mine ? 1 : 0) + (v3.x < mine ? 1 : 0) + (v3.y < mine ? 1 : 0);
			}
		}
		else
			for (int k = 0; k < n; ++k) rank += keys[k] < mine ? 1 : 0;
		const int4 q = refs[j];
		const int row = first + rank;
		W.li_ref[row] = q;
		W.hubRowOf[q.x] = row;
		W.hubList[rank] = row;
	}
	if (t == 0)
	{
		S->c.nHubRows = n;
		S->c.nHubWide = wideMode ? s_wide : 0;
	}
	(void)C;
}

// One constraint of a hub chunk, evaluated from the hub row `hubIn` the lane assumes it will meet at its turn. Works on
// copies: returns the hub row after the constraint, the partner row and the impulses it would leave.
struct HubTrial
{
	float4 hubOut, otherOut;
	float imp[4];
	float minSep;
};

__device__ __forceinline__ HubTrial hubEvaluate(int mode, ContactConstraint& cc, const float impIn[4], bool hubIsA, bool otherDynamic,
	float4 hubIn, float4 other)
{
	HubTrial r;
	r.minSep = 0.0f;
	if (mode == 2)
	{
		BodyPos pH, pO;
		pH.c = v2(hubIn.x, hubIn.y); pH.a = hubIn.z;
		pO.c = v2(other.x, other.y); pO.a = other.z;
		if (hubIsA) b2dSolvePosition(&cc, &pH, &pO, B2D_BAUMGARTE, &r.minSep);
		else b2dSolvePosition(&cc, &pO, &pH, B2D_BAUMGARTE, &r.minSep);
		r.hubOut = make_float4(pH.c.x, pH.c.y, pH.a, hubIn.w);
		r.otherOut = make_float4(pO.c.x, pO.c.y, pO.a, other.w);
	}
	else
	{
		cc.normalImpulse[0] = impIn[0]; cc.tangentImpulse[0] = impIn[1];
		cc.normalImpulse[1] = impIn[2]; cc.tangentImpulse[1] = impIn[3];
		BodyVel vH, vO;
		vH.v = v2(hubIn.x, hubIn.y); vH.w = hubIn.z;
		vO.v = v2(other.x, other.y); vO.w = other.z;
		if (!otherDynamic) { vO.v = v2(0, 0); vO.w = 0; }
		if (mode == 0)
		{
			if (hubIsA) b2dWarmStart(&cc, &vH, &vO); else b2dWarmStart(&cc, &vO, &vH);
		}
		else
		{
			if (hubIsA) b2dSolveVelocity(&cc, &vH, &vO); else b2dSolveVelocity(&cc, &vO, &vH);
		}
		r.hubOut = make_float4(vH.v.x, vH.v.y, vH.w, 0.0f);
		r.otherOut = make_float4(vO.v.x, vO.v.y, vO.w, 0.0f);
	}
	r.imp[0] = cc.normalImpulse[0]; r.imp[1] = cc.tangentImpulse[0];
	r.imp[2] = cc.normalImpulse[1]; r.imp[3] = cc.tangentImpulse[1];
	return r;
}

#define HUB_FIXPOINT_ROUNDS 24

// mode 0 warm start, 1 velocity, 2 position. Rows in hubList order, a chunk of 64 rows per wave.
// Per chunk: every lane fetches its constraint and its non-hub body in parallel. Then
//   * the common case - one hub, 64 different partners, none of them a hub itself: the sequential sweep through the hub is
//     found as a FIXED POINT. Every lane evaluates its constraint from the hub row it assumes it will meet; the changes it
//     makes to the hub row are prefix-summed over the lanes, which gives every lane a better assumption; repeat until no
//     lane's assumption moves by more than 2^-21 of the row. Lane k's assumption only depends on lanes < k, so after k
//     rounds it is final: the loop ends after at most 64 rounds with the rows a lane-after-lane sweep meets (up to that
//     2^-21 and the rounding of "row + sum of changes" against "row changed step by step"), and in practice after two or
//     three - a partner changes the hub's row by (its mass / the hub's mass), which is what an error shrinks by per round,
//     and the first assumption is what the previous sweep found (hubDelta);
//   * otherwise (or if HUB_FIXPOINT_ROUNDS were not enough) the lanes take turns: the hub's row travels from turn to turn
//     in registers (wave shuffle) as long as consecutive constraints sit on the same hub; a partner body that occurs twice
//     in a chunk is re-read at its turn.
// The chunks themselves are a chain (the hub row goes from chunk to chunk), but only the rounds are: with NW waves in the
// workgroup, wave w owns chunks w, w + NW, ... and FETCHES a chunk (three levels of dependent loads, ~6 us of the 13 us a
// chunk took in the one-wave form) while the waves before it have their turn. The hub row is handed from wave to wave
// through LDS (s_turn counts the chunks done); a wave that finds one of its prefetched partner rows was written by an
// earlier chunk of this launch (s_dirty, a bitmap over body ids) waits until those stores have landed (s_landed) and
// reads the row again. Every chunk computes from the same inputs as in the one-wave form: the result is the same bits
// for any NW (tests/test_gpu_parity.py::test_hub_sweep_is_the_same_for_any_number_of_waves).
#define HUB_DIRTY_WORDS 2048

// `first`: the sweep covers rows [first, nHubRows) of hubList (k_sweep_end takes the rows before that as one fixed point).
template <int NW>
__device__ __forceinline__ void hubSweep(const DW& W, int mode, int useGuess, int first = 0)
{
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int n = S->c.nHubRows;
	const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6);
	const float4* rows = mode == 2 ? W.b_pos : W.b_vel;
	float4* rowsOut = mode == 2 ? W.b_pos : W.b_vel;
	__shared__ float4 s_carry;
	__shared__ int s_carryBody, s_turn, s_landed;
	__shared__ uint32_t s_dirty[NW > 1 ? HUB_DIRTY_WORDS : 1];
	if (NW > 1)
	{
		for (int i = threadIdx.x; i < HUB_DIRTY_WORDS; i += 64 * NW) s_dirty[i] = 0u;
		if (threadIdx.x == 0)
		{
			s_carry = make_float4(0, 0, 0, 0);
			s_carryBody = -1;
			s_turn = 0;
			s_landed = 0;
		}
		__syncthreads();
	}
	auto markDirty = [&](int body)
	{
		if (NW > 1) atomicOr(&s_dirty[((uint32_t)body >> 5) & (HUB_DIRTY_WORDS - 1)], 1u << (body & 31));
	};
	auto isDirty = [&](int body) -> bool
	{
		return NW > 1 && (s_dirty[((uint32_t)body >> 5) & (HUB_DIRTY_WORDS - 1)] >> (body & 31)) & 1u;
	};
	int carryBody = -1;
	float4 carry = make_float4(0, 0, 0, 0);
	int statRounds = 0, statSerial = 0;
	const int nChunks = n > first ? (n - first + 63) / 64 : 0;
	for (int chunk = wave; chunk < nChunks; chunk += NW)
	{
		const int base = first + chunk * 64;
		const int k = base + lane;
		const bool have = k < n;
		LargeRef r;
		r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
		ContactConstraint cc;
		memset(&cc, 0, sizeof(cc));
		int row = 0;
		bool active = have;
		int hubBody = -1, otherBody = -1;
		bool hubIsA = true, otherDynamic = false;
		float4 other = make_float4(0, 0, 0, 0);
		if (have)
		{
			row = W.hubList[k];
			r = largeRef(W, C, row);
			// the body whose row is carried from turn to turn: the hub; for a constraint that is here for another reason
			// (rowIsSerial) any of its moving bodies
			hubIsA = r.nsA && (W.deg[r.bodyA] > HUB_DEGREE || !r.nsB || W.deg[r.bodyB] <= HUB_DEGREE);
			hubBody = hubIsA ? r.bodyA : r.bodyB;
			otherBody = hubIsA ? r.bodyB : r.bodyA;
			otherDynamic = hubIsA ? r.nsB : r.nsA;
			if (mode == 2)
			{
				active = W.rootDone[r.root] == 0;
				lcLoad(W, row, cc, LC_MASS_FIRST, LC_MASS_FIRST + 4);
				lcLoad(W, row, cc, LC_POS_FIRST, LC_WORDS);
				other = rows[otherBody]; // static partners have a position too
			}
			else
			{
				lcLoad(W, row, cc, 0, LC_VEL_WORDS);
				if (otherDynamic) other = rows[otherBody];
			}
		}
		// does my partner body occur earlier in this chunk (as partner or as hub)? then my copy may be stale at my turn
		bool reload = false;
		for (int t = 0; t < 64; ++t)
		{
			const int ob = __shfl(otherBody, t), hb = __shfl(hubBody, t);
			if (t < lane && have && (ob == otherBody || hb == otherBody)) reload = true;
		}
		float minSep = 0.0f;
		const int cnt = n - base < 64 ? n - base : 64;
		// ---- the fixed-point path ----------------------------------------------------------------------------------------
		const int hub0 = __shfl(hubBody, 0);
		bool simple = !W.hubSerial;
		{
			const bool partnerIsHub = have && W.deg[otherBody] > HUB_DEGREE && otherBody != hubBody;
			bool dup = false;
			for (int t = 0; t < cnt; ++t)
			{
				const int ob = __shfl(otherBody, t);
				const int od = __shfl(otherDynamic ? 1 : 0, t);
				if (t < lane && have && otherDynamic && od && ob == otherBody) dup = true;
			}
			if (__ballot(have && (hubBody != hub0 || partnerIsHub || dup || otherBody == hubBody)) != 0ull) simple = false;
		}

		// ---- my turn: the chunks before this one are through --------------------------------------------------------------------
		bool landedWaited = false;
		auto waitLanded = [&]()
		{
			// every store of the chunks before this one has reached memory (they may have written rows this chunk reads)
			if (NW > 1 && !landedWaited)
			{
				while (__hip_atomic_load(&s_landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < chunk) __builtin_amdgcn_s_sleep(1);
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				landedWaited = true;
			}
		};
		if (NW > 1)
		{
			while (__hip_atomic_load(&s_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < chunk) __builtin_amdgcn_s_sleep(1);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			carry = s_carry;
			carryBody = s_carryBody;
			// a partner row fetched ahead that an earlier chunk has written since: read it again
			const bool stale = have && (mode == 2 || otherDynamic) && isDirty(otherBody);
			if (__ballot(stale) != 0ull)
			{
				waitLanded();
				if (stale) other = rows[otherBody];
			}
			if (!simple) waitLanded(); // (the turn-by-turn path reads rows as it goes)
		}
		if (have && otherBody == carryBody) reload = true; // its memory row is stale while it is carried
		bool solved = false;
		if (simple)
		{
			if (carryBody >= 0 && carryBody != hub0)
			{
				if (lane == 0) { rowsOut[carryBody] = carry; markDirty(carryBody); }
				carryBody = -1;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			}
			if (carryBody != hub0) waitLanded(); // (the hub's row comes from memory: an earlier chunk may have put it there)
			const float4 u0 = carryBody == hub0 ? carry : rows[hub0];
			float imp0[4] = { cc.normalImpulse[0], cc.tangentImpulse[0], cc.normalImpulse[1], cc.tangentImpulse[1] };
			float4 incoming = u0;
			if (useGuess)
			{
				// start from what the previous sweep of the same kind found: the changes this constraint made to the hub row
				// then (they move little from sweep to sweep, so the first assumption is already close)
				float4 g = make_float4(0, 0, 0, 0);
				if (have) g = W.hubDelta[k];
				float sx = g.x, sy = g.y, sz = g.z;
				for (int off = 1; off < 64; off <<= 1)
				{
					const float ux = __shfl_up(sx, off), uy = __shfl_up(sy, off), uz = __shfl_up(sz, off);
					if (lane >= off) { sx += ux; sy += uy; sz += uz; }
				}
				incoming = make_float4(u0.x + (sx - g.x), u0.y + (sy - g.y), u0.z + (sz - g.z), u0.w);
			}
			HubTrial tr;
			float lastDx = 0.0f, lastDy = 0.0f, lastDz = 0.0f;
			for (int round = 0; round < HUB_FIXPOINT_ROUNDS + 1 && !solved; ++round)
			{
				float dx = 0.0f, dy = 0.0f, dz = 0.0f;
				if (active)
				{
					tr = hubEvaluate(mode, cc, imp0, hubIsA, otherDynamic, incoming, other);
					dx = tr.hubOut.x - incoming.x;
					dy = tr.hubOut.y - incoming.y;
					dz = tr.hubOut.z - incoming.z;
				}
				else
				{
					tr.hubOut = incoming;
				}
				// exclusive prefix sums of the changes, in a fixed tree order
				float sx = dx, sy = dy, sz = dz;
				for (int off = 1; off < 64; off <<= 1)
				{
					const float ux = __shfl_up(sx, off), uy = __shfl_up(sy, off), uz = __shfl_up(sz, off);
					if (lane >= off) { sx += ux; sy += uy; sz += uz; }
				}
				const float4 next = make_float4(u0.x + (sx - dx), u0.y + (sy - dy), u0.z + (sz - dz), u0.w);
				// settled = no lane's assumption moves by more than 2^-21 of the hub row - a 64th of the rounding a sweep over 500
				// constraints accumulates in that row anyway (or of the sweep's total change to it, if
				// that is larger). Not "by a bit": a lane's change is measured as (row after - row before), which carries the
				// rounding of the row itself, so assumptions keep flickering in their last bit long after the sweep is decided
				const float tx = 0x1p-21f * fmaxf(fabsf(u0.x), fabsf(__shfl(sx, 63)));
				const float ty = 0x1p-21f * fmaxf(fabsf(u0.y), fabsf(__shfl(sy, 63)));
				const float tz = 0x1p-21f * fmaxf(fabsf(u0.z), fabsf(__shfl(sz, 63)));
				const bool changed = fabsf(next.x - incoming.x) > tx || fabsf(next.y - incoming.y) > ty || fabsf(next.z - incoming.z) > tz;
				incoming = next;
				lastDx = dx; lastDy = dy; lastDz = dz;
				++statRounds;
				if (__ballot(changed) == 0ull) solved = true;
			}
			if (solved)
			{
				// every lane met the hub row it assumed: what it computed last stands
				const float4 last = tr.hubOut; // (an inactive lane hands its assumption on)
				carry.x = __shfl(last.x, cnt - 1);
				carry.y = __shfl(last.y, cnt - 1);
				carry.z = __shfl(last.z, cnt - 1);
				carry.w = u0.w;
				carryBody = hub0;
				if (NW > 1)
				{
					// the next chunk can start its rounds: it needs the hub row, and to know which partner rows are on their way
					if (active && otherDynamic) markDirty(otherBody);
					if (lane == 0) { s_carry = carry; s_carryBody = carryBody; }
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					if (lane == 0) __hip_atomic_store(&s_turn, chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
				if (active)
				{
					if (mode != 2)
					{
						cc.normalImpulse[0] = tr.imp[0]; cc.tangentImpulse[0] = tr.imp[1];
						cc.normalImpulse[1] = tr.imp[2]; cc.tangentImpulse[1] = tr.imp[3];
					}
					if (otherDynamic) rowsOut[otherBody] = tr.otherOut;
					minSep = tr.minSep;
				}
				if (have) W.hubDelta[k] = make_float4(lastDx, lastDy, lastDz, 0.0f);
				// the partner rows written here may be read by the next chunk
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			}
			else if (mode != 2)
			{
				cc.normalImpulse[0] = imp0[0]; cc.tangentImpulse[0] = imp0[1];
				cc.normalImpulse[1] = imp0[2]; cc.tangentImpulse[1] = imp0[3];
			}
		}
		// ---- lanes take turns ----------------------------------------------------------------------------------------------
		if (!solved)
		{
			++statSerial;
			waitLanded();
		}
		for (int t = 0; t < cnt && !solved; ++t)
		{
			float4 hubOut = carry;
			int hubOutBody = carryBody;
			if (lane == t && active)
			{
				int cb = carryBody;
				if (cb >= 0 && cb == otherBody)
				{
					// hub-to-hub contact: the partner is the body being carried; put it back first
					rowsOut[cb] = carry;
					markDirty(cb);
					cb = -1;
				}
				float4 hubRow = cb == hubBody ? carry : rows[hubBody];
				if (reload && (mode == 2 || otherDynamic)) other = rows[otherBody];
				if (mode == 2)
				{
					BodyPos pH, pO;
					pH.c = v2(hubRow.x, hubRow.y); pH.a = hubRow.z;
					pO.c = v2(other.x, other.y); pO.a = other.z;
					if (hubIsA) b2dSolvePosition(&cc, &pH, &pO, B2D_BAUMGARTE, &minSep);
					else b2dSolvePosition(&cc, &pO, &pH, B2D_BAUMGARTE, &minSep);
					hubRow = make_float4(pH.c.x, pH.c.y, pH.a, hubRow.w);
					if (otherDynamic) { rowsOut[otherBody] = make_float4(pO.c.x, pO.c.y, pO.a, other.w); markDirty(otherBody); }
				}
				else
				{
					BodyVel vH, vO;
					vH.v = v2(hubRow.x, hubRow.y); vH.w = hubRow.z;
					vO.v = v2(other.x, other.y); vO.w = other.z;
					if (!otherDynamic) { vO.v = v2(0, 0); vO.w = 0; }
					if (mode == 0)
					{
						if (hubIsA) b2dWarmStart(&cc, &vH, &vO); else b2dWarmStart(&cc, &vO, &vH);
					}
					else
					{
						if (hubIsA) b2dSolveVelocity(&cc, &vH, &vO); else b2dSolveVelocity(&cc, &vO, &vH);
					}
					hubRow = make_float4(vH.v.x, vH.v.y, vH.w, 0.0f);
					if (otherDynamic) { rowsOut[otherBody] = make_float4(vO.v.x, vO.v.y, vO.w, 0.0f); markDirty(otherBody); }
				}
				// a different hub was being carried: its row goes back to memory now
				if (cb >= 0 && cb != hubBody) { rowsOut[cb] = carry; markDirty(cb); }
				hubOut = hubRow;
				hubOutBody = hubBody;
			}
			// hand the hub row to the next turn; order this lane's stores before the next lane's loads
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			carry.x = __shfl(hubOut.x, t);
			carry.y = __shfl(hubOut.y, t);
			carry.z = __shfl(hubOut.z, t);
			carry.w = __shfl(hubOut.w, t);
			carryBody = __shfl(hubOutBody, t);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		if (NW > 1 && !solved)
		{
			if (lane == 0) { s_carry = carry; s_carryBody = carryBody; }
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) __hip_atomic_store(&s_turn, chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		if (mode == 1 && have) lcStore(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
		if (mode == 2) waveAtomicMaxU32(W.rootPen, r.root, floatBits(0.0f - minSep), have && active);
		if (NW > 1)
		{
			// my stores have landed - and, in order, those of every chunk before mine
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			while (__hip_atomic_load(&s_landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < chunk) __builtin_amdgcn_s_sleep(1);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) __hip_atomic_store(&s_landed, chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	if (NW > 1)
	{
		// the row still being carried goes back to memory: by the wave that had the last turn
		if (nChunks > 0 && wave == (nChunks - 1) % NW && lane == 0 && carryBody >= 0) rowsOut[carryBody] = carry;
	}
	else if (lane == 0 && carryBody >= 0) rowsOut[carryBody] = carry;
	if (lane == 0 && (statRounds | statSerial))
	{
		atomicAdd(&S->c.hubRounds, statRounds);
		atomicAdd(&S->c.hubSerialChunks, statSerial);
	}
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_large_hub(DW W, int mode, int useGuess, int behindWide)
{
	b2dPhaseStamp(W);
	if (mode == 2 && W.st->c.allLargeDone) return;
	// (behindWide: only the rows k_sweep_end's one fixed point left - those behind Counters::nHubWide)
	hubSweep<NW>(W, mode, useGuess, behindWide ? W.st->c.nHubWide : 0);
}

__global__ __launch_bounds__(256) void k_large_init(DW W, StepParams sp, int warmDeltas)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x)
	{
		LargeRef r = largeRef(W, C, row);
		int4 ids = C.ids[r.ci];
		float4 pa = W.b_pos[r.bodyA], pb = W.b_pos[r.bodyB];
		float4 va = W.b_vel[r.bodyA], vb = W.b_vel[r.bodyB];
		float4 mA4 = W.b_mass[r.bodyA], mB4 = W.b_mass[r.bodyB];
		BodyPos pA, pB;
		BodyVel vA, vB;
		pA.c = v2(pa.x, pa.y); pA.a = pa.z;
		pB.c = v2(pb.x, pb.y); pB.a = pb.z;
		vA.v = r.nsA ? v2(va.x, va.y) : v2(0, 0); vA.w = r.nsA ? va.z : 0.0f;
		vB.v = r.nsB ? v2(vb.x, vb.y) : v2(0, 0); vB.w = r.nsB ? vb.z : 0.0f;
		float4 cmat = C.mat[r.ci];
		float4 m0 = C.man0[r.ci], m1 = C.man1[r.ci], im = C.imp[r.ci];
		int4 m3 = C.man3[r.ci];
		Manifold mf;
		mf.localNormal = v2(m0.x, m0.y);
		mf.localPoint = v2(m0.z, m0.w);
		mf.p[0] = v2(m1.x, m1.y);
		mf.p[1] = v2(m1.z, m1.w);
		mf.ni[0] = im.x; mf.ti[0] = im.y; mf.ni[1] = im.z; mf.ti[1] = im.w;
		mf.id[0] = (uint32_t)m3.x; mf.id[1] = (uint32_t)m3.y;
		mf.type = m3.z;
		mf.pointCount = m3.w;
		ContactConstraint cc;
		b2dInitConstraint(&cc, &mf, cmat.x, cmat.y, cmat.z,
			mA4.x, mA4.y, v2(mA4.z, mA4.w), W.shapes[W.p_shape[ids.x]].radius,
			mB4.x, mB4.y, v2(mB4.z, mB4.w), W.shapes[W.p_shape[ids.y]].radius,
			pA, vA, pB, vB, sp.warmStarting != 0, sp.dtRatio);
		lcStore(W, row, cc, 0, LC_WORDS);
		if (warmDeltas && sp.warmStarting)
		{
			// what b2dWarmStart would subtract from body A and add to body B, point by point - the same expressions, so the same
			// floats: k_large_warm applies them body by body in the order the colour launches would (b2d_solver.h: b2dWarmStart)
			const float mA = cc.invMassA, iA = cc.invIA, mB = cc.invMassB, iB = cc.invIB;
			const V2 normal = cc.normal;
			const V2 tangent = b2dCrossVS(normal, 1.0f);
			float4 q[4];
			q[0] = q[1] = q[2] = q[3] = make_float4(0, 0, 0, 0);
			B2D_FOR_POINTS(j, cc.pointCount)
			{
				const V2 P = cc.normalImpulse[j] * normal + cc.tangentImpulse[j] * tangent;
				const float dwA = iA * b2dCross(cc.rA[j], P);
				const V2 dvA = mA * P;
				const float dwB = iB * b2dCross(cc.rB[j], P);
				const V2 dvB = mB * P;
				q[j] = make_float4(dvA.x, dvA.y, dwA, 0.0f);
				q[2 + j] = make_float4(dvB.x, dvB.y, dwB, 0.0f);
			}
			q[0].w = __int_as_float(cc.pointCount);
			q[2].w = __int_as_float(cc.pointCount);
			float4* dst = W.warmDelta + (size_t)row * 4;
			dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];
		}
	}
}

// The warm start of the launch-per-colour solver in ONE launch, body by body. A warm start only ADDS what the constraints
// carried over from the last step (b2ContactSolver::WarmStart, b2ContactSolver.cpp:253-291): no constraint reads a velocity, so
// the only order that matters is the order of the additions ON EVERY BODY - and every body has its adjacency segment and every
// row its colour, so a lane takes a body, walks its rows in ascending colour (the order the colour launches would have
// visited it in) and applies the deltas k_large_init stored: the same additions in the same order, the same bits, one launch
// where a sweep of 12 - 20 launches was. The rows of a hub are left to k_sweep_end (their partners meet them last either way).
__global__ __launch_bounds__(256) void k_large_warm(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLBodies;
	// a lane's rows as keys (colour << 26 | row), sorted in LDS: entry e of lane t at [e * 256 + t] (no bank conflicts);
	// a body below the hub threshold has at most HUB_DEGREE = 30 rows
	__shared__ uint32_t s_key[32 * 256];
	const int t = (int)threadIdx.x;
	for (int k = blockIdx.x * blockDim.x + t; k < n; k += gridDim.x * blockDim.x)
	{
		const int body = W.li_bodies[k];
		const int d = W.deg[body];
		if (d <= 0 || d > HUB_DEGREE) continue; // (a hub's rows are all hub rows)
		const int s = W.adjStart[body];
		int m = 0;
		for (int e = 0; e < d; ++e)
		{
			const int row = W.hubRowOf[W.adj[s + e]];
			const int c = W.rowColor[row];
			if (c < 0 || c >= MAX_COLORS || c == HUB_COLOR) continue; // (the hub's rows: k_sweep_end)
			// insertion into the sorted prefix (colours of one body are all different)
			const uint32_t key = ((uint32_t)c << 26) | (uint32_t)row;
			int j = m;
			while (j > 0 && s_key[(j - 1) * 256 + t] > key) { s_key[j * 256 + t] = s_key[(j - 1) * 256 + t]; --j; }
			s_key[j * 256 + t] = key;
			++m;
		}
		float4 v4 = W.b_vel[body];
		V2 v = v2(v4.x, v4.y);
		float w = v4.z;
		for (int e = 0; e < m; ++e)
		{
			const int row = (int)(s_key[e * 256 + t] & 0x3ffffffu);
			const bool sideA = W.li_ref[row].y == body;
			const float4* q = W.warmDelta + (size_t)row * 4 + (sideA ? 0 : 2);
			const float4 q0 = q[0], q1 = q[1];
			const int pc = __float_as_int(q0.w);
			if (sideA)
			{
				if (pc > 0) { w -= q0.z; v -= v2(q0.x, q0.y); }
				if (pc > 1) { w -= q1.z; v -= v2(q1.x, q1.y); }
			}
			else
			{
				if (pc > 0) { w += q0.z; v += v2(q0.x, q0.y); }
				if (pc > 1) { w += q1.z; v += v2(q1.x, q1.y); }
			}
		}
		W.b_vel[body] = make_float4(v.x, v.y, w, 0.0f);
	}
}

// A colour launch is a chain of dependent loads, and its first links are the kernel ARGUMENTS: the pointer block is 1.9 KB in a
// fresh kernarg slot (cold in every cache), the compiler loads a field where it is first used, and the phase-stamp branch at
// the top splits those loads into three rounds of s_load + wait before the first row is asked for (the disassembly of round
// 5's k_large_velocity: offsets 0x8..0x3c0, wait, 0x1c / 0x698 / 0x6ac, wait, colorStart, wait, 0x6a0, wait). Naming every
// field the kernel will need as an input of ONE empty asm statement makes them live - loaded - at the top: one round.
// mode 0 = warm start, 1 = velocity iteration
__global__ __launch_bounds__(256) void k_large_velocity(DW W, int color, int mode)
{
	asm volatile("" :: "s"(W.capContacts), "s"(color), "s"(mode), "s"((unsigned long long)(uintptr_t)W.colorStart), "s"((unsigned long long)(uintptr_t)W.li_ref),
		"s"((unsigned long long)(uintptr_t)W.lc), "s"((unsigned long long)(uintptr_t)W.b_vel), "s"((unsigned)blockDim.x), "s"((unsigned)gridDim.x), "s"(W.stampMask));
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int begin = W.colorStart[color], end = W.colorStart[color + 1];
	for (int row = begin + blockIdx.x * blockDim.x + threadIdx.x; row < end; row += gridDim.x * blockDim.x)
	{
		LargeRef r = largeRef(W, C, row);
		ContactConstraint cc;
		memset(&cc, 0, sizeof(cc));
		lcLoad(W, row, cc, 0, LC_VEL_WORDS);
		BodyVel vA, vB;
		vA.v = v2(0, 0); vA.w = 0; vB = vA;
		if (r.nsA) { float4 v = W.b_vel[r.bodyA]; vA.v = v2(v.x, v.y); vA.w = v.z; }
		if (r.nsB) { float4 v = W.b_vel[r.bodyB]; vB.v = v2(v.x, v.y); vB.w = v.z; }
		if (mode == 0)
		{
			b2dWarmStart(&cc, &vA, &vB);
		}
		else
		{
			b2dSolveVelocity(&cc, &vA, &vB);
			lcStore(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
		}
		if (r.nsA) W.b_vel[r.bodyA] = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
		if (r.nsB) W.b_vel[r.bodyB] = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
	}
}

// Joints of the large islands: one lane per island walks its joints in order (they are solved before the
// contacts in every velocity iteration and after them in every position iteration, b2Island.cpp:259-335).
// mode 0 = InitVelocityConstraints (+ joint warm start), 1 = velocity, 2 = position
__global__ __launch_bounds__(64) void k_large_joints(DW W, StepParams sp, int mode)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (mode == 2 && S->c.allLargeDone) return;
	const int n = S->c.nLIslands;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int root = W.li_roots[k];
		const int nj = W.rootJoints[root];
		if (nj == 0) continue;
		if (mode == 2 && W.rootDone[root]) continue;
		const int start = W.rootJointStart[root];
		JointBodiesGlobal bodies(W);
		const int okay = b2dSolveIslandJoints(W, sp, mode, start, nj, bodies);
		if (mode == 2) W.rootJointOkay[root] = okay;
	}
}

__global__ __launch_bounds__(256) void k_large_store_impulses(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x)
	{
		const int ci = W.li_ref[row].x;
		ContactConstraint cc;
		memset(&cc, 0, sizeof(cc));
		lcLoad(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
		lcLoad(W, row, cc, 35, 36);
		if (W.postSolveOn)
		{
			lcLoad(W, row, cc, LC_WORDS - 1, LC_WORDS); // pcPointCount: the manifold's own count
			if (cc.pointCount < cc.pcPointCount) C.flags[ci] |= CF_VC_ONE_POINT;
		}
		float4 im = C.imp[ci];
		if (cc.pointCount > 0) { im.x = cc.normalImpulse[0]; im.y = cc.tangentImpulse[0]; }
		if (cc.pointCount > 1) { im.z = cc.normalImpulse[1]; im.w = cc.tangentImpulse[1]; }
		C.imp[ci] = im;
	}
}

__global__ __launch_bounds__(256) void k_large_integrate_positions(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLBodies;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int body = W.li_bodies[k];
		float4 p = W.b_pos[body], v = W.b_vel[body];
		V2 c = v2(p.x, p.y), vv = v2(v.x, v.y);
		float a = p.z, w = v.z;
		b2dIntegratePosition(&c, &a, &vv, &w, sp.dt);
		W.b_pos[body] = make_float4(c.x, c.y, a, p.w);
		W.b_vel[body] = make_float4(vv.x, vv.y, w, 0.0f);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) S->c.allLargeDone = 0;
}

__global__ __launch_bounds__(256) void k_large_pos_begin(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.allLargeDone) return;
	const int n = S->c.nLIslands;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		W.rootPen[W.li_roots[k]] = 0;
		W.rootJointOkay[W.li_roots[k]] = 1;
	}
}

// Between the last velocity sweep and the first position iteration, in ONE launch (round 6: three launches of ~5, ~17 and ~5 us
// in a chain of dependent launches): k_large_store_impulses (rows), k_large_integrate_positions (bodies), k_large_pos_begin
// (islands) - three loops over different things, none of which reads what another writes.
__global__ __launch_bounds__(256) void k_large_after_velocity(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	{
		const int n = S->c.nLContacts;
		for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x)
		{
			const int ci = W.li_ref[row].x;
			ContactConstraint cc;
			memset(&cc, 0, sizeof(cc));
			lcLoad(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
			lcLoad(W, row, cc, 35, 36);
			if (W.postSolveOn)
			{
				lcLoad(W, row, cc, LC_WORDS - 1, LC_WORDS); // pcPointCount: the manifold's own count
				if (cc.pointCount < cc.pcPointCount) C.flags[ci] |= CF_VC_ONE_POINT;
			}
			float4 im = C.imp[ci];
			if (cc.pointCount > 0) { im.x = cc.normalImpulse[0]; im.y = cc.tangentImpulse[0]; }
			if (cc.pointCount > 1) { im.z = cc.normalImpulse[1]; im.w = cc.tangentImpulse[1]; }
			C.imp[ci] = im;
		}
	}
	{
		const int n = S->c.nLBodies;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		{
			const int body = W.li_bodies[k];
			float4 p = W.b_pos[body], v = W.b_vel[body];
			V2 c = v2(p.x, p.y), vv = v2(v.x, v.y);
			float a = p.z, w = v.z;
			b2dIntegratePosition(&c, &a, &vv, &w, sp.dt);
			W.b_pos[body] = make_float4(c.x, c.y, a, p.w);
			W.b_vel[body] = make_float4(vv.x, vv.y, w, 0.0f);
		}
	}
	{
		const int n = S->c.nLIslands;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		{
			W.rootPen[W.li_roots[k]] = 0;
			W.rootJointOkay[W.li_roots[k]] = 1;
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) S->c.allLargeDone = 0;
}

__global__ __launch_bounds__(256) void k_large_position(DW W, int color)
{
	asm volatile("" :: "s"(W.capContacts), "s"(color), "s"((unsigned long long)(uintptr_t)W.colorStart), "s"((unsigned long long)(uintptr_t)W.st),
		"s"((unsigned long long)(uintptr_t)W.li_ref), "s"((unsigned long long)(uintptr_t)W.lc), "s"((unsigned long long)(uintptr_t)W.b_pos),
		"s"((unsigned long long)(uintptr_t)W.rootDone), "s"((unsigned long long)(uintptr_t)W.rootPen), "s"((unsigned)blockDim.x), "s"((unsigned)gridDim.x), "s"(W.stampMask));
	b2dPhaseStamp(W);
	DState* S = W.st;
	// (the three scalar loads together, THEN the branch: the early-out's word is a dependent load at the head of a kernel that
	// is a chain of dependent loads - asked for on its own it cost a memory round trip per position launch)
	const int allDone = S->c.allLargeDone;
	const int begin = W.colorStart[color], end = W.colorStart[color + 1];
	if (allDone) return;
	const ContactArrays& C = W.ca[S->cur];
	BlockMaxU32 pen;
	pen.key = -1; pen.val = 0u;
	for (int base = begin + blockIdx.x * blockDim.x; base < end; base += gridDim.x * blockDim.x)
	{
		const int row = base + threadIdx.x;
		bool valid = row < end;
		LargeRef r;
		r.root = 0;
		if (valid)
		{
			r = largeRef(W, C, row);
			valid = W.rootDone[r.root] == 0;
		}
		float minSep = 0.0f;
		if (valid)
		{
			ContactConstraint cc;
			memset(&cc, 0, sizeof(cc));
			lcLoad(W, row, cc, LC_MASS_FIRST, LC_MASS_FIRST + 4);
			lcLoad(W, row, cc, LC_POS_FIRST, LC_WORDS);
			float4 pa = W.b_pos[r.bodyA], pb = W.b_pos[r.bodyB];
			BodyPos pA, pB;
			pA.c = v2(pa.x, pa.y); pA.a = pa.z;
			pB.c = v2(pb.x, pb.y); pB.a = pb.z;
			b2dSolvePosition(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
			if (r.nsA) W.b_pos[r.bodyA] = make_float4(pA.c.x, pA.c.y, pA.a, pa.w);
			if (r.nsB) W.b_pos[r.bodyB] = make_float4(pB.c.x, pB.c.y, pB.a, pb.w);
		}
		blockMaxU32Offer(pen, W.rootPen, r.root, floatBits(0.0f - minSep), valid);
	}
	blockMaxU32Flush(pen, W.rootPen);
}

// After all colours of one position iteration: per-island early out (b2Island.cpp:329-334).
__global__ __launch_bounds__(256) void k_large_pos_end(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.allLargeDone) return;
	__shared__ int s_open;
	if (threadIdx.x == 0) s_open = 0;
	__syncthreads();
	const int n = S->c.nLIslands;
	int open = 0;
	for (int k = threadIdx.x; k < n; k += blockDim.x)
	{
		const int root = W.li_roots[k];
		if (W.rootDone[root]) continue;
		float minSeparation = -__uint_as_float(W.rootPen[root]);
		if (minSeparation >= -3.0f * B2D_LINEAR_SLOP && W.rootJointOkay[root])
		{
			W.rootDone[root] = 1;
		}
		else
		{
			++open;
		}
	}
	if (open) atomicAdd(&s_open, open);
	__syncthreads();
	if (threadIdx.x == 0)
	{
		S->c.posItersLarge += 1;
		if (s_open == 0) S->c.allLargeDone = 1;
	}
}

// Write back + SynchronizeTransform + sleep timers (b2Island.cpp:338-382)
__global__ __launch_bounds__(256) void k_large_finalize(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLBodies;
	for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x)
	{
		const int k = base + threadIdx.x;
		const bool valid = k < n;
		int root = 0;
		float sleepTime = 0.0f;
		if (valid)
		{
			const int body = W.li_bodies[k];
			float4 p = W.b_pos[body], v = W.b_vel[body], m = W.b_mass[body];
			Xf xf = b2dXfFromSweep(v2(p.x, p.y), p.z, v2(m.z, m.w));
			W.b_xf[body] = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
			root = W.parent[body];
			if (sp.allowSleep)
			{
				const float linTolSqr = B2D_LINEAR_SLEEP_TOL * B2D_LINEAR_SLEEP_TOL;
				const float angTolSqr = B2D_ANGULAR_SLEEP_TOL * B2D_ANGULAR_SLEEP_TOL;
				uint32_t f = W.b_flags[body];
				sleepTime = p.w;
				if ((f & BF_AUTOSLEEP) == 0 || v.z * v.z > angTolSqr || b2dDot(v2(v.x, v.y), v2(v.x, v.y)) > linTolSqr)
				{
					sleepTime = 0.0f;
				}
				else
				{
					sleepTime += sp.dt;
				}
				W.b_pos[body] = make_float4(p.x, p.y, p.z, sleepTime);
			}
		}
		if (sp.allowSleep) waveAtomicMinU32(W.rootSleepMin, root, floatBits(sleepTime), valid);
	}
}

__global__ __launch_bounds__(256) void k_large_sleep(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (!sp.allowSleep) return;
	const int n = S->c.nLBodies;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int body = W.li_bodies[k];
		const int root = W.parent[body];
		float minSleepTime = __uint_as_float(W.rootSleepMin[root]);
		if (minSleepTime >= B2D_TIME_TO_SLEEP && W.rootDone[root])
		{
			float4 p = W.b_pos[body];
			W.b_flags[body] &= ~BF_AWAKE;
			W.b_pos[body] = make_float4(p.x, p.y, p.z, 0.0f);
			W.b_vel[body] = make_float4(0, 0, 0, 0);
			W.b_force[body] = make_float4(0, 0, 0, 0);
		}
	}
}

#endif
