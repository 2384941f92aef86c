// b2d_kernels_solve_blocks.h - the large-island solver on a block partition: bodies in LDS, boundary bodies through memory.
//
// k_solve_mailbox (b2d_kernels_solve_mailbox.h) hands EVERY body row from constraint to constraint through memory: a sweep
// over a coloured island costs one cross-workgroup hand-off (~3.3 us on loaded CUs) per colour, 8 colours x 9 sweeps on the
// 10 011-box pyramid = 209 us of pure latency for 4.7 MB of algorithmic traffic per sweep.
//
// Here the bodies of the large islands are partitioned into spatial blocks (Morton order of their positions, cut where
// the contact degrees add up to BLOCK_TARGET_DEG; the home block of a body is persistent state, b_blk1). ONE workgroup
// solves one block: its bodies' velocity / position rows live in LDS, its constraints (one per lane, in registers, as in
// the mailbox kernel) read and write them there, and a colour boundary is a workgroup barrier. Only a constraint between
// bodies of two blocks (a CUT constraint) needs memory - and the colouring knows the partition: cut constraints own the
// colours of the upper range (CUT_COLOR_BASE..), so in a sweep every body sees its in-block constraints first and its cut
// constraints last. Per sweep and block:
//     interior colours (LDS, one s_barrier each)
//  -> the home lanes publish the rows of the boundary bodies (one tagged 16-byte sc1 store per body)
//  -> upper-range constraints: poll both body rows for the expected update count, solve, publish both (dataflow, as in
//     k_solve_dataflow; a body with k upper-range constraints is handed over k + 1 times per sweep)
//  -> the home lanes take the final rows of their boundary bodies back into LDS.
// A sweep costs (interior colours) workgroup barriers + (upper-range colours on the busiest body + 1) memory hand-offs
// instead of (all colours) memory hand-offs, and the hand-offs are few (the block surfaces), so the CUs' memory queues -
// which price a hand-off - are nearly idle. No grid barrier in the velocity phase at all; one per position iteration
// for the islands' convergence verdicts (b2Island.cpp:329-334), none when the world's large islands fit one block.
//
// The visiting order is the colour order on every body (interior colours of different blocks never share a non-static body),
// so the floats are bit-identical to the launch-per-colour path (k_large_velocity ...) under the same colouring
// (tests/test_gpu_parity.py::test_block_solver_matches_launch_per_colour).
//
// Reference: b2Island::Solve (b2Island.cpp:184-396), b2ContactSolver (b2ContactSolver.cpp:47-843).
#ifndef B2D_KERNELS_SOLVE_BLOCKS_H
#define B2D_KERNELS_SOLVE_BLOCKS_H

#include "b2d_handover.h"

// ---- exclusive scan + maximum of up to 1024 ints held one per lane (1024-lane workgroup) -------------------------------------
__device__ __forceinline__ int blockScan1024(int v, int* s_buf /* [2 * 1024] */, int* total, int* maximum)
{
	const int t = threadIdx.x;
	int* a = s_buf;
	int* b = s_buf + 1024;
	a[t] = v;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1)
	{
		b[t] = a[t] + (t >= off ? a[t - off] : 0);
		__syncthreads();
		int* tmp = a; a = b; b = tmp;
	}
	const int incl = a[t];
	if (total) *total = a[1023];
	__syncthreads();
	if (maximum)
	{
		b[t] = v;
		__syncthreads();
		for (int off = 512; off > 0; off >>= 1)
		{
			if (t < off) b[t] = b[t] > b[t + off] ? b[t] : b[t + off];
			__syncthreads();
		}
		*maximum = b[0];
		__syncthreads();
	}
	return incl - v;
}

// After k_color_check's census (rows per block): row segments of the blocks, home bodies grouped by block (and each
// body's slot in its block), adoptions made permanent, and the two capacity figures the host decides on.
__global__ __launch_bounds__(1024) void k_block_census(DW W, DState* pub)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	__shared__ int s_buf[2048];
	__shared__ int s_cnt[MAX_BLOCKS], s_start[MAX_BLOCKS];
	const int t = threadIdx.x;
	const int nb = S->c.nBlocks < MAX_BLOCKS ? S->c.nBlocks : MAX_BLOCKS;
	int total = 0, mx = 0;
	const int rows = t < nb ? W.blkRows[(size_t)t * BLK_SLOT] : 0;
	const int rowStart = blockScan1024(rows, s_buf, &total, &mx);
	if (t < nb) W.blkRowStart[t] = rowStart;
	if (t == 0)
	{
		W.blkRowStart[nb] = total;
		S->c.blkMaxRows = mx;
	}
	// The walk over the bodies of the large islands (home bodies per block, adoptions made permanent, slots): by this one
	// workgroup with LDS counters up to CENSUS_WG_MAX_BODIES bodies (10 011-box pyramid: 6 us less than the grid-wide form),
	// beyond that by every workgroup of k_color_check (counts, adoptions) and k_color_fill (slots) with global counters
	// (100 000-box Tumbler: 0.15 ms less than this workgroup walking 45 000 bodies).
	if (S->c.nLBodies > CENSUS_WG_MAX_BODIES)
	{
		const int cnt = t < nb ? W.blkBodyCount[(size_t)t * BLK_SLOT] : 0;
		const int start = blockScan1024(cnt, s_buf, &total, &mx);
		if (t < nb) W.blkBodyStart[t] = start;
		if (t == 0)
		{
			W.blkBodyStart[nb] = total;
			S->c.blkMaxBodies = mx;
		}
	}
	else
	{
		s_cnt[t] = 0;
		__syncthreads();
		const int nLB = S->c.nLBodies;
		// (four bodies per lane and trip: the loads of a trip are issued together - one workgroup walks all bodies of the large
		// islands here, and a trip is a chain of three dependent loads)
		for (int k0 = t; k0 < nLB; k0 += 4 * 1024)
		{
			int body[4], e[4];
	#pragma unroll
			for (int u = 0; u < 4; ++u) body[u] = k0 + u * 1024 < nLB ? W.li_bodies[k0 + u * 1024] : -1;
	#pragma unroll
			for (int u = 0; u < 4; ++u) e[u] = body[u] >= 0 ? effBlk(W, body[u]) : 0;
	#pragma unroll
			for (int u = 0; u < 4; ++u)
			{
				if (e[u] > 0 && e[u] <= nb)
				{
					W.b_blk1[body[u]] = e[u];
					atomicAdd(&s_cnt[e[u] - 1], 1);
				}
				else if (body[u] >= 0 && atomicCAS(&S->dbgCensus[0], 0, body[u] + 1) == 0)
				{
					S->dbgCensus[1] = e[u]; S->dbgCensus[2] = W.b_blk1[body[u]]; S->dbgCensus[3] = W.b_adopt[body[u]];
					const uint32_t hx = (uint32_t)body[u] * 2654435761u >> 8;
					const int nbx = W.st->c.nBlocks;
					S->dbgCensus[4] = (int)hx; S->dbgCensus[5] = nbx; S->dbgCensus[6] = nbx > 0 ? ownIdBlock(body[u], nbx) - 1 : -1; /* (the solver's own function: the plain % was miscompiled here, round 5) */ S->dbgCensus[7] = effBlk(W, body[u]);
				}
			}
		}
		__syncthreads();
		const int cnt = s_cnt[t];
		const int start = blockScan1024(cnt, s_buf, &total, &mx);
		s_start[t] = start;
		if (t < nb) W.blkBodyStart[t] = start;
		if (t == 0)
		{
			W.blkBodyStart[nb] = total;
			S->c.blkMaxBodies = mx;
		}
		s_cnt[t] = 0;
		__syncthreads();
		for (int k0 = t; k0 < nLB; k0 += 4 * 1024)
		{
			int body[4], e[4];
	#pragma unroll
			for (int u = 0; u < 4; ++u) body[u] = k0 + u * 1024 < nLB ? W.li_bodies[k0 + u * 1024] : -1;
	#pragma unroll
			for (int u = 0; u < 4; ++u) e[u] = body[u] >= 0 ? W.b_blk1[body[u]] : 0;
	#pragma unroll
			for (int u = 0; u < 4; ++u)
			{
				if (e[u] > 0 && e[u] <= nb)
				{
					const int slot = atomicAdd(&s_cnt[e[u] - 1], 1);
					W.blkBodies[s_start[e[u] - 1] + slot] = body[u];
					W.b_slot[body[u]] = slot;
				}
			}
		}
	}
	if (t == 0) S->gapClock[0] = wall_clock64();
	// (the colour census k_color_check has just taken goes home with the counters: the host picks k_sweep_end's tail colours from it)
	if (t < MAX_COLORS) S->c.colorRows[t] = t == HUB_COLOR ? 0 : __hip_atomic_load(&W.colorCount[colorSlot(t)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	// the island build ends here and the host is waiting for its census to size the solver launches
	if (pub != nullptr) b2dPublishCensus(W, pub);
}

// ---- partition ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mortonSpread16(uint32_t x)
{
	x &= 0xffffu;
	x = (x | (x << 8)) & 0x00ff00ffu;
	x = (x | (x << 4)) & 0x0f0f0f0fu;
	x = (x | (x << 2)) & 0x33333333u;
	x = (x | (x << 1)) & 0x55555555u;
	return x;
}

// Sort keys of the large-island bodies: (Morton code of the cell of the body's centre << 32) | body id. Bodies outside
// the large islands lose their block (a stale block id would make them a far-away member of it when they come back).
__global__ __launch_bounds__(256) void k_part_keys(DW W, uint64_t* keys, int2* vals)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if ((W.b_flags[i] & BF_LARGE) == 0) W.b_blk1[i] = 0;
	}
	const int n = S->c.nLBodies;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int body = W.li_bodies[k];
		const float4 p = W.b_pos[body];
		int ix = (int)floorf(p.x * W.invCellSize) + 32768, iy = (int)floorf(p.y * W.invCellSize) + 32768;
		ix = ix < 0 ? 0 : (ix > 65535 ? 65535 : ix);
		iy = iy < 0 ? 0 : (iy > 65535 ? 65535 : iy);
		const uint32_t m = mortonSpread16((uint32_t)ix) | (mortonSpread16((uint32_t)iy) << 1);
		keys[k] = ((uint64_t)m << 32) | (uint32_t)body;
		const int dg = W.deg[body];
		vals[k] = make_int2(body, dg > 0 ? dg : 1);
	}
}

__global__ __launch_bounds__(256) void k_part_weights(DW W, const int2* vals, int* weights)
{
	b2dPhaseStamp(W);
	const int n = W.st->c.nLBodies;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) weights[k] = vals[k].y;
}

// Bodies in sorted order, `prefix` = exclusive scan of their degrees: a block is closed every blkTargetDeg.
__global__ __launch_bounds__(256) void k_part_assign(DW W, const int2* vals, const int* prefix)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nLBodies;
	const int target = S->c.blkTargetDeg > 0 ? S->c.blkTargetDeg : BLOCK_TARGET_DEG;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		int blk = prefix[k] / target;
		if (blk >= MAX_BLOCKS) blk = MAX_BLOCKS - 1; // (over-full last block: the host keeps such a world off k_solve_blocks)
		W.b_blk1[vals[k].x] = blk + 1;
		if (k == n - 1)
		{
			S->c.nBlocks = blk + 1;
			S->c.partitions += 1;
			S->c.partitionAge = 0;
		}
	}
	if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) S->c.nBlocks = 0;
}

// Before k_color_check runs a second time in one step (after a new partition): what k_island_init had prepared for it.
__global__ __launch_bounds__(256) void k_color_recheck_begin(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		W.bodyColorMask[i] = 0;
		W.bodyClaim[i] = 0;
		W.b_adopt[i] = 0;
	}
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MAX_BLOCKS + 1; i += gridDim.x * blockDim.x)
	{
		W.blkRows[(size_t)i * BLK_SLOT] = 0;
		W.blkCursor[(size_t)i * BLK_SLOT] = 0;
		W.blkBodyCount[(size_t)i * BLK_SLOT] = 0;
		W.blkBodyCursor[(size_t)i * BLK_SLOT] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x <= MAX_COLORS)
	{
		W.colorCount[colorSlot(threadIdx.x)] = 0;
		W.colorCursor[colorSlot(threadIdx.x)] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		S->c.nCompact = 0;
		S->c.needRecolor = 0;
		S->c.nColors = 0;
		S->c.nUncolored = 0;
		S->c.nUncolList = 0;
		S->c.nOrphanRows = 0;
		S->c.nSerialOrphans = 0;
		S->c.blkMaxRows = 0;
		S->c.blkMaxBodies = 0;
		S->c.nCutRows = 0;
		S->c.colorMaskLo = S->c.colorMaskHi = 0u;
	}
}

// ---- the solver --------------------------------------------------------------------------------------------------------------------
// Integrated velocity of a body that is not in this workgroup's LDS (the other body of an upper-range constraint): the same
// arithmetic on the same inputs as its home lane performs, so the same bits.
__device__ __forceinline__ void integratedVelocityOf(const DW& W, const StepParams& sp, int body, V2* v, float* w)
{
	const float4 vel = W.b_vel[body];
	*v = v2(vel.x, vel.y);
	*w = vel.z;
	if ((W.b_flags[body] & BF_TYPE_MASK) == BT_DYNAMIC)
	{
		const float4 m = W.b_mass[body], damp = W.b_damp[body], force = W.b_force[body];
		b2dIntegrateVelocity(v, w, sp.dt, sp.gravity, damp.z, m.x, m.y, v2(force.x, force.y), force.z, damp.x, damp.y);
	}
}

// Grid barrier of k_solve_blocks; with one workgroup (a mid-size island of its own) it is a workgroup barrier behind the
// drain of this wave's memory operations (no-return atomics included).
__device__ __forceinline__ bool blockBarrier(const GridBarrier& gb)
{
	if (gb.nWG > 1) return gridBarrier(gb);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	return true;
}

template <int LANES>
__global__ __launch_bounds__(LANES) void k_solve_blocks(DW W, StepParams sp, int* bar, int epoch)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int tid = (int)threadIdx.x, blk = (int)blockIdx.x;
	GridBarrier gb;
	gb.bar = bar;
	gb.overflow = &S->c.overflow;
	gb.nWG = (int)gridDim.x;
	const int gtid = blk * LANES + tid, gsize = (int)gridDim.x * LANES;
	const int nIslands = S->c.nLIslands;
	__shared__ float4 s_row[BLOCK_MAX_BODIES];
	__shared__ int s_perm[LANES];
	__shared__ int s_hist[MAX_COLORS], s_colStart[MAX_COLORS + 1];
	__shared__ unsigned long long s_colMask;
	if (gtid == 0) { S->c.allLargeDone = 0; S->gapClock[3] = wall_clock64(); }
	// The colours come from k_color_small, which the host did not wait for: if it ran out of rounds or of colours (round 6:
	// neither ends the step any more) this kernel must not sweep - a constraint without a colour is in no colour step, and a
	// neighbour block would wait for its hand-over until the spin limit. Every workgroup reads the same two words and leaves;
	// the host sees bit 6, puts the state back, finishes the colouring and solves launch by launch (b2hip_host_phases.h).
	if (S->c.nUncolored != 0 || (S->c.overflow & 4) != 0)
	{
		if (gtid == 0) { stcI(&bar[4], 1); atomicOr(gb.overflow, 64); }
		return;
	}
	// (penetration maxima: every slot was wiped by k_island_init)
	const unsigned long long t0 = wall_clock64();
#define BLK_STAMP(k) do { if (gtid == 0) bar[8 + (k)] = (int)(wall_clock64() - t0); } while (0)
	const int tagV = ((2 * epoch + 1) & 0x7fff) << 16, tagP = ((2 * epoch + 2) & 0x7fff) << 16;

	const int rowStart = W.blkRowStart[blk];
	int nR = W.blkRowStart[blk + 1] - rowStart;
	const int bodyStart = W.blkBodyStart[blk];
	int nB = W.blkBodyStart[blk + 1] - bodyStart;
	if (nR > LANES || nB > BLOCK_MAX_BODIES || nB > LANES)
	{
		// the host checked the census before choosing this kernel: cannot happen, fail loudly
		if (tid == 0) { stcI(&bar[4], 1); atomicOr(gb.overflow, 64 | 0x2000); } // (0x2000: a block holds more than a workgroup takes - not a wait that timed out)
		nR = nR > LANES ? LANES : nR;
		nB = 0;
	}

	// ---- my rows, sorted by colour in LDS (k_color_fill grouped them by block only) ------------------------------------------
	if (tid < MAX_COLORS) s_hist[tid] = 0;
	__syncthreads();
	int color0 = 0;
	if (tid < nR)
	{
		color0 = W.rowColor[rowStart + tid] & (MAX_COLORS - 1);
		atomicAdd(&s_hist[color0], 1);
	}
	__syncthreads();
	if (tid == 0)
	{
		int run = 0;
		unsigned long long mask = 0ull;
		for (int c = 0; c < MAX_COLORS; ++c)
		{
			s_colStart[c] = run;
			if (s_hist[c]) mask |= 1ull << c;
			run += s_hist[c];
			s_hist[c] = 0;
		}
		s_colStart[MAX_COLORS] = run;
		s_colMask = mask;
	}
	__syncthreads();
	if (tid < nR) s_perm[s_colStart[color0] + atomicAdd(&s_hist[color0], 1)] = tid;
	__syncthreads();
	const bool have = tid < nR;
	const int row = have ? rowStart + s_perm[tid] : 0;
	const int myColor = have ? (W.rowColor[row] & (MAX_COLORS - 1)) : -1;
	const bool upper = have && myColor >= CUT_COLOR_BASE;  // handed over through memory
	const bool inner = have && myColor < CUT_COLOR_BASE;   // both bodies in this workgroup's LDS
	const unsigned long long colMask = s_colMask;

	// ---- my home body: integrate its velocity into LDS (b2Island.cpp:192-230) -----------------------------------------------------
	const bool isBody = tid < nB;
	int hBody = 0, hRoot = 0, hCutDeg = 0;
	float hSleepTime = 0.0f;
	if (isBody)
	{
		hBody = W.blkBodies[bodyStart + tid];
		hRoot = W.parent[hBody];
		hCutDeg = __popcll(W.bodyActive[hBody]);
		const float4 pos = W.b_pos[hBody];
		hSleepTime = pos.w;
		W.b_pos0[hBody] = make_float4(pos.x, pos.y, pos.z, 0.0f);
		V2 v;
		float w;
		integratedVelocityOf(W, sp, hBody, &v, &w);
		s_row[tid] = make_float4(v.x, v.y, w, 0.0f);
	}
	__syncthreads();

	// ---- my constraint (b2ContactSolver::b2ContactSolver + InitializeVelocityConstraints, b2ContactSolver.cpp:47-251) --------------
	LargeRef r;
	r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
	ContactConstraint cc;
	float4 oldImp = make_float4(0, 0, 0, 0);
	int slotA = 0, slotB = 0;
	int degA = 0, rankA = 0, degB = 0, rankB = 0; // upper-range rows: the body's upper-range constraints and my place among them
	float4 statA = make_float4(0, 0, 0, 0), statB = statA; // position of a static body (constant)
	if (have)
	{
		r = largeRef(W, C, row);
		const int4 ids = C.ids[r.ci];
		const float4 pa = W.b_pos[r.bodyA], pb = W.b_pos[r.bodyB];
		statA = pa;
		statB = pb;
		BodyPos pA, pB;
		BodyVel vA, vB;
		pA.c = v2(pa.x, pa.y); pA.a = pa.z;
		pB.c = v2(pb.x, pb.y); pB.a = pb.z;
		vA.v = v2(0, 0); vA.w = 0.0f;
		vB = vA;
		if (inner)
		{
			if (r.nsA) { slotA = W.b_slot[r.bodyA]; const float4 q = s_row[slotA]; vA.v = v2(q.x, q.y); vA.w = q.z; }
			if (r.nsB) { slotB = W.b_slot[r.bodyB]; const float4 q = s_row[slotB]; vB.v = v2(q.x, q.y); vB.w = q.z; }
		}
		else
		{
			const unsigned long long below = (1ull << myColor) - 1ull;
			if (r.nsA)
			{
				integratedVelocityOf(W, sp, r.bodyA, &vA.v, &vA.w);
				const unsigned long long m = W.bodyActive[r.bodyA];
				degA = __popcll(m);
				rankA = __popcll(m & below);
			}
			if (r.nsB)
			{
				integratedVelocityOf(W, sp, r.bodyB, &vB.v, &vB.w);
				const unsigned long long m = W.bodyActive[r.bodyB];
				degB = __popcll(m);
				rankB = __popcll(m & below);
			}
		}
		const float4 mA4 = W.b_mass[r.bodyA], mB4 = W.b_mass[r.bodyB];
		const float4 cmat = C.mat[r.ci];
		const float4 m0 = C.man0[r.ci], m1 = C.man1[r.ci];
		oldImp = C.imp[r.ci];
		const int4 m3 = C.man3[r.ci];
		Manifold mf;
		mf.localNormal = v2(m0.x, m0.y);
		mf.localPoint = v2(m0.z, m0.w);
		mf.p[0] = v2(m1.x, m1.y);
		mf.p[1] = v2(m1.z, m1.w);
		mf.ni[0] = oldImp.x; mf.ti[0] = oldImp.y; mf.ni[1] = oldImp.z; mf.ti[1] = oldImp.w;
		mf.id[0] = (uint32_t)m3.x; mf.id[1] = (uint32_t)m3.y;
		mf.type = m3.z;
		mf.pointCount = m3.w;
		b2dInitConstraint(&cc, &mf, cmat.x, cmat.y, cmat.z,
			mA4.x, mA4.y, v2(mA4.z, mA4.w), W.shapes[W.p_shape[ids.x]].radius,
			mB4.x, mB4.y, v2(mB4.z, mB4.w), W.shapes[W.p_shape[ids.y]].radius,
			pA, vA, pB, vB, sp.warmStarting != 0, sp.dtRatio);
	}
	const bool nsA = have && r.nsA, nsB = have && r.nsB;
	const bool hBoundary = isBody && hCutDeg > 0;
	float4* const cutRowA = nsA ? &W.b_cutv[r.bodyA] : nullptr;
	float4* const cutRowB = nsB ? &W.b_cutv[r.bodyB] : nullptr;
	float4* const posRowA = nsA ? &W.b_posv[r.bodyA] : nullptr;
	float4* const posRowB = nsB ? &W.b_posv[r.bodyB] : nullptr;
	// does anything of this block cross a block boundary (an upper-range row of its own, or a home body that another block's
	// row touches)? A block in the middle of nowhere - a mid-size island of its own - never goes to memory in a sweep.
	const bool crossing = __syncthreads_or(hBoundary || upper) != 0;
	BLK_STAMP(0);

	// ---- warm start + velocity iterations (b2ContactSolver.cpp:253-603) --------------------------------------------------------------
	const int sweeps = (sp.warmStarting ? 1 : 0) + sp.velIters;
	for (int sweep = 0; sweep < sweeps; ++sweep)
	{
		const bool warm = sp.warmStarting && sweep == 0;
		// interior colours: LDS rows, one workgroup barrier per colour that occurs in this block
		for (unsigned long long m = colMask & COLOR_INTERIOR_BITS; m != 0ull; m &= m - 1ull)
		{
			const int c = __ffsll((long long)m) - 1;
			if (myColor == c)
			{
				BodyVel vA, vB;
				vA.v = v2(0, 0); vA.w = 0.0f;
				vB = vA;
				if (nsA) { const float4 q = s_row[slotA]; vA.v = v2(q.x, q.y); vA.w = q.z; }
				if (nsB) { const float4 q = s_row[slotB]; vB.v = v2(q.x, q.y); vB.w = q.z; }
				if (warm) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
				if (nsA) s_row[slotA] = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
				if (nsB) s_row[slotB] = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
			}
			__syncthreads();
		}
		if (!crossing) continue;
		// boundary bodies out ...
		if (hBoundary)
		{
			const float4 q = s_row[tid];
			stRow(&W.b_cutv[hBody], q.x, q.y, q.z, tagV + sweep * (hCutDeg + 1) + 1);
		}
		// ... through their upper-range constraints ...
		{
			const int needA = tagV + sweep * (degA + 1) + 1 + rankA, needB = tagV + sweep * (degB + 1) + 1 + rankB;
			const bool ok = dataflowRun(upper, cutRowA, needA, cutRowB, needB, bar, gb.overflow, 1, [&](f4v ra, f4v rb)
			{
				BodyVel vA, vB;
				vA.v = v2(ra.x, ra.y); vA.w = ra.z;
				vB.v = v2(rb.x, rb.y); vB.w = rb.z;
				if (!nsA) { vA.v = v2(0, 0); vA.w = 0.0f; }
				if (!nsB) { vB.v = v2(0, 0); vB.w = 0.0f; }
				if (warm) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
				if (nsA) stRow(cutRowA, vA.v.x, vA.v.y, vA.w, needA + 1);
				if (nsB) stRow(cutRowB, vB.v.x, vB.v.y, vB.w, needB + 1);
			});
			if (!ok) return;
		}
		// ... and back in
		{
			const int need = tagV + (sweep + 1) * (hCutDeg + 1);
			const bool ok = dataflowRun(hBoundary, &W.b_cutv[hBody], need, nullptr, 0, bar, gb.overflow, 1, [&](f4v ra, f4v)
			{
				s_row[tid] = make_float4(ra.x, ra.y, ra.z, 0.0f);
			});
			if (!ok) return;
		}
		__syncthreads();
	}
	BLK_STAMP(1);

	// ---- StoreImpulses (b2ContactSolver.cpp:605-618) ---------------------------------------------------------------------------------------
	if (have)
	{
		float4 im = oldImp;
		if (cc.pointCount > 0) { im.x = cc.normalImpulse[0]; im.y = cc.tangentImpulse[0]; }
		if (cc.pointCount > 1) { im.z = cc.normalImpulse[1]; im.w = cc.tangentImpulse[1]; }
		C.imp[r.ci] = im;
		if (W.postSolveOn && cc.pointCount < cc.pcPointCount) C.flags[r.ci] |= CF_VC_ONE_POINT; // PostSolve reports the solver's point count
	}

	// ---- integrate positions (b2Island.cpp:283-313): the LDS rows become position rows -------------------------------------------------
	if (isBody)
	{
		const float4 p = W.b_pos[hBody];
		const float4 q = s_row[tid];
		V2 c = v2(p.x, p.y), vv = v2(q.x, q.y);
		float a = p.z, w = q.z;
		b2dIntegratePosition(&c, &a, &vv, &w, sp.dt);
		W.b_vel[hBody] = make_float4(vv.x, vv.y, w, 0.0f);
		s_row[tid] = make_float4(c.x, c.y, a, 0.0f);
	}
	__syncthreads();
	BLK_STAMP(2);

	// ---- position iterations (b2Island.cpp:316-335, b2ContactSolver.cpp:676-752) -----------------------------------------------------------
	// An island is closed once an iteration leaves its minimum separation >= -3 linearSlop (b2Island.cpp:329-334): a verdict
	// over ALL rows of the island, i.e. a grid barrier per iteration - which made a position iteration three times as long as
	// a velocity sweep (the sweeps overlap between workgroups, an iteration behind a barrier pays the whole chain of
	// hand-offs plus the slowest workgroup). So the verdict is taken ONE ITERATION LATE: iteration it + 1 starts without
	// waiting for it, on every row not yet known to be closed; when the verdict of iteration `it` arrives (by then it almost
	// always has: no stall), the home bodies of islands it closed are put back to the state iteration `it` left - nothing
	// else has seen what the surplus iteration did to them (islands share no moving bodies, and a closed island's rows
	// stay out of every later iteration). The result is the state the barrier-per-iteration form computes, bit for bit.
	//   pen slots : the penetration maxima of iteration `it` go to slot it % ROOT_PEN_SLOTS of rootPen (all wiped by
	//               k_island_init); a slot is wiped for re-use after wait(it + 2), before arrive(it + 3)
	//   barriers  : arrive(it) after the rows of iteration `it`; wait(it - 1) right after it. bar[0] counts arrivals.
	__shared__ float4 s_back[BLOCK_MAX_BODIES];
	__shared__ int s_waitOk;
	auto arrive = [&]()
	{
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		// (relaxed: everything other workgroups read of this one - penetration maxima, flags, hand-over rows - was written with
		// atomics or sc1 stores, complete by the s_waitcnt above; a release here would write the whole L2 back)
		if (gb.nWG > 1 && tid == 0) __hip_atomic_fetch_add(&bar[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	};
	auto waitFor = [&](int g) -> bool
	{
		if (gb.nWG <= 1) return true;
		if (tid == 0)
		{
			int ok = 1, spins = 0;
			const int need = (g + 1) * gb.nWG;
			while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) // (what follows reads past the L2: ldc*)
			{
				if (++spins > (bar[6] != 0 ? bar[6] : PERSIST_SPIN_MAX) || ldcI(&bar[4]) != 0)
				{
					stcI(&bar[4], 1);
					atomicOr(gb.overflow, 64);
					ok = 0;
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			s_waitOk = ok;
		}
		__syncthreads();
		return s_waitOk != 0;
	};
	const float closedAt = -3.0f * B2D_LINEAR_SLOP;
	bool rowDone = false, bodyDone = false;
	int itDone = sp.posIters; // iterations run
	for (int it = 0; it < sp.posIters; ++it)
	{
		uint32_t* pen = W.rootPen + (size_t)(it % ROOT_PEN_SLOTS) * W.nBodies;
		const bool active = have && !rowDone;
		const bool hActive = hBoundary && !bodyDone;
		float minSep = 0.0f;
		for (unsigned long long m = colMask & COLOR_INTERIOR_BITS; m != 0ull; m &= m - 1ull)
		{
			const int c = __ffsll((long long)m) - 1;
			if (myColor == c && active)
			{
				BodyPos pA, pB;
				float4 qa = statA, qb = statB;
				if (nsA) qa = s_row[slotA];
				if (nsB) qb = s_row[slotB];
				pA.c = v2(qa.x, qa.y); pA.a = qa.z;
				pB.c = v2(qb.x, qb.y); pB.a = qb.z;
				b2dSolvePosition<true>(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
				if (nsA) s_row[slotA] = make_float4(pA.c.x, pA.c.y, pA.a, 0.0f);
				if (nsB) s_row[slotB] = make_float4(pB.c.x, pB.c.y, pB.a, 0.0f);
			}
			__syncthreads();
		}
		if (crossing)
		{
			if (hActive)
			{
				const float4 q = s_row[tid];
				stRow(&W.b_posv[hBody], q.x, q.y, q.z, tagP + it * (hCutDeg + 1) + 1);
			}
			{
				const int needA = tagP + it * (degA + 1) + 1 + rankA, needB = tagP + it * (degB + 1) + 1 + rankB;
				const bool ok = dataflowRun(upper && active, posRowA, needA, posRowB, needB, bar, gb.overflow, 1, [&](f4v ra, f4v rb)
				{
					BodyPos pA, pB;
					pA.c = v2(ra.x, ra.y); pA.a = ra.z;
					pB.c = v2(rb.x, rb.y); pB.a = rb.z;
					if (!nsA) { pA.c = v2(statA.x, statA.y); pA.a = statA.z; }
					if (!nsB) { pB.c = v2(statB.x, statB.y); pB.a = statB.z; }
					b2dSolvePosition<true>(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
					if (nsA) stRow(posRowA, pA.c.x, pA.c.y, pA.a, needA + 1);
					if (nsB) stRow(posRowB, pB.c.x, pB.c.y, pB.a, needB + 1);
				});
				if (!ok) return;
			}
			{
				const int need = tagP + (it + 1) * (hCutDeg + 1);
				const bool ok = dataflowRun(hActive, &W.b_posv[hBody], need, nullptr, 0, bar, gb.overflow, 1, [&](f4v ra, f4v)
				{
					s_row[tid] = make_float4(ra.x, ra.y, ra.z, 0.0f);
				});
				if (!ok) return;
			}
		}
		waveAtomicMaxU32(pen, r.root, floatBits(0.0f - minSep), active);
		arrive();
		if (it > 0)
		{
			// ---- the verdict of iteration it - 1 ----------------------------------------------------------------------------------
			if (!waitFor(it - 1)) return;
			const uint32_t* penPrev = W.rootPen + (size_t)((it - 1) % ROOT_PEN_SLOTS) * W.nBodies;
			if (have && !rowDone) rowDone = -__uint_as_float(ldcU(&penPrev[r.root])) >= closedAt;
			if (isBody && !bodyDone && -__uint_as_float(ldcU(&penPrev[hRoot])) >= closedAt)
			{
				bodyDone = true;
				s_row[tid] = s_back[tid]; // what iteration it - 1 left: iteration `it` should not have touched this island
			}
			// one lane per island: the flag k_large_sleep reads, the census of open islands, the iteration count
			int open = 0;
			for (int k = gtid; k < nIslands; k += gsize)
			{
				const int root = W.li_roots[k];
				if (ldcI(&W.rootDone[root]) != 0) continue;
				if (-__uint_as_float(ldcU(&penPrev[root])) >= closedAt)
				{
					stcI(&W.rootDone[root], 1);
					atomicMax(&S->c.posItersLarge, it); // iterations 0 .. it - 1 ran on it
				}
				else
				{
					++open;
				}
			}
			if (open) __hip_atomic_fetch_add(&bar[16 + ((it - 1) & 7)], open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// slots for re-use (long iteration counts): everybody has read slot it - 3 before arriving at it - 1 ...
			if (it >= 3)
			{
				uint32_t* wipe = W.rootPen + (size_t)((it - 3) % ROOT_PEN_SLOTS) * W.nBodies;
				for (int k = gtid; k < nIslands; k += gsize) stcU(&wipe[W.li_roots[k]], 0u);
			}
			if (gtid == 0 && it >= 5) stcI(&bar[16 + ((it - 5) & 7)], 0);
			// ... and the census of iteration it - 3 is complete: nothing open then = nothing to do since
			if (it >= 3 && ldcI(&bar[16 + ((it - 3) & 7)]) == 0)
			{
				itDone = it + 1;
				break;
			}
		}
		// what this iteration left, for the islands the next verdict closes
		if (isBody && !bodyDone) s_back[tid] = s_row[tid];
		__syncthreads();
	}
	if (itDone == sp.posIters && sp.posIters > 0)
	{
		// the verdict of the last iteration (nothing ran after it: flags only)
		const int it = sp.posIters;
		if (!waitFor(it - 1)) return;
		const uint32_t* penPrev = W.rootPen + (size_t)((it - 1) % ROOT_PEN_SLOTS) * W.nBodies;
		for (int k = gtid; k < nIslands; k += gsize)
		{
			const int root = W.li_roots[k];
			if (ldcI(&W.rootDone[root]) != 0) continue;
			if (-__uint_as_float(ldcU(&penPrev[root])) >= closedAt) stcI(&W.rootDone[root], 1);
			atomicMax(&S->c.posItersLarge, it);
		}
	}
	BLK_STAMP(3);

	// ---- positions back into the body table (sleepTime in the 4th word is untouched) ------------------------------------------------------------
	if (isBody)
	{
		const float4 q = s_row[tid];
		W.b_pos[hBody] = make_float4(q.x, q.y, q.z, hSleepTime);
	}
	BLK_STAMP(4);
#undef BLK_STAMP
}

// ---- one sweep per launch ------------------------------------------------------------------------------------------------------
// Islands with joints or hub bodies cannot stay inside one resident kernel: between the contact sweeps the island's joints are
// walked in order (k_large_joints) and the hub constraints are swept by k_large_hub. Instead of one launch per COLOUR
// (k_large_velocity / k_large_position: 17 colours x 12 sweeps = 200 launches of ~5 us for the 100 000-box Tumbler, each
// bound by the launch itself) this kernel does one SWEEP per launch with the block machinery of k_solve_blocks: a workgroup
// per block, its bodies' rows in LDS for the interior colours, the upper-range (cut) constraints handed over through
// tagged rows in memory. The constraints live in the field-major rows k_large_init wrote (W.lc), the bodies in b_vel /
// b_pos between launches, exactly as for the launch-per-colour kernels - whose result it reproduces bit for bit (same
// colours, same order on every body: interior colours ascending, then upper colours ascending, then the hub sweep).
//   mode 0 warm start, 1 velocity iteration, 2 position iteration
template <int LANES>
__global__ __launch_bounds__(LANES) void k_blocks_sweep(DW W, StepParams sp, int mode, int* bar, int epoch)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (mode == 2 && S->c.allLargeDone) return;
	const ContactArrays& C = W.ca[S->cur];
	const int tid = (int)threadIdx.x, blk = (int)blockIdx.x;
	__shared__ float4 s_row[BLOCK_MAX_BODIES];
	__shared__ int s_perm[LANES];
	__shared__ int s_hist[MAX_COLORS], s_colStart[MAX_COLORS + 1];
	__shared__ unsigned long long s_colMask;
	const int tag = (epoch & 0x7fff) << 16;
	const int rowStart = W.blkRowStart[blk];
	int nR = W.blkRowStart[blk + 1] - rowStart;
	const int bodyStart = W.blkBodyStart[blk];
	int nB = W.blkBodyStart[blk + 1] - bodyStart;
	if (nR > LANES || nB > BLOCK_MAX_BODIES || nB > LANES)
	{
		if (tid == 0) { stcI(&bar[4], 1); atomicOr(&S->c.overflow, 64 | 0x2000); }
		nR = nR > LANES ? LANES : nR;
		nB = 0;
	}
	// ---- my rows, sorted by colour in LDS -------------------------------------------------------------------------------------
	if (tid < MAX_COLORS) s_hist[tid] = 0;
	__syncthreads();
	int color0 = 0;
	if (tid < nR)
	{
		color0 = W.rowColor[rowStart + tid] & (MAX_COLORS - 1);
		atomicAdd(&s_hist[color0], 1);
	}
	__syncthreads();
	if (tid == 0)
	{
		int run = 0;
		unsigned long long mask = 0ull;
		for (int c = 0; c < MAX_COLORS; ++c)
		{
			s_colStart[c] = run;
			if (s_hist[c]) mask |= 1ull << c;
			run += s_hist[c];
			s_hist[c] = 0;
		}
		s_colStart[MAX_COLORS] = run;
		s_colMask = mask;
	}
	__syncthreads();
	if (tid < nR) s_perm[s_colStart[color0] + atomicAdd(&s_hist[color0], 1)] = tid;
	__syncthreads();
	const bool have = tid < nR;
	const int row = have ? rowStart + s_perm[tid] : 0;
	const int myColor = have ? (W.rowColor[row] & (MAX_COLORS - 1)) : -1;
	const bool upper = have && myColor >= CUT_COLOR_BASE && myColor != HUB_COLOR;
	const unsigned long long colMask = s_colMask;
	float4* const rows = mode == 2 ? W.b_pos : W.b_vel;
	float4* const xch = mode == 2 ? W.b_posv : W.b_cutv;

	// ---- my home body into LDS ---------------------------------------------------------------------------------------------------
	const bool isBody = tid < nB;
	int hBody = 0, hCutDeg = 0;
	float hW = 0.0f;
	if (isBody)
	{
		hBody = W.blkBodies[bodyStart + tid];
		hCutDeg = __popcll(W.bodyActive[hBody]);
		const float4 q = rows[hBody];
		hW = q.w;
		s_row[tid] = make_float4(q.x, q.y, q.z, 0.0f);
	}
	// ---- my constraint -----------------------------------------------------------------------------------------------------------
	LargeRef r;
	r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
	ContactConstraint cc;
	memset(&cc, 0, sizeof(cc));
	int slotA = 0, slotB = 0, degA = 0, rankA = 0, degB = 0, rankB = 0;
	float4 statA = make_float4(0, 0, 0, 0), statB = statA;
	bool active = have && myColor != HUB_COLOR;
	if (active)
	{
		r = largeRef(W, C, row);
		if (mode == 2)
		{
			active = W.rootDone[r.root] == 0;
			lcLoad(W, row, cc, LC_MASS_FIRST, LC_MASS_FIRST + 4);
			lcLoad(W, row, cc, LC_POS_FIRST, LC_WORDS);
			statA = W.b_pos[r.bodyA];
			statB = W.b_pos[r.bodyB];
		}
		else
		{
			lcLoad(W, row, cc, 0, LC_VEL_WORDS);
		}
		if (!upper)
		{
			if (r.nsA) slotA = W.b_slot[r.bodyA];
			if (r.nsB) slotB = W.b_slot[r.bodyB];
		}
		else
		{
			const unsigned long long below = (1ull << myColor) - 1ull;
			if (r.nsA) { const unsigned long long m = W.bodyActive[r.bodyA]; degA = __popcll(m); rankA = __popcll(m & below); }
			if (r.nsB) { const unsigned long long m = W.bodyActive[r.bodyB]; degB = __popcll(m); rankB = __popcll(m & below); }
		}
	}
	const bool nsA = have && r.nsA, nsB = have && r.nsB;
	// (a closed island's bodies take no part: its rows are inactive everywhere, so nobody waits for them either)
	const bool hBoundary = isBody && hCutDeg > 0 && !(mode == 2 && W.rootDone[W.parent[hBody]] != 0);
	float4* const xA = nsA ? &xch[r.bodyA] : nullptr;
	float4* const xB = nsB ? &xch[r.bodyB] : nullptr;
	const bool crossing = __syncthreads_or(hBoundary || (upper && active)) != 0;
	float minSep = 0.0f;

	// ---- interior colours ----------------------------------------------------------------------------------------------------------
	for (unsigned long long m = colMask & COLOR_INTERIOR_BITS; m != 0ull; m &= m - 1ull)
	{
		const int c = __ffsll((long long)m) - 1;
		if (myColor == c && active)
		{
			if (mode == 2)
			{
				float4 qa = statA, qb = statB;
				if (nsA) qa = s_row[slotA];
				if (nsB) qb = s_row[slotB];
				BodyPos pA, pB;
				pA.c = v2(qa.x, qa.y); pA.a = qa.z;
				pB.c = v2(qb.x, qb.y); pB.a = qb.z;
				b2dSolvePosition<true>(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
				if (nsA) s_row[slotA] = make_float4(pA.c.x, pA.c.y, pA.a, 0.0f);
				if (nsB) s_row[slotB] = make_float4(pB.c.x, pB.c.y, pB.a, 0.0f);
			}
			else
			{
				BodyVel vA, vB;
				vA.v = v2(0, 0); vA.w = 0.0f;
				vB = vA;
				if (nsA) { const float4 q = s_row[slotA]; vA.v = v2(q.x, q.y); vA.w = q.z; }
				if (nsB) { const float4 q = s_row[slotB]; vB.v = v2(q.x, q.y); vB.w = q.z; }
				if (mode == 0) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
				if (nsA) s_row[slotA] = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
				if (nsB) s_row[slotB] = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
			}
		}
		__syncthreads();
	}
	// ---- upper colours through memory ------------------------------------------------------------------------------------------------
	if (crossing)
	{
		if (hBoundary)
		{
			const float4 q = s_row[tid];
			stRow(&xch[hBody], q.x, q.y, q.z, tag + 1);
		}
		{
			const int needA = tag + 1 + rankA, needB = tag + 1 + rankB;
			bool ok;
			if (mode == 2)
			{
				ok = dataflowRun(upper && active, xA, needA, xB, needB, bar, &S->c.overflow, 1, [&](f4v ra, f4v rb)
				{
					BodyPos pA, pB;
					pA.c = v2(ra.x, ra.y); pA.a = ra.z;
					pB.c = v2(rb.x, rb.y); pB.a = rb.z;
					if (!nsA) { pA.c = v2(statA.x, statA.y); pA.a = statA.z; }
					if (!nsB) { pB.c = v2(statB.x, statB.y); pB.a = statB.z; }
					b2dSolvePosition<true>(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
					if (nsA) stRow(xA, pA.c.x, pA.c.y, pA.a, needA + 1);
					if (nsB) stRow(xB, pB.c.x, pB.c.y, pB.a, needB + 1);
				});
			}
			else
			{
				ok = dataflowRun(upper && active, xA, needA, xB, needB, bar, &S->c.overflow, 1, [&](f4v ra, f4v rb)
				{
					BodyVel vA, vB;
					vA.v = v2(ra.x, ra.y); vA.w = ra.z;
					vB.v = v2(rb.x, rb.y); vB.w = rb.z;
					if (!nsA) { vA.v = v2(0, 0); vA.w = 0.0f; }
					if (!nsB) { vB.v = v2(0, 0); vB.w = 0.0f; }
					if (mode == 0) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
					if (nsA) stRow(xA, vA.v.x, vA.v.y, vA.w, needA + 1);
					if (nsB) stRow(xB, vB.v.x, vB.v.y, vB.w, needB + 1);
				});
			}
			if (!ok) return;
		}
		{
			const int need = tag + (hCutDeg + 1);
			const bool ok = dataflowRun(hBoundary, &xch[hBody], need, nullptr, 0, bar, &S->c.overflow, 1, [&](f4v ra, f4v)
			{
				s_row[tid] = make_float4(ra.x, ra.y, ra.z, 0.0f);
			});
			if (!ok) return;
		}
	}
	(void)degA; (void)degB;
	// ---- out ---------------------------------------------------------------------------------------------------------------------------
	if (mode == 1 && active) lcStore(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
	if (mode == 2) waveAtomicMaxU32(W.rootPen, r.root, floatBits(0.0f - minSep), active);
	if (isBody)
	{
		const float4 q = s_row[tid];
		rows[hBody] = make_float4(q.x, q.y, q.z, mode == 2 ? hW : 0.0f);
	}
}

#endif
