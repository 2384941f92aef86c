// b2d_kernels_solve_small.h - the island solver for SMALL islands (b2Island::Solve, b2Island.cpp:184-396).
//
// One 256-lane workgroup owns a chunk of whole islands (<= 256 bodies, <= 256 constraints in total).
// Lane t owns body t of the chunk during the per-body phases and constraint t during the
// per-constraint phases. Body velocities/positions live in LDS for the whole solve; each constraint
// lives in the registers of its lane. HBM is touched once on the way in and once on the way out.
//
// Bit-exactness: constraints are visited level by level, where the level of a constraint is its depth
// in the dependency DAG of the reference's sequential order (two constraints conflict iff they share a
// non-static body). All constraints of one level are mutually independent, so running them in
// parallel commutes exactly with the sequential sweep; levels are separated by workgroup barriers.
#ifndef B2D_KERNELS_SOLVE_SMALL_H
#define B2D_KERNELS_SOLVE_SMALL_H

#include "b2d_island_joints.h"

__device__ __forceinline__ uint32_t floatBits(float f) { return __float_as_uint(f); }

// JOINTS: the islands may hold joints (at most SMALL_ISLAND_MAX_JOINTS each): lane i of the chunk walks island i's joints in
// the island's joint order (b2d_island_joints.h) on the LDS rows, between the contact sweeps where b2Island::Solve has them.
// Worlds without joints run the lean instantiation.
template <int LANES, bool JOINTS>
__global__ __launch_bounds__(LANES) void k_solve_small(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int chunk = blockIdx.x;
	if (chunk >= S->c.nChunks) return;
	const ContactArrays& C = W.ca[S->cur];
	const int nS = S->c.nSIslands;
	const int i0 = W.chunkFirst[chunk];
	const int i1 = (chunk + 1 < S->c.nChunks) ? W.chunkFirst[chunk + 1] : nS;
	const int bStart = W.si_bodyStart[i0];
	const int nB = W.si_bodyStart[i1] - bStart;
	const int cStart = W.si_contactStart[i0];
	const int nC = W.si_contactStart[i1] - cStart;
	const int nI = i1 - i0;
	const int tid = threadIdx.x;
	const float h = sp.dt;

	__shared__ float4 s_vel[LANES];     // v.xy, w
	__shared__ float4 s_pos[LANES];     // c.xy, a
	__shared__ uint32_t s_pen[LANES];   // per island: bits of max penetration (= -minSeparation)
	__shared__ int s_done[LANES];       // per island: positionSolved
	__shared__ uint32_t s_sleepMin[LANES];
	__shared__ int s_maxLevel;
	__shared__ int s_notDone;

	if (tid == 0) s_maxLevel = 0;
	if (tid < nI)
	{
		s_done[tid] = 0;
		s_sleepMin[tid] = 0x7f7fffffu; // b2_maxFloat
	}
	__syncthreads();
	if (tid < nI) atomicMax(&s_maxLevel, W.si_maxLevel[i0 + tid]);
	int jStart = 0, jCount = 0;
	if (JOINTS && tid < nI)
	{
		const int root = W.si_root[i0 + tid];
		jCount = W.rootJoints[root];
		jStart = W.rootJointStart[root];
	}
	JointBodiesLds jointBodies(W, s_pos, s_vel, bStart, nB);

	// ---- per-body: load, stash c0/a0, integrate velocities (b2Island.cpp:192-230) ---------------
	int body = -1;
	uint32_t bflags = 0;
	float4 massv = make_float4(0, 0, 0, 0);
	float sleepTime = 0.0f;
	int myBodyIsland = 0;
	if (tid < nB)
	{
		body = W.si_bodies[bStart + tid];
		bflags = W.b_flags[body];
		float4 pos = W.b_pos[body];
		float4 vel = W.b_vel[body];
		massv = W.b_mass[body];
		sleepTime = pos.w;
		myBodyIsland = W.b_island[body] - i0;
		W.b_pos0[body] = make_float4(pos.x, pos.y, pos.z, 0.0f);
		V2 v = v2(vel.x, vel.y);
		float w = vel.z;
		if ((bflags & BF_TYPE_MASK) == BT_DYNAMIC)
		{
			float4 damp = W.b_damp[body];
			float4 force = W.b_force[body];
			b2dIntegrateVelocity(&v, &w, h, sp.gravity, damp.z, massv.x, massv.y, v2(force.x, force.y), force.z, damp.x, damp.y);
		}
		s_vel[tid] = make_float4(v.x, v.y, w, 0.0f);
		s_pos[tid] = make_float4(pos.x, pos.y, pos.z, 0.0f);
	}

	// ---- lanes are re-dealt so that constraints of one dependency level sit in consecutive lanes: at level L only the
	// waves that hold level-L constraints execute the solver body, the others branch over it (a stack is a chain: with
	// the discovery order every wave would run every level for a handful of active lanes) ---------------------------
	__shared__ int s_levelStart[LANES];
	__shared__ int s_perm[LANES];
	__shared__ int s_waveSum[LANES / 64];
	s_levelStart[tid] = 0;
	__syncthreads();
	int myLevel0 = 0;
	if (tid < nC)
	{
		// (the clamp only affects which lane a constraint sits in, never the order it is solved in)
		myLevel0 = W.si_level[cStart + tid];
		if (myLevel0 > LANES - 2) myLevel0 = LANES - 2;
		atomicAdd(&s_levelStart[myLevel0 + 1], 1);
	}
	__syncthreads();
	{
		// exclusive scan of the level census, one entry per lane
		const int v = s_levelStart[tid];
		int incl = v;
		for (int off = 1; off < 64; off <<= 1)
		{
			const int n = __shfl_up(incl, off);
			if ((tid & 63) >= off) incl += n;
		}
		if ((tid & 63) == 63) s_waveSum[tid >> 6] = incl;
		__syncthreads();
		int base = 0;
		for (int wv = 0; wv < (tid >> 6); ++wv) base += s_waveSum[wv];
		s_levelStart[tid] = base + incl - v;
	}
	__syncthreads();
	if (tid < nC) s_perm[atomicAdd(&s_levelStart[myLevel0 + 1], 1)] = tid;
	__syncthreads();

	// ---- per-constraint: gather ------------------------------------------------------------------
	ContactConstraint cc;
	int ci = -1, la = -1, lb = -1, level = 0, myIsland = 0;
	BodyPos staticPosA, staticPosB;
	staticPosA.c = v2(0, 0); staticPosA.a = 0;
	staticPosB = staticPosA;
	Manifold mf;
	float4 cmat = make_float4(0, 0, 0, 0);
	float4 mA4 = make_float4(0, 0, 0, 0), mB4 = mA4;
	float radiusA = 0, radiusB = 0;
	if (tid < nC)
	{
		const int mine = s_perm[tid];
		ci = W.si_contacts[cStart + mine];
		level = W.si_level[cStart + mine];
		int4 ids = C.ids[ci];
		const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
		const bool nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
		la = nsA ? W.b_slot[ids.z] - bStart : -1;
		lb = nsB ? W.b_slot[ids.w] - bStart : -1;
		myIsland = W.b_island[nsA ? ids.z : ids.w] - i0;
		if (!nsA)
		{
			float4 p = W.b_pos[ids.z];
			staticPosA.c = v2(p.x, p.y);
			staticPosA.a = p.z;
		}
		if (!nsB)
		{
			float4 p = W.b_pos[ids.w];
			staticPosB.c = v2(p.x, p.y);
			staticPosB.a = p.z;
		}
		mA4 = W.b_mass[ids.z];
		mB4 = W.b_mass[ids.w];
		radiusA = W.shapes[W.p_shape[ids.x]].radius;
		radiusB = W.shapes[W.p_shape[ids.y]].radius;
		cmat = C.mat[ci];
		float4 m0 = C.man0[ci], m1 = C.man1[ci], im = C.imp[ci];
		int4 m3 = C.man3[ci];
		mf.localNormal = v2(m0.x, m0.y);
		mf.localPoint = v2(m0.z, m0.w);
		mf.p[0] = v2(m1.x, m1.y);
		mf.p[1] = v2(m1.z, m1.w);
		mf.ni[0] = im.x; mf.ti[0] = im.y; mf.ni[1] = im.z; mf.ti[1] = im.w;
		mf.id[0] = (uint32_t)m3.x; mf.id[1] = (uint32_t)m3.y;
		mf.type = m3.z;
		mf.pointCount = m3.w;
	}
	__syncthreads();
	const int maxLevel = s_maxLevel;

	// ---- init (b2ContactSolver ctor + InitializeVelocityConstraints): reads pre-warm-start state ----
	if (ci >= 0)
	{
		BodyPos pA, pB;
		BodyVel vA, vB;
		if (la >= 0) { float4 p = s_pos[la], v = s_vel[la]; pA.c = v2(p.x, p.y); pA.a = p.z; vA.v = v2(v.x, v.y); vA.w = v.z; }
		else { pA = staticPosA; vA.v = v2(0, 0); vA.w = 0; }
		if (lb >= 0) { float4 p = s_pos[lb], v = s_vel[lb]; pB.c = v2(p.x, p.y); pB.a = p.z; vB.v = v2(v.x, v.y); vB.w = v.z; }
		else { pB = staticPosB; vB.v = v2(0, 0); vB.w = 0; }
		b2dInitConstraint<true>(&cc, &mf, cmat.x, cmat.y, cmat.z,
			mA4.x, mA4.y, v2(mA4.z, mA4.w), radiusA,
			mB4.x, mB4.y, v2(mB4.z, mB4.w), radiusB,
			pA, vA, pB, vB, sp.warmStarting != 0, sp.dtRatio);
	}
	__syncthreads();

	// ---- warm start + velocity iterations, level by level -----------------------------------------
	const int sweeps = (sp.warmStarting ? 1 : 0) + sp.velIters;
	if (JOINTS && !sp.warmStarting)
	{
		if (jCount > 0) b2dSolveIslandJoints(W, sp, JOINTS_INIT, jStart, jCount, jointBodies);
		__syncthreads();
	}
	for (int sweep = 0; sweep < sweeps; ++sweep)
	{
		const bool warm = sp.warmStarting && sweep == 0;
		if (JOINTS && !warm)
		{
			// joints before contacts in every velocity iteration (b2Island.cpp:268-276)
			if (jCount > 0) b2dSolveIslandJoints(W, sp, JOINTS_VELOCITY, jStart, jCount, jointBodies);
			__syncthreads();
		}
		for (int L = 1; L <= maxLevel; ++L)
		{
			if (ci >= 0 && level == L)
			{
				BodyVel vA, vB;
				if (la >= 0) { float4 v = s_vel[la]; vA.v = v2(v.x, v.y); vA.w = v.z; } else { vA.v = v2(0, 0); vA.w = 0; }
				if (lb >= 0) { float4 v = s_vel[lb]; vB.v = v2(v.x, v.y); vB.w = v.z; } else { vB.v = v2(0, 0); vB.w = 0; }
				if (warm) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
				if (la >= 0) s_vel[la] = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
				if (lb >= 0) s_vel[lb] = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
			}
			__syncthreads();
		}
		if (JOINTS && warm)
		{
			// InitVelocityConstraints of the joints (with their warm start) after the contacts' warm start (:251-259)
			if (jCount > 0) b2dSolveIslandJoints(W, sp, JOINTS_INIT, jStart, jCount, jointBodies);
			__syncthreads();
		}
	}

	// ---- store impulses (b2ContactSolver::StoreImpulses :605-618) ----------------------------------
	if (ci >= 0)
	{
		float4 im = make_float4(mf.ni[0], mf.ti[0], mf.ni[1], mf.ti[1]);
		if (cc.pointCount > 0) { im.x = cc.normalImpulse[0]; im.y = cc.tangentImpulse[0]; }
		if (cc.pointCount > 1) { im.z = cc.normalImpulse[1]; im.w = cc.tangentImpulse[1]; }
		C.imp[ci] = im;
		if (W.postSolveOn && cc.pointCount < cc.pcPointCount) C.flags[ci] |= CF_VC_ONE_POINT; // PostSolve reports the solver's point count
	}

	// ---- integrate positions (b2Island.cpp:283-313) ---------------------------------------------------
	if (body >= 0)
	{
		float4 p = s_pos[tid], v = s_vel[tid];
		V2 c = v2(p.x, p.y), vv = v2(v.x, v.y);
		float a = p.z, w = v.z;
		b2dIntegratePosition(&c, &a, &vv, &w, h);
		s_pos[tid] = make_float4(c.x, c.y, a, 0.0f);
		s_vel[tid] = make_float4(vv.x, vv.y, w, 0.0f);
	}
	__syncthreads();

	// ---- position iterations with per-island early out (b2Island.cpp:316-335) ---------------------
	for (int it = 0; it < sp.posIters; ++it)
	{
		if (tid < nI) s_pen[tid] = 0;
		if (tid == 0) s_notDone = 0;
		__syncthreads();
		for (int L = 1; L <= maxLevel; ++L)
		{
			if (ci >= 0 && level == L && !s_done[myIsland])
			{
				BodyPos pA, pB;
				if (la >= 0) { float4 p = s_pos[la]; pA.c = v2(p.x, p.y); pA.a = p.z; } else pA = staticPosA;
				if (lb >= 0) { float4 p = s_pos[lb]; pB.c = v2(p.x, p.y); pB.a = p.z; } else pB = staticPosB;
				float minSep = 0.0f;
				b2dSolvePosition<true>(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
				if (la >= 0) s_pos[la] = make_float4(pA.c.x, pA.c.y, pA.a, 0.0f);
				if (lb >= 0) s_pos[lb] = make_float4(pB.c.x, pB.c.y, pB.a, 0.0f);
				// minSep <= 0: track max of (0 - minSep) as unsigned bits (monotone for non-negative floats;
				// 0 - (+0) is +0, whereas -(+0) would be -0 = 0x80000000 and win every unsigned max)
				atomicMax(&s_pen[myIsland], floatBits(0.0f - minSep));
			}
			__syncthreads();
		}
		if (tid < nI && !s_done[tid])
		{
			int jointsOkay = 1;
			if (JOINTS && jCount > 0) jointsOkay = b2dSolveIslandJoints(W, sp, JOINTS_POSITION, jStart, jCount, jointBodies);
			float minSeparation = -__uint_as_float(s_pen[tid]);
			if (minSeparation >= -3.0f * B2D_LINEAR_SLOP && jointsOkay)
			{
				s_done[tid] = 1; // contactsOkay && jointsOkay -> positionSolved, break
			}
			else
			{
				atomicAdd(&s_notDone, 1);
			}
		}
		__syncthreads();
		// read the verdict, THEN barrier again: the next iteration's reset of s_notDone must not
		// overtake a slower wave that has not looked at it yet (it would leave the loop alone)
		const int notDone = s_notDone;
		__syncthreads();
		if (notDone == 0) break;
	}
	__syncthreads();

	// ---- write back + SynchronizeTransform (b2Island.cpp:338-349) + sleep (:355-395) -----------------
	V2 vOut = v2(0, 0);
	float wOut = 0.0f;
	float4 pOut = make_float4(0, 0, 0, 0);
	if (body >= 0)
	{
		float4 p = s_pos[tid], v = s_vel[tid];
		pOut = p;
		vOut = v2(v.x, v.y);
		wOut = v.z;
		Xf xf = b2dXfFromSweep(v2(p.x, p.y), p.z, v2(massv.z, massv.w));
		W.b_xf[body] = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
		if (sp.allowSleep)
		{
			const float linTolSqr = B2D_LINEAR_SLEEP_TOL * B2D_LINEAR_SLEEP_TOL;
			const float angTolSqr = B2D_ANGULAR_SLEEP_TOL * B2D_ANGULAR_SLEEP_TOL;
			if ((bflags & BF_AUTOSLEEP) == 0 || wOut * wOut > angTolSqr || b2dDot(vOut, vOut) > linTolSqr)
			{
				sleepTime = 0.0f;
				atomicMin(&s_sleepMin[myBodyIsland], floatBits(0.0f));
			}
			else
			{
				sleepTime += h;
				atomicMin(&s_sleepMin[myBodyIsland], floatBits(sleepTime));
			}
		}
	}
	__syncthreads();
	if (body >= 0)
	{
		uint32_t f = bflags | BF_ISLAND | BF_AWAKE;
		bool sleep = false;
		if (sp.allowSleep)
		{
			float minSleepTime = __uint_as_float(s_sleepMin[myBodyIsland]);
			sleep = minSleepTime >= B2D_TIME_TO_SLEEP && s_done[myBodyIsland];
		}
		if (sleep)
		{
			// b2Body::SetAwake(false) (b2Body.h:704-712)
			f &= ~BF_AWAKE;
			sleepTime = 0.0f;
			vOut = v2(0, 0);
			wOut = 0.0f;
			W.b_force[body] = make_float4(0, 0, 0, 0);
		}
		W.b_flags[body] = f;
		W.b_pos[body] = make_float4(pOut.x, pOut.y, pOut.z, sleepTime);
		W.b_vel[body] = make_float4(vOut.x, vOut.y, wOut, 0.0f);
	}
}

#endif
