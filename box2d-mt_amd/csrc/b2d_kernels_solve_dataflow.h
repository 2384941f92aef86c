// b2d_kernels_solve_dataflow.h - the coloured large-island solver without colour-wide barriers.
//
// A grid barrier costs 4-7 us on MI355X (MI355X_MICROARCH.md, "barrier-counter" / "barrier-xcd"), about what a kernel
// boundary costs, and a Gauss-Seidel sweep over a coloured island needs one per colour: ~100 of them per step for the
// velocity iterations alone. But a constraint does not depend on "its colour having started": it depends on the previous
// update of its own two bodies. This kernel synchronises exactly that:
//   * every non-static body row carries a version in its 4th word = number of constraint updates applied to the body in
//     the current phase: velocity rows (v.x, v.y, w, version), position rows (c.x, c.y, a, version);
//   * the constraints of one body are totally ordered by colour (a colouring never gives two constraints of one body the
//     same colour), so constraint i knows the two versions it must see: sweep * degree(body) + rank(i on body);
//   * a lane polls its two rows with one 16-byte agent-scope load each (data and version arrive together), solves, and
//     publishes both rows with one 16-byte agent-scope store each, version + 1. A hop is one store -> load hand-off
//     (~1-2 us) instead of a grid-wide barrier, and the critical path of a sweep is the longest dependency chain.
// The per-body update order equals the colour order of the barrier versions, and constraints that run concurrently never
// share a body, so the floats are bit-identical to k_solve_persistent and to the launch-per-colour path (tested).
// Grid barriers remain only where the reference itself has a global step: after velocity integration / constraint set-up,
// around position integration, and around each position iteration's per-island convergence test (b2Island.cpp:329-334).
// Every spin is bounded: a stuck wave raises Counters::overflow bit 6 and all workgroups leave.
#ifndef B2D_KERNELS_SOLVE_DATAFLOW_H
#define B2D_KERNELS_SOLVE_DATAFLOW_H

#include "b2d_kernels_solve_persist.h"

#define DATAFLOW_SPIN_MAX (1 << 20)

typedef float f4v __attribute__((ext_vector_type(4)));

// One 16-byte agent-scope (sc1: L1-bypassing, coherent across the XCD L2s) access per body row. The row is the
// hand-off granule: its last word is the version, written by the same store instruction as the data.
__device__ __forceinline__ f4v ldRow(const float4* p)
{
	f4v r;
	asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
	return r;
}

__device__ __forceinline__ void ldRow2(const float4* p, const float4* q, f4v* a, f4v* b)
{
	f4v r, s;
	asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
		: "=&v"(r), "=&v"(s) : "v"(p), "v"(q) : "memory");
	*a = r;
	*b = s;
}

__device__ __forceinline__ void stRow(float4* p, float x, float y, float z, int version)
{
	f4v v;
	v.x = x;
	v.y = y;
	v.z = z;
	v.w = __int_as_float(version);
	asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void atomicOr64(uint64_t* p, uint64_t v)
{
	atomicOr((unsigned long long*)p, (unsigned long long)v);
}

// One dataflow phase for this lane's constraint: wait until both rows show the expected versions, then run `body(ra, rb)`.
// rowA / rowB are null for static bodies (nothing to wait for, nothing to publish). Returns false if the wait was abandoned.
template <typename F>
__device__ __forceinline__ bool dataflowRun(bool pending, const float4* rowA, int needA, const float4* rowB, int needB, int* bar, int* overflow, int pollSleep, F body)
{
	int spins = 0;
	while (__any(pending))
	{
		if (pending)
		{
			f4v ra = { 0.0f, 0.0f, 0.0f, 0.0f }, rb = ra;
			if (rowA && rowB) ldRow2(rowA, rowB, &ra, &rb);
			else if (rowA) ra = ldRow(rowA);
			else if (rowB) rb = ldRow(rowB);
			const bool ready = (!rowA || __float_as_int(ra.w) == needA) && (!rowB || __float_as_int(rb.w) == needB);
			if (ready)
			{
				body(ra, rb);
				pending = false;
			}
		}
		// wave-uniform bookkeeping: every lane counts every trip
		++spins;
		if (spins > DATAFLOW_SPIN_MAX || ((spins & 1023) == 0 && __any(ldcI(&bar[4]) != 0)))
		{
			stcI(&bar[4], 1);
			atomicOr(overflow, 64);
			return false;
		}
		if (__any(pending))
		{
			if (pollSleep == 1) __builtin_amdgcn_s_sleep(1);
			else if (pollSleep == 2) __builtin_amdgcn_s_sleep(4);
			else if (pollSleep == 3) __builtin_amdgcn_s_sleep(12);
		}
	}
	return true;
}

#ifdef B2HIP_VALIDATION_SOLVERS // (cross-check solver: see b2hip.hip)
__global__ __launch_bounds__(PERSIST_LANES) void k_solve_dataflow(DW W, StepParams sp, int nColorsArg, int* bar, int pollSleep)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int nColors = nColorsArg >= 0 ? nColorsArg : (S->c.nColors < MAX_COLORS ? S->c.nColors : MAX_COLORS);
	const ContactArrays& C = W.ca[S->cur];
	GridBarrier gb;
	gb.bar = bar;
	gb.overflow = &S->c.overflow;
	gb.nWG = (int)gridDim.x;
	const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
	const int gsize = gridDim.x * blockDim.x;
	const int nRows = S->c.nLContacts, nBodies = S->c.nLBodies, nIslands = S->c.nLIslands;
	__shared__ int s_colorStart[MAX_COLORS + 2];
	if ((int)threadIdx.x <= nColors && threadIdx.x <= MAX_COLORS) s_colorStart[threadIdx.x] = W.colorStart[threadIdx.x];
	if (gtid == 0) S->c.allLargeDone = 0;
	// phase timestamps of workgroup 0 (100 MHz ticks since kernel start) in bar[8..15]: a debugging aid read by
	// b2hip_debug_read(11); one scalar store per phase
	const unsigned long long t0 = wall_clock64();
#define DF_STAMP(k) do { if (gtid == 0) bar[8 + (k)] = (int)(wall_clock64() - t0); } while (0)

	// ---- integrate velocities (b2Island.cpp:192-230); velocity rows start at version 0 ---------------------------------
	for (int k = gtid; k < nBodies; k += gsize)
	{
		const int body = W.li_bodies[k];
		const float4 pos = W.b_pos[body];
		W.b_pos0[body] = make_float4(pos.x, pos.y, pos.z, 0.0f);
		__hip_atomic_store((unsigned long long*)&W.bodyActive[body], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t f = W.b_flags[body];
		const float4 vel = W.b_vel[body];
		V2 v = v2(vel.x, vel.y);
		float w = vel.z;
		if ((f & BF_TYPE_MASK) == BT_DYNAMIC)
		{
			const float4 m = W.b_mass[body], damp = W.b_damp[body], force = W.b_force[body];
			b2dIntegrateVelocity(&v, &w, sp.dt, sp.gravity, damp.z, m.x, m.y, v2(force.x, force.y), force.z, damp.x, damp.y);
		}
		stRow(&W.b_vel[body], v.x, v.y, w, 0);
	}
	if (!gridBarrier(gb)) return;
	DF_STAMP(0);

	// ---- my constraint: row = gtid ------------------------------------------------------------------------------------
	const bool have = gtid < nRows;
	LargeRef r;
	r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
	int myColor = 0;
	ContactConstraint cc;
	float4 oldImp = make_float4(0, 0, 0, 0);
	if (have)
	{
		r = largeRef(W, C, gtid);
		for (int c = 0; c < nColors; ++c)
		{
			if (gtid >= s_colorStart[c] && gtid < s_colorStart[c + 1]) myColor = c;
		}
		const uint64_t bit = 1ull << myColor;
		if (r.nsA) atomicOr64(&W.bodyActive[r.bodyA], bit);
		if (r.nsB) atomicOr64(&W.bodyActive[r.bodyB], bit);
		const int4 ids = C.ids[r.ci];
		const float4 pa = W.b_pos[r.bodyA], pb = W.b_pos[r.bodyB];
		f4v va = { 0.0f, 0.0f, 0.0f, 0.0f }, vb = va;
		if (r.nsA) va = ldRow(&W.b_vel[r.bodyA]);
		if (r.nsB) vb = ldRow(&W.b_vel[r.bodyB]);
		const float4 mA4 = W.b_mass[r.bodyA], mB4 = W.b_mass[r.bodyB];
		BodyPos pA, pB;
		BodyVel vA, vB;
		pA.c = v2(pa.x, pa.y); pA.a = pa.z;
		pB.c = v2(pb.x, pb.y); pB.a = pb.z;
		vA.v = v2(va.x, va.y); vA.w = va.z;
		vB.v = v2(vb.x, vb.y); vB.w = vb.z;
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
	if (!gridBarrier(gb)) return; // all constructors have read the pre-warm-start velocities; the colour masks are complete
	DF_STAMP(1);

	// ---- my place in the update order of my two bodies --------------------------------------------------------------------
	int degA = 0, rankA = 0, degB = 0, rankB = 0;
	if (have)
	{
		const uint64_t below = (1ull << myColor) - 1ull;
		if (r.nsA)
		{
			const uint64_t m = __hip_atomic_load((unsigned long long*)&W.bodyActive[r.bodyA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			degA = __popcll(m);
			rankA = __popcll(m & below);
		}
		if (r.nsB)
		{
			const uint64_t m = __hip_atomic_load((unsigned long long*)&W.bodyActive[r.bodyB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			degB = __popcll(m);
			rankB = __popcll(m & below);
		}
	}
	float4* const velA = (have && r.nsA) ? &W.b_vel[r.bodyA] : nullptr;
	float4* const velB = (have && r.nsB) ? &W.b_vel[r.bodyB] : nullptr;
	float4* const posA = (have && r.nsA) ? &W.b_posv[r.bodyA] : nullptr;
	float4* const posB = (have && r.nsB) ? &W.b_posv[r.bodyB] : nullptr;

	// ---- warm start + velocity iterations: body-level dataflow ------------------------------------------------------------
	const int sweeps = (sp.warmStarting ? 1 : 0) + sp.velIters;
	for (int sweep = 0; sweep < sweeps; ++sweep)
	{
		const bool warm = sp.warmStarting && sweep == 0;
		const int needA = sweep * degA + rankA, needB = sweep * degB + rankB;
		const bool ok = dataflowRun(have, velA, needA, velB, needB, bar, gb.overflow, pollSleep, [&](f4v ra, f4v rb)
		{
			BodyVel vA, vB;
			vA.v = v2(ra.x, ra.y); vA.w = ra.z;
			vB.v = v2(rb.x, rb.y); vB.w = rb.z;
			if (warm) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
			if (velA) stRow(velA, vA.v.x, vA.v.y, vA.w, needA + 1);
			if (velB) stRow(velB, vB.v.x, vB.v.y, vB.w, needB + 1);
		});
		if (!ok) return;
	}

	DF_STAMP(2);
	// ---- StoreImpulses (b2ContactSolver.cpp:605-618) ----------------------------------------------------------------------
	if (have)
	{
		float4 im = oldImp;
		if (cc.pointCount > 0) { im.x = cc.normalImpulse[0]; im.y = cc.tangentImpulse[0]; }
		if (cc.pointCount > 1) { im.z = cc.normalImpulse[1]; im.w = cc.tangentImpulse[1]; }
		C.imp[r.ci] = im;
		if (W.postSolveOn && cc.pointCount < cc.pcPointCount) C.flags[r.ci] |= CF_VC_ONE_POINT; // PostSolve reports the solver's point count
	}
	if (!gridBarrier(gb)) return; // every body has its final velocity
	DF_STAMP(3);

	// ---- integrate positions (b2Island.cpp:283-313); position rows start at version 0 -------------------------------------------
	for (int k = gtid; k < nBodies; k += gsize)
	{
		const int body = W.li_bodies[k];
		const float4 p = W.b_pos[body];
		const f4v v = ldRow(&W.b_vel[body]);
		V2 c = v2(p.x, p.y), vv = v2(v.x, v.y);
		float a = p.z, w = v.z;
		b2dIntegratePosition(&c, &a, &vv, &w, sp.dt);
		stRow(&W.b_posv[body], c.x, c.y, a, 0);
		W.b_vel[body] = make_float4(vv.x, vv.y, w, 0.0f);
	}
	for (int k = gtid; k < nIslands; k += gsize) stcU(&W.rootPen[W.li_roots[k]], 0u);
	if (gtid == 0)
	{
		stcI(&gb.bar[2], 0);
		stcI(&gb.bar[3], 0);
	}
	if (!gridBarrier(gb)) return;

	DF_STAMP(4);
	// ---- position iterations (b2Island.cpp:316-335): dataflow inside an iteration, island verdicts between iterations ------------
	int executed = 0; // iterations in which my island was still open = version epochs of my bodies
	for (int it = 0; it < sp.posIters; ++it)
	{
		int* openNow = &gb.bar[2 + (it & 1)];
		int* openNext = &gb.bar[2 + ((it + 1) & 1)];
		const bool active = have && ldcI(&W.rootDone[r.root]) == 0;
		const int needA = executed * degA + rankA, needB = executed * degB + rankB;
		float minSep = 0.0f;
		// static bodies are not versioned: their position comes from the (read-only) body table
		BodyPos sA, sB;
		sA.c = v2(0, 0); sA.a = 0; sB = sA;
		if (active && !r.nsA) { const float4 p = W.b_pos[r.bodyA]; sA.c = v2(p.x, p.y); sA.a = p.z; }
		if (active && !r.nsB) { const float4 p = W.b_pos[r.bodyB]; sB.c = v2(p.x, p.y); sB.a = p.z; }
		const bool ok = dataflowRun(active, posA, needA, posB, needB, bar, gb.overflow, pollSleep, [&](f4v ra, f4v rb)
		{
			BodyPos pA, pB;
			if (posA) { pA.c = v2(ra.x, ra.y); pA.a = ra.z; } else pA = sA;
			if (posB) { pB.c = v2(rb.x, rb.y); pB.a = rb.z; } else pB = sB;
			b2dSolvePosition(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
			if (posA) stRow(posA, pA.c.x, pA.c.y, pA.a, needA + 1);
			if (posB) stRow(posB, pB.c.x, pB.c.y, pB.a, needB + 1);
		});
		if (!ok) return;
		if (active) ++executed;
		waveAtomicMaxU32(W.rootPen, r.root, floatBits(0.0f - minSep), active);
		if (!gridBarrier(gb)) return;
		if (gtid == 0) stcI(openNext, 0);
		int open = 0;
		for (int k = gtid; k < nIslands; k += gsize)
		{
			const int root = W.li_roots[k];
			if (ldcI(&W.rootDone[root])) continue;
			const float minSeparation = -__uint_as_float(ldcU(&W.rootPen[root]));
			if (minSeparation >= -3.0f * B2D_LINEAR_SLOP) stcI(&W.rootDone[root], 1); else ++open;
			stcU(&W.rootPen[root], 0u);
		}
		if (open) __hip_atomic_fetch_add(openNow, open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (!gridBarrier(gb)) return;
		if (gtid == 0) S->c.posItersLarge += 1;
		if (ldcI(openNow) == 0)
		{
			if (gtid == 0) S->c.allLargeDone = 1;
			break;
		}
	}

	DF_STAMP(5);
	// ---- positions back into the body table (sleepTime in the 4th word is untouched) -----------------------------------------------------
	for (int k = gtid; k < nBodies; k += gsize)
	{
		const int body = W.li_bodies[k];
		const f4v p = ldRow(&W.b_posv[body]);
		const float sleepTime = W.b_pos[body].w;
		W.b_pos[body] = make_float4(p.x, p.y, p.z, sleepTime);
	}
	DF_STAMP(6);
#undef DF_STAMP
}

#endif // B2HIP_VALIDATION_SOLVERS

#endif
