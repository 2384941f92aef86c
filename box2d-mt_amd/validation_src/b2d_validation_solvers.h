// b2d_validation_solvers.h - the three resident large-island solvers of round 1 (grid barrier per colour; polled body rows;
// pushed mailboxes), superseded by k_solve_blocks. TEST BUILD ONLY (make -C box2d-mt_amd validation, -DB2HIP_VALIDATION_SOLVERS):
// tests/test_gpu_parity.py::test_block_solver_matches_launch_per_colour cross-checks the block solver against them bit for bit.
// The product library does not contain them.
#ifndef B2D_VALIDATION_SOLVERS_H
#define B2D_VALIDATION_SOLVERS_H

#include "../csrc/b2d_handover.h"

// ---- (1) one persistent kernel, a grid barrier per colour ------------------------------------------------------------------------
// (round 1, first form) the coloured large-island solver as ONE persistent kernel.
//
// The multi-launch version (b2d_kernels_solve_large.h) pays a kernel boundary (~7 us measured, of which the
// algorithmic work of a colour is < 1 us on a 10^4-body island) for every colour of every sweep: ~135 dependent
// launches per step. Here one grid stays resident for the whole b2Island::Solve of all large islands:
//   * one constraint per lane, held in REGISTERS from b2ContactSolver's constructor to StoreImpulses and through the
//     position iterations (the row never travels again);
//   * body velocities / positions are the only data shared between workgroups. They live in HBM/L2 and are accessed with
//     agent-scope (sc1) loads and stores, so they are coherent across the 8 XCD L2s without any cache flush;
//   * a colour boundary is a grid barrier: one agent-scope atomic per workgroup + a generation flag (bounded spin:
//     a lost workgroup raises Counters::overflow bit 6 and every workgroup leaves, instead of hanging the GPU).
// Arithmetic and visiting order are exactly those of the multi-launch path (same colours, same sweep structure), so the
// results are bit-identical to it (tests/test_gpu_parity.py::test_persistent_solver_matches_launch_per_colour).
__global__ __launch_bounds__(PERSIST_LANES) void k_solve_persistent(DW W, StepParams sp, int nColorsArg, int* bar)
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

	// ---- integrate velocities (b2Island.cpp:192-230) ------------------------------------------------------------
	for (int k = gtid; k < nBodies; k += gsize)
	{
		const int body = W.li_bodies[k];
		const float4 pos = W.b_pos[body];
		W.b_pos0[body] = make_float4(pos.x, pos.y, pos.z, 0.0f);
		const uint32_t f = W.b_flags[body];
		if ((f & BF_TYPE_MASK) == BT_DYNAMIC)
		{
			const float4 vel = W.b_vel[body];
			const float4 m = W.b_mass[body], damp = W.b_damp[body], force = W.b_force[body];
			V2 v = v2(vel.x, vel.y);
			float w = vel.z;
			b2dIntegrateVelocity(&v, &w, sp.dt, sp.gravity, damp.z, m.x, m.y, v2(force.x, force.y), force.z, damp.x, damp.y);
			stc4(&W.b_vel[body], make_float4(v.x, v.y, w, 0.0f));
		}
	}
	if (!gridBarrier(gb)) return;

	// ---- my constraint: row = gtid (rows are sorted by colour) -------------------------------------------------------
	const bool have = gtid < nRows;
	LargeRef r;
	r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
	int myColor = -1;
	ContactConstraint cc;
	float4 oldImp = make_float4(0, 0, 0, 0);
	if (have)
	{
		r = largeRef(W, C, gtid);
		for (int c = 0; c < nColors; ++c)
		{
			if (gtid >= s_colorStart[c] && gtid < s_colorStart[c + 1]) myColor = c;
		}
		const int4 ids = C.ids[r.ci];
		const float4 pa = W.b_pos[r.bodyA], pb = W.b_pos[r.bodyB];
		const float4 va = r.nsA ? ldc4(&W.b_vel[r.bodyA]) : make_float4(0, 0, 0, 0);
		const float4 vb = r.nsB ? ldc4(&W.b_vel[r.bodyB]) : make_float4(0, 0, 0, 0);
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
	if (!gridBarrier(gb)) return; // every constructor has read the pre-warm-start velocities

	// ---- warm start + velocity iterations, colour by colour ------------------------------------------------------------
	const int sweeps = (sp.warmStarting ? 1 : 0) + sp.velIters;
	for (int sweep = 0; sweep < sweeps; ++sweep)
	{
		const bool warm = sp.warmStarting && sweep == 0;
		for (int c = 0; c < nColors; ++c)
		{
			if (myColor == c)
			{
				BodyVel vA, vB;
				vA.v = v2(0, 0); vA.w = 0; vB = vA;
				if (r.nsA) { const float4 v = ldc4(&W.b_vel[r.bodyA]); vA.v = v2(v.x, v.y); vA.w = v.z; }
				if (r.nsB) { const float4 v = ldc4(&W.b_vel[r.bodyB]); vB.v = v2(v.x, v.y); vB.w = v.z; }
				if (warm) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
				if (r.nsA) stc4(&W.b_vel[r.bodyA], make_float4(vA.v.x, vA.v.y, vA.w, 0.0f));
				if (r.nsB) stc4(&W.b_vel[r.bodyB], make_float4(vB.v.x, vB.v.y, vB.w, 0.0f));
			}
			if (!gridBarrier(gb)) return;
		}
	}

	// ---- StoreImpulses (b2ContactSolver.cpp:605-618) ----------------------------------------------------------------------
	if (have)
	{
		float4 im = oldImp;
		if (cc.pointCount > 0) { im.x = cc.normalImpulse[0]; im.y = cc.tangentImpulse[0]; }
		if (cc.pointCount > 1) { im.z = cc.normalImpulse[1]; im.w = cc.tangentImpulse[1]; }
		C.imp[r.ci] = im;
		if (W.postSolveOn && cc.pointCount < cc.pcPointCount) C.flags[r.ci] |= CF_VC_ONE_POINT; // PostSolve reports the solver's point count
	}

	// ---- integrate positions (b2Island.cpp:283-313) ---------------------------------------------------------------------------
	for (int k = gtid; k < nBodies; k += gsize)
	{
		const int body = W.li_bodies[k];
		const float4 p = W.b_pos[body], v = ldc4(&W.b_vel[body]);
		V2 c = v2(p.x, p.y), vv = v2(v.x, v.y);
		float a = p.z, w = v.z;
		b2dIntegratePosition(&c, &a, &vv, &w, sp.dt);
		stc4(&W.b_pos[body], make_float4(c.x, c.y, a, p.w));
		stc4(&W.b_vel[body], make_float4(vv.x, vv.y, w, 0.0f));
	}
	if (gtid == 0) stcI(&gb.bar[2], 0);
	if (!gridBarrier(gb)) return;

	// ---- position iterations with per-island early out (b2Island.cpp:316-335) -----------------------------------------------------
	for (int it = 0; it < sp.posIters; ++it)
	{
		int* openNow = &gb.bar[2 + (it & 1)];
		int* openNext = &gb.bar[2 + ((it + 1) & 1)];
		for (int k = gtid; k < nIslands; k += gsize) stcU(&W.rootPen[W.li_roots[k]], 0u);
		if (!gridBarrier(gb)) return;
		if (gtid == 0) stcI(openNext, 0); // nobody reads this counter before the barrier that ends this iteration
		for (int c = 0; c < nColors; ++c)
		{
			// wave-uniform call of the aggregated atomic: every lane takes part, lanes without work pass valid = false
			bool valid = myColor == c && ldcI(&W.rootDone[r.root]) == 0;
			float minSep = 0.0f;
			if (valid)
			{
				const float4 pa = ldc4(&W.b_pos[r.bodyA]), pb = ldc4(&W.b_pos[r.bodyB]);
				BodyPos pA, pB;
				pA.c = v2(pa.x, pa.y); pA.a = pa.z;
				pB.c = v2(pb.x, pb.y); pB.a = pb.z;
				b2dSolvePosition(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
				if (r.nsA) stc4(&W.b_pos[r.bodyA], make_float4(pA.c.x, pA.c.y, pA.a, pa.w));
				if (r.nsB) stc4(&W.b_pos[r.bodyB], make_float4(pB.c.x, pB.c.y, pB.a, pb.w));
			}
			waveAtomicMaxU32(W.rootPen, r.root, floatBits(0.0f - minSep), valid);
			if (!gridBarrier(gb)) return;
		}
		int open = 0;
		for (int k = gtid; k < nIslands; k += gsize)
		{
			const int root = W.li_roots[k];
			if (ldcI(&W.rootDone[root])) continue;
			const float minSeparation = -__uint_as_float(ldcU(&W.rootPen[root]));
			if (minSeparation >= -3.0f * B2D_LINEAR_SLOP) stcI(&W.rootDone[root], 1); else ++open;
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
}


// ---- (2) versioned body rows instead of colour-wide barriers ---------------------------------------------------------------------
// (round 1, second form) the coloured large-island solver without colour-wide barriers.
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


// ---- (3) pushed mailboxes ----------------------------------------------------------------------------------------------------------
// (round 1, third form) the dataflow large-island solver with pushed hand-offs.
//
// k_solve_dataflow (above) synchronises constraints through versioned body rows: every lane
// polls the rows of its two bodies, 2 x 64 scattered 16-byte lines per wave and poll, all of them served by the
// fabric (an sc1 store drops the line from L2). On MI355X the price of a hand-off sits in the consumer CU's memory
// queue (MI355X_MICROARCH.md, "handoff-1to1": 0.8 us idle, 2.3-3.5 us on loaded CUs), and those polls are the load:
// a hop measured 3.3-3.8 us on the 10k-body pyramid.
//
// Here the producer pushes instead. The constraints of a body are totally ordered by colour, so each constraint knows
// its successor on either body (cyclically: the last one hands over to the first one of the next sweep). Every lane owns
// a mailbox of two 16-byte slots, one per body, laid out by lane: (x, y, angle-or-w, tag). After solving, a lane stores
// each updated body row into the successor's slot for that body; a wave then polls 64 x 32 contiguous bytes = 16 lines
// instead of 128. The tag is (phase epoch << 16) + update count of the body, so a slot never needs clearing and a
// stale row of an earlier phase or step cannot match.
//
// Update order per body, arithmetic and barriers are those of k_solve_dataflow: results are bit-identical to it, to
// k_solve_persistent and to the launch-per-colour path (tests/test_gpu_parity.py).
#define DF_RANKS 32 // successor table entries per body; bodies with more constraints are hubs (HUB_DEGREE) and never get here

// Both slots of a lane with one wait (they are adjacent: offset 16).
__device__ __forceinline__ void ldMailbox(const float4* box, f4v* a, f4v* b)
{
	f4v r, s;
	asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
		: "=&v"(r), "=&v"(s) : "v"(box) : "memory");
	*a = r;
	*b = s;
}

// Mailbox store. LOCAL: every workgroup of the launch sits on one XCD (see k_solve_mailbox), so a plain store - it goes
// through the write-through vector L1 into that XCD's L2 and stays there - is visible to the consumer's L1-bypassing
// load; otherwise the sc1 (write-through to memory) form that is coherent across XCDs.
template <bool LOCAL>
__device__ __forceinline__ void stBox(float4* p, float x, float y, float z, int tag)
{
	f4v v;
	v.x = x;
	v.y = y;
	v.z = z;
	v.w = __int_as_float(tag);
	if (LOCAL) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
	else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// Wait until the awaited slots carry their tags, then run `body(ra, rb)`. A side that is not awaited (static body, or
// the first update of a body in a phase, which reads the body table) is ignored. Returns false if the wait was abandoned.
template <typename F>
__device__ __forceinline__ bool mailboxRun(bool pending, const float4* box, bool waitA, int tagA, bool waitB, int tagB, int* bar, int* overflow, F body)
{
	int spins = 0;
	while (__any(pending))
	{
		if (pending)
		{
			f4v ra = { 0.0f, 0.0f, 0.0f, 0.0f }, rb = ra;
			if (waitA || waitB) ldMailbox(box, &ra, &rb);
			const bool ready = (!waitA || __float_as_int(ra.w) == tagA) && (!waitB || __float_as_int(rb.w) == tagB);
			if (ready)
			{
				body(ra, rb);
				pending = false;
			}
		}
		++spins;
		if (spins > DATAFLOW_SPIN_MAX || ((spins & 1023) == 0 && __any(ldcI(&bar[4]) != 0)))
		{
			stcI(&bar[4], 1);
			atomicOr(overflow, 64);
			return false;
		}
		if (__any(pending)) __builtin_amdgcn_s_sleep(1);
	}
	return true;
}
// LOCAL = true: the single-XCD form. The per-XCD L2s are not coherent with each other, which is why a cross-XCD hand-off
// has to go through memory (~3.3 us a hop here). If all workgroups share one XCD the hand-offs stay in its L2. HIP
// promises nothing about placement, so the launch asks for 8 x nWG workgroups (they are dealt round-robin over the 8 XCDs),
// every workgroup reads its HW_REG_XCC_ID, the first one to arrive names the target XCD, workgroups elsewhere leave at
// once and the first nWG on the target take the work. If fewer than nWG show up within 30 us (a different dealing order, a
// partitioned device) nothing has been touched yet: bar[7] becomes 2, everybody leaves, and the ordinary launch that
// follows (LOCAL = false, skipIfDone = 1) does the step. If they do show up, bar[7] = 1 and that launch returns at once.
template <bool LOCAL>
__global__ __launch_bounds__(PERSIST_LANES) void k_solve_mailbox(DW W, StepParams sp, int nColorsArg, int* bar, int epoch, int nWGArg, int skipIfDone)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	__shared__ int s_wg;
	if (LOCAL)
	{
		if (threadIdx.x == 0)
		{
			int xcc = 0;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			xcc = (xcc & 0xf) + 1;
			int target = atomicCAS(&bar[5], 0, xcc);
			if (target == 0) target = xcc;
			int slot = -1;
			if (target == xcc)
			{
				slot = atomicAdd(&bar[6], 1);
				if (slot >= nWGArg) slot = -1;
			}
			if (slot >= 0)
			{
				if (slot == nWGArg - 1) atomicCAS(&bar[7], 0, 1);
				const unsigned long long tStart = wall_clock64();
				while (ldcI(&bar[7]) == 0)
				{
					if (wall_clock64() - tStart > 3000ull) atomicCAS(&bar[7], 0, 2);
					__builtin_amdgcn_s_sleep(2);
				}
				if (ldcI(&bar[7]) != 1) slot = -1;
			}
			s_wg = slot;
		}
		__syncthreads();
		if (s_wg < 0) return;
	}
	else
	{
		if (skipIfDone && ldcI(&bar[7]) == 1) return;
		if (threadIdx.x == 0) s_wg = (int)blockIdx.x;
		__syncthreads();
	}
	const int wgIndex = s_wg, wgCount = LOCAL ? nWGArg : (int)gridDim.x;
	const int nColors = nColorsArg >= 0 ? nColorsArg : (S->c.nColors < MAX_COLORS ? S->c.nColors : MAX_COLORS);
	const ContactArrays& C = W.ca[S->cur];
	GridBarrier gb;
	gb.bar = bar;
	gb.overflow = &S->c.overflow;
	gb.nWG = wgCount;
	const int gtid = wgIndex * blockDim.x + threadIdx.x;
	const int gsize = wgCount * blockDim.x;
	const int nRows = S->c.nLContacts, nBodies = S->c.nLBodies, nIslands = S->c.nLIslands;
	__shared__ int s_colorStart[MAX_COLORS + 2];
	if ((int)threadIdx.x <= nColors && threadIdx.x <= MAX_COLORS) s_colorStart[threadIdx.x] = W.colorStart[threadIdx.x];
	if (gtid == 0) S->c.allLargeDone = 0;
	const unsigned long long t0 = wall_clock64();
#define DF_STAMP(k) do { if (gtid == 0) bar[8 + (k)] = (int)(wall_clock64() - t0); } while (0)
	const int tagV = ((2 * epoch + 1) & 0x7fff) << 16, tagP = ((2 * epoch + 2) & 0x7fff) << 16;

	// ---- integrate velocities (b2Island.cpp:192-230) -----------------------------------------------------------------
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
	f4v seedA = { 0.0f, 0.0f, 0.0f, 0.0f }, seedB = seedA; // the integrated velocities = input of a body's first update
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
		if (r.nsA) seedA = ldRow(&W.b_vel[r.bodyA]);
		if (r.nsB) seedB = ldRow(&W.b_vel[r.bodyB]);
		const float4 mA4 = W.b_mass[r.bodyA], mB4 = W.b_mass[r.bodyB];
		BodyPos pA, pB;
		BodyVel vA, vB;
		pA.c = v2(pa.x, pa.y); pA.a = pa.z;
		pB.c = v2(pb.x, pb.y); pB.a = pb.z;
		vA.v = v2(seedA.x, seedA.y); vA.w = seedA.z;
		vB.v = v2(seedB.x, seedB.y); vB.w = seedB.z;
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
	if (!gridBarrier(gb)) return; // the colour masks are complete
	DF_STAMP(1);

	// ---- my place in the update order of my two bodies, published for the predecessor to find --------------------------------
	int degA = 0, rankA = 0, degB = 0, rankB = 0;
	if (have)
	{
		const uint64_t below = (1ull << myColor) - 1ull;
		if (r.nsA)
		{
			const uint64_t m = __hip_atomic_load((unsigned long long*)&W.bodyActive[r.bodyA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			degA = __popcll(m);
			rankA = __popcll(m & below);
			if (rankA < DF_RANKS) stcI(&W.dfRank[(size_t)r.bodyA * DF_RANKS + rankA], 2 * gtid);
		}
		if (r.nsB)
		{
			const uint64_t m = __hip_atomic_load((unsigned long long*)&W.bodyActive[r.bodyB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			degB = __popcll(m);
			rankB = __popcll(m & below);
			if (rankB < DF_RANKS) stcI(&W.dfRank[(size_t)r.bodyB * DF_RANKS + rankB], 2 * gtid + 1);
		}
		if (degA > DF_RANKS || degB > DF_RANKS)
		{
			// cannot happen while HUB_DEGREE < DF_RANKS (the host keeps hub islands off this kernel): fail loudly
			stcI(&bar[4], 1);
			atomicOr(gb.overflow, 64);
		}
	}
	if (!gridBarrier(gb)) return;
	float4* succA = nullptr; // the slot of my successor on body A (its A or B side, whichever that body is for it)
	float4* succB = nullptr;
	if (have && r.nsA) succA = W.dfInbox + ldcI(&W.dfRank[(size_t)r.bodyA * DF_RANKS + (rankA + 1 == degA ? 0 : rankA + 1)]);
	if (have && r.nsB) succB = W.dfInbox + ldcI(&W.dfRank[(size_t)r.bodyB * DF_RANKS + (rankB + 1 == degB ? 0 : rankB + 1)]);
	const float4* const box = W.dfInbox + 2 * (size_t)gtid;
	const bool nsA = have && r.nsA, nsB = have && r.nsB;

	// ---- warm start + velocity iterations ---------------------------------------------------------------------------------
	const int sweeps = (sp.warmStarting ? 1 : 0) + sp.velIters;
	for (int sweep = 0; sweep < sweeps; ++sweep)
	{
		const bool warm = sp.warmStarting && sweep == 0;
		const bool last = sweep == sweeps - 1;
		const int needA = sweep * degA + rankA, needB = sweep * degB + rankB;
		const bool ok = mailboxRun(have, box, nsA && needA != 0, tagV + needA, nsB && needB != 0, tagV + needB, bar, gb.overflow, [&](f4v ra, f4v rb)
		{
			if (needA == 0) ra = seedA;
			if (needB == 0) rb = seedB;
			BodyVel vA, vB;
			vA.v = v2(ra.x, ra.y); vA.w = ra.z;
			vB.v = v2(rb.x, rb.y); vB.w = rb.z;
			if (warm) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
			if (nsA)
			{
				// the last update of a body in this phase goes to the body table, every other one to the successor
				if (last && rankA + 1 == degA) stRow(&W.b_vel[r.bodyA], vA.v.x, vA.v.y, vA.w, 0);
				else stBox<LOCAL>(succA, vA.v.x, vA.v.y, vA.w, tagV + needA + 1);
			}
			if (nsB)
			{
				if (last && rankB + 1 == degB) stRow(&W.b_vel[r.bodyB], vB.v.x, vB.v.y, vB.w, 0);
				else stBox<LOCAL>(succB, vB.v.x, vB.v.y, vB.w, tagV + needB + 1);
			}
		});
		if (!ok) return;
	}
	if (sweeps == 0)
	{
		// no iterations at all: the integrated velocities are final (they already sit in the body table)
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

	// ---- integrate positions (b2Island.cpp:283-313) -----------------------------------------------------------------------------
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
	// ---- position iterations (b2Island.cpp:316-335): pushed hand-offs inside an iteration, island verdicts between iterations ---------
	int executed = 0; // iterations in which my island was still open
	for (int it = 0; it < sp.posIters; ++it)
	{
		int* openNow = &gb.bar[2 + (it & 1)];
		int* openNext = &gb.bar[2 + ((it + 1) & 1)];
		const bool active = have && ldcI(&W.rootDone[r.root]) == 0;
		const int needA = executed * degA + rankA, needB = executed * degB + rankB;
		float minSep = 0.0f;
		// a body's first update of the phase reads the integrated position; static bodies come from the (read-only) body table
		f4v firstA = { 0.0f, 0.0f, 0.0f, 0.0f }, firstB = firstA;
		if (active && (!r.nsA || needA == 0))
		{
			if (r.nsA) firstA = ldRow(&W.b_posv[r.bodyA]);
			else { const float4 p = W.b_pos[r.bodyA]; firstA.x = p.x; firstA.y = p.y; firstA.z = p.z; }
		}
		if (active && (!r.nsB || needB == 0))
		{
			if (r.nsB) firstB = ldRow(&W.b_posv[r.bodyB]);
			else { const float4 p = W.b_pos[r.bodyB]; firstB.x = p.x; firstB.y = p.y; firstB.z = p.z; }
		}
		const bool ok = mailboxRun(active, box, nsA && needA != 0, tagP + needA, nsB && needB != 0, tagP + needB, bar, gb.overflow, [&](f4v ra, f4v rb)
		{
			if (!nsA || needA == 0) ra = firstA;
			if (!nsB || needB == 0) rb = firstB;
			BodyPos pA, pB;
			pA.c = v2(ra.x, ra.y); pA.a = ra.z;
			pB.c = v2(rb.x, rb.y); pB.a = rb.z;
			b2dSolvePosition(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
			if (nsA)
			{
				// whether another iteration follows is only known after the verdict: the last update of an iteration goes to
				// the successor AND to the body table
				stBox<LOCAL>(succA, pA.c.x, pA.c.y, pA.a, tagP + needA + 1);
				if (rankA + 1 == degA) stRow(&W.b_posv[r.bodyA], pA.c.x, pA.c.y, pA.a, 0);
			}
			if (nsB)
			{
				stBox<LOCAL>(succB, pB.c.x, pB.c.y, pB.a, tagP + needB + 1);
				if (rankB + 1 == degB) stRow(&W.b_posv[r.bodyB], pB.c.x, pB.c.y, pB.a, 0);
			}
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


#endif
