// b2d_kernels_solve_persist.h - the coloured large-island solver as ONE persistent kernel.
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
#ifndef B2D_KERNELS_SOLVE_PERSIST_H
#define B2D_KERNELS_SOLVE_PERSIST_H

#include "b2d_kernels_solve_large.h"

#define PERSIST_LANES 256
#define PERSIST_SPIN_MAX (1 << 22)

// bar[0] arrivals (monotonic: barrier g is complete when it reaches (g + 1) * nWG), bar[1] generation,
// bar[2..3] open-island counters (alternating), bar[4] abort. Zeroed by the host before every launch.
struct GridBarrier
{
	int* bar;
	int* overflow;
	int nWG;
};

__device__ __forceinline__ int ldcI(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stcI(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ldcU(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stcU(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte rows shared between workgroups: two 8-byte agent-scope accesses (a row is never read while it is written:
// the phases are separated by grid barriers)
__device__ __forceinline__ float4 ldc4(const float4* p)
{
	const unsigned long long* q = (const unsigned long long*)p;
	const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	float4 r;
	r.x = __uint_as_float((uint32_t)a);
	r.y = __uint_as_float((uint32_t)(a >> 32));
	r.z = __uint_as_float((uint32_t)b);
	r.w = __uint_as_float((uint32_t)(b >> 32));
	return r;
}

__device__ __forceinline__ void stc4(float4* p, float4 v)
{
	unsigned long long* q = (unsigned long long*)p;
	const unsigned long long a = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
	const unsigned long long b = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
	__hip_atomic_store(q, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returns false if the barrier was abandoned (some workgroup never arrived).
__device__ __forceinline__ bool gridBarrier(const GridBarrier& gb)
{
	__shared__ int s_ok;
	// every storing wave drains its sc1 stores, THEN the workgroup barrier, THEN one lane signals for all of them
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0)
	{
		int ok = 1;
		const int gen = ldcI(&gb.bar[1]);
		const int prev = __hip_atomic_fetch_add(&gb.bar[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (prev + 1 == (gen + 1) * gb.nWG)
		{
			__hip_atomic_fetch_add(&gb.bar[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		else
		{
			int spins = 0;
			while (ldcI(&gb.bar[1]) == gen)
			{
				if (++spins > PERSIST_SPIN_MAX || ldcI(&gb.bar[4]) != 0)
				{
					stcI(&gb.bar[4], 1);
					atomicOr(gb.overflow, 64);
					ok = 0;
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
		}
		if (ldcI(&gb.bar[4]) != 0) ok = 0;
		s_ok = ok;
	}
	__syncthreads();
	return s_ok != 0;
}

#ifdef B2HIP_VALIDATION_SOLVERS // (cross-check solver: see b2hip.hip)
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

#endif // B2HIP_VALIDATION_SOLVERS

#endif
