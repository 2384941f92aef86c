// b2d_handover.h - what one workgroup hands to another inside ONE launch: agent-scope loads and stores that go past the
// XCD's L2 (sc1), 16-byte rows that carry their version in the fourth word, the polling loop that waits for two such rows
// (dataflowRun), and a grid barrier with a bounded spin. Used by the block solvers (b2d_kernels_solve_blocks.h); the three
// round-1 resident solvers these helpers were written for live in box2d-mt_amd/validation_src/ (test build only).
//
// A dependent hand-over between workgroups costs about what a kernel boundary costs on this part (4-7 us for a grid barrier,
// 1-3 us for a tagged row on idle CUs, 8 us under load: DESIGN.md sections 3 and 4): the solvers are built to need few of them.
// Every spin is bounded: a stuck wave raises Counters::overflow bit 6 and all workgroups leave.
#ifndef B2D_HANDOVER_H
#define B2D_HANDOVER_H

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
			const int spinMax = gb.bar[6] != 0 ? gb.bar[6] : PERSIST_SPIN_MAX; // (bar[6]: tests force the time-out, k_step_begin)
			while (ldcI(&gb.bar[1]) == gen)
			{
				if (++spins > spinMax || ldcI(&gb.bar[4]) != 0)
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
	// (s_nop 1: a store of more than 8 bytes reads its data registers late, and a VALU write to them within the next two
	// instructions can reach the store first. The compiler's hazard pass keeps that distance for its own stores; an
	// instruction inside an asm statement is invisible to it. k_solve_blocks once had `global_store_dwordx4 v[.], v[4:7]` /
	// `s_or_b64 exec` / `v_mad_u64_u32 v[4:5]`: a hand-over row with words of the next address computation in it, now and
	// then - the bench scene stopped repeating bit for bit (round 5; tools/asm_store_hazard.py looks for the pattern).)
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
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
	int lastA = 0, lastB = 0; // (the versions the last poll saw: pollSleep >= 4)
	const int spinMax = bar[6] != 0 ? bar[6] : DATAFLOW_SPIN_MAX; // (bar[6]: tests force the time-out, k_step_begin)
	while (__any(pending))
	{
		if (pending)
		{
			f4v ra = { 0.0f, 0.0f, 0.0f, 0.0f }, rb = ra;
			if (rowA && rowB) ldRow2(rowA, rowB, &ra, &rb);
			else if (rowA) ra = ldRow(rowA);
			else if (rowB) rb = ldRow(rowB);
			lastA = __float_as_int(ra.w); lastB = __float_as_int(rb.w);
			const bool ready = (!rowA || __float_as_int(ra.w) == needA) && (!rowB || __float_as_int(rb.w) == needB);
			if (ready)
			{
				body(ra, rb);
				pending = false;
			}
		}
		// wave-uniform bookkeeping: every lane counts every trip
		++spins;
		if (spins > spinMax || ((spins & 1023) == 0 && __any(ldcI(&bar[4]) != 0)))
		{
			// (post mortem, B2HIP_HANDOVER_WHY=1: the first lane that gave up on its OWN count leaves what it waited for - the rows'
			// addresses in 16-byte units, the versions it needed and the ones it last saw - in bar[24..31])
			if (pending && spins > spinMax && atomicCAS(&bar[24], 0, 1) == 0)
			{
				f4v ra = { 0.0f, 0.0f, 0.0f, 0.0f }, rb = ra;
				if (rowA) ra = ldRow(rowA);
				if (rowB) rb = ldRow(rowB);
				bar[25] = needA; bar[26] = __float_as_int(ra.w);
				bar[27] = needB; bar[28] = __float_as_int(rb.w);
				bar[29] = rowA ? (int)(uint32_t)((uintptr_t)rowA >> 4) : -1;
				bar[30] = rowB ? (int)(uint32_t)((uintptr_t)rowB >> 4) : -1;
				bar[31] = (int)blockIdx.x * 4096 + (int)threadIdx.x;
			}
			stcI(&bar[4], 1);
			atomicOr(overflow, 64);
			return false;
		}
		if (__any(pending))
		{
			if (pollSleep == 1) __builtin_amdgcn_s_sleep(1);
			else if (pollSleep == 2) __builtin_amdgcn_s_sleep(4);
			else if (pollSleep == 3) __builtin_amdgcn_s_sleep(12);
			else if (pollSleep >= 4)
			{
				// back off by DISTANCE (round 6): the versions count the hand-overs of a body within this launch, so what a lane has
				// just read says how many are still to come before its turn; the wave sleeps for the nearest of its lanes' turns,
				// ~(pollSleep - 3) x 0.43 us per hand-over still ahead of it (a hop is ~2.5 us): fewer polls past the L2 per hop
				int dist = 64;
				if (pending)
				{
					const int da = rowA ? needA - lastA : 0, db = rowB ? needB - lastB : 0;
					int d = da > db ? da : db;
					if (d < 1 || d > 63) d = 1; // (another launch's tag: nothing known)
					dist = d;
				}
				for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(dist, off); dist = o < dist ? o : dist; }
				const int naps = (dist - 1) * (pollSleep - 3);
				for (int k = 0; k < naps && k < 64; ++k) __builtin_amdgcn_s_sleep(16);
				__builtin_amdgcn_s_sleep(1);
			}
		}
	}
	return true;
}

#endif
