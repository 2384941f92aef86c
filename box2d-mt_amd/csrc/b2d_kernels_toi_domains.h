// b2d_kernels_toi_domains.h - b2World::SolveTOI by connected components, the general parallel case.
//
// Everything a TOI event reads or writes is reached through existing contacts: the two seed bodies, the partners in
// their contact lists that join the mini island, the contacts of the island's dynamic bodies whose impacts are
// recomputed, and the partners of those contacts whose sweeps are put on the same interval. All of that lies inside the
// connected component of the seeds in the graph {non-static bodies, contacts between them} (touching or not). Static
// bodies are shared by components but only read: the reference also "advances" them, which changes nothing but their
// alpha0, and a static body is always level with or behind the body it is paired with when an impact is (re)computed
// (that body was advanced to the current event time, and event times never decrease), so a component treats its
// static partners as in sync and never touches their alpha0. Events of different components therefore commute, and the
// reference's global order (alpha, proxy ids) restricted to one component is that component's own order. What does
// couple components is detected and sends the phase back to the serial loop from the snapshot (b2hip_step_end):
//   * an event moves a proxy so that a NEW contact would be created (creation order is global),
//   * a sleeping body is woken (if it rests on a static body, the reference reads that body's alpha0, which events of
//     other components have advanced),
//   * a capacity of the per-component scratch is exceeded,
//   * two proxies moved by different components overlap at any time of the phase without a contact between them
//     (k_toi_domains_end, on the hulls of all the boxes each proxy has had).
// The bullet-versus-bodies fields of config 5 are thousands of tiny components; the serial loop costs ~52 us per event.
#ifndef B2D_KERNELS_TOI_DOMAINS_H
#define B2D_KERNELS_TOI_DOMAINS_H

#include "b2d_kernels_toi_chains.h"

#define TOI_DOMAINS_MAX 65536

__global__ __launch_bounds__(256) void k_toi_dom_init(DW W)
{
	b2dPhaseStamp(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		W.toiParent[i] = i;
		W.toiDomOf[i] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		W.st->c.nToiDomains = 0;
		W.st->c.nToiMoved = 0;
		W.st->c.nToiNewPairs = 0;
		W.st->c.nToiPartial = 0;
		W.st->c.toiAnyFailed = 0;
	}
}

__global__ __launch_bounds__(256) void k_toi_dom_union(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		if ((W.b_flags[ids.z] & BF_TYPE_MASK) == BT_STATIC || (W.b_flags[ids.w] & BF_TYPE_MASK) == BT_STATIC) continue;
		ufUnion(W.toiParent, ids.z, ids.w);
	}
}

__global__ __launch_bounds__(256) void k_toi_dom_flatten(DW W)
{
	b2dPhaseStamp(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		const int r = ufFindReadOnly(W.toiParent, i);
		__hip_atomic_store(&W.toiParent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

__device__ __forceinline__ int toiContactLabel(const DW& W, int4 ids)
{
	const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
	const bool nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
	if (!nsA && !nsB) return -1;
	return W.toiParent[nsA ? ids.z : ids.w];
}

// One component per label that owns a pending impact.
__global__ __launch_bounds__(256) void k_toi_dom_mark(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nToiList < W.capContacts ? S->c.nToiList : W.capContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int label = toiContactLabel(W, C.ids[W.toiList[k]]);
		if (label < 0) continue;
		if (atomicCAS(&W.toiDomOf[label], 0, -1) == 0)
		{
			const int d = atomicAdd(&S->c.nToiDomains, 1);
			if (d < TOI_DOMAINS_MAX)
			{
				W.toiDomRoot[d] = label;
				W.toiDomCount[d] = 0;
				W.toiDomFill[d] = 0;
				W.toiDomFailed[d] = 0;
				W.toiDomEvents[d] = 0;
				atomicExch(&W.toiDomOf[label], d + 1);
			}
			else atomicOr(&S->c.toiUnsafe, TOI_UNSAFE_CAPACITY);
		}
	}
}

// Contacts per component = the capacity of its pending list (a contact is listed at most once).
__global__ __launch_bounds__(256) void k_toi_dom_count(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nToiDomains == 0) return;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int label = toiContactLabel(W, C.ids[i]);
		if (label < 0) continue;
		const int d = W.toiDomOf[label] - 1;
		if (d >= 0) atomicAdd(&W.toiDomCount[d], 1);
	}
}

// Slices of toiDomList (one workgroup: the number of components is small next to the number of contacts).
__global__ __launch_bounds__(1024) void k_toi_dom_scan(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nToiDomains < TOI_DOMAINS_MAX ? S->c.nToiDomains : TOI_DOMAINS_MAX;
	__shared__ int s_wave[16], s_carry;
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (int base = 0; base < n; base += 1024)
	{
		const int i = base + (int)threadIdx.x;
		const int v = i < n ? W.toiDomCount[i] : 0;
		int incl = v;
		for (int off = 1; off < 64; off <<= 1)
		{
			const int o = __shfl_up(incl, off);
			if ((threadIdx.x & 63) >= off) incl += o;
		}
		if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
		__syncthreads();
		int add = s_carry;
		for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) add += s_wave[wv];
		if (i < n) W.toiDomBase[i] = add + incl - v;
		__syncthreads();
		if (threadIdx.x == 1023) s_carry = add + incl;
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void k_toi_dom_fill(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nToiList < W.capContacts ? S->c.nToiList : W.capContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int c = W.toiList[k];
		const int label = toiContactLabel(W, C.ids[c]);
		if (label < 0) continue;
		const int d = W.toiDomOf[label] - 1;
		if (d < 0) continue;
		const int slot = atomicAdd(&W.toiDomFill[d], 1);
		if (slot < W.toiDomCount[d]) W.toiDomList[W.toiDomBase[d] + slot] = c;
	}
}

// The event loop of every component with a pending impact, one workgroup each (a fixed grid takes them in turn).
// LANES = 64: one wave per component. A component of config 5 is a bullet and what it hits - one or two pending impacts, a
// handful of candidate contacts - and an event is ~50 dependent stages with a workgroup barrier between them: with eight waves
// every barrier is a rendezvous, with one the compiler drops it (a workgroup no wider than a wave needs none). The loop takes
// at most LANES candidate contacts per event then (one lane each); a component that has more is a capacity cut - the serial
// loop gets the phase, and the host launches the wide form for a while (b2hip_host_phases.h).
template <int LANES>
__global__ __launch_bounds__(LANES) void k_toi_domains(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.toiUnsafe & TOI_UNSAFE_CAPACITY) return;
	const int n = S->c.nToiDomains < TOI_DOMAINS_MAX ? S->c.nToiDomains : TOI_DOMAINS_MAX;
	if (blockIdx.x == 0 && threadIdx.x == 0 && W.capContacts > 16) W.hubList[12] = n; // (diagnostics: how many rows behind word 16 are this step's)
	for (int d = blockIdx.x; d < n; d += gridDim.x)
	{
		toiLoopRun<true, LANES>(W, sp, d, 0);
		__syncthreads();
	}
}

// Proxies moved by different components must not have overlapped at any time of the phase unless a contact between them
// exists: compared on the hulls of all the fat AABBs each has had (a superset of every momentary overlap).
__global__ __launch_bounds__(256) void k_toi_domains_end(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nToiMoved < TOI_MOVED_MAX ? S->c.nToiMoved : TOI_MOVED_MAX;
	if (S->c.nToiMoved > TOI_MOVED_MAX) atomicOr(&S->c.toiUnsafe, TOI_UNSAFE_CAPACITY);
	if (n < 2 || S->c.toiUnsafe) return;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x; i < n; i += gridDim.x)
	{
		const int p = W.toiMoved[i];
		const int bodyP = W.p_body[p];
		const AABB hp = loadAabb(W.toiHull, p);
		for (int j = i + 1 + (int)threadIdx.x; j < n; j += blockDim.x)
		{
			const int q = W.toiMoved[j];
			const int bodyQ = W.p_body[q];
			if (q == p || bodyQ == bodyP || W.toiParent[bodyQ] == W.toiParent[bodyP]) continue;
			if (!b2dAabbOverlap(hp, loadAabb(W.toiHull, q))) continue;
			// (different components never share a contact, so an overlap here is a contact the serial order would create -
			// unless the pair may not collide at all)
			const int keyP = W.p_key[p], keyQ = W.p_key[q];
			const int lo = keyP < keyQ ? p : q, hi = keyP < keyQ ? q : p;
			if (!bodiesShouldCollide(W, W.p_body[hi], W.p_body[lo])) continue;
			if (!filterShouldCollide(W.p_filter0[lo], W.p_filter1[lo], W.p_filter0[hi], W.p_filter1[hi])) continue;
			if (b2dContactSwap(W.shapes[W.p_shape[lo]].type, W.shapes[W.p_shape[hi]].type) < 0) continue;
			// the two components are tied together by a contact the serial order would have created: both are replayed
			const int dP = W.toiDomOf[W.toiParent[bodyP]] - 1, dQ = W.toiDomOf[W.toiParent[bodyQ]] - 1;
			if (dP >= 0) W.toiDomFailed[dP] = 1;
			if (dQ >= 0) W.toiDomFailed[dQ] = 1;
			S->c.toiAnyFailed = 1;
		}
	}
	(void)C;
}

// Components that could not finish on their own go back to the snapshot (bodies, proxies, contacts) and their pending
// impacts form the list of the serial replay (k_toi_loop_partial); the events they had counted are taken back.
__global__ __launch_bounds__(256) void k_toi_dom_rollback(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.toiUnsafe) return; // the whole phase is redone anyway
	// (no component failed - nine steps of ten on config 5: nothing to take back, and the three passes below over every body,
	// proxy and contact - 20 us there - find that out the long way)
	if (__hip_atomic_load(&S->c.toiAnyFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
	const int nD = S->c.nToiDomains < TOI_DOMAINS_MAX ? S->c.nToiDomains : TOI_DOMAINS_MAX;
	const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
	const ContactArrays& A = W.ca[S->cur];
	const ContactArrays& B = W.ca[1 - S->cur];
	auto failedBody = [&](int body) -> bool
	{
		if ((W.b_flags[body] & BF_TYPE_MASK) == BT_STATIC) return false;
		const int d = W.toiDomOf[W.toiParent[body]] - 1;
		return d >= 0 && W.toiDomFailed[d] != 0;
	};
	for (int d = t0; d < nD; d += stride)
	{
		if (W.toiDomFailed[d] && W.toiDomEvents[d]) atomicSub(&S->c.nToiEvents, W.toiDomEvents[d]);
	}
	for (int i = t0; i < W.nBodies; i += stride)
	{
		if (!failedBody(i)) continue;
		W.b_pos[i] = W.snapBody[5 * (size_t)i + 0];
		W.b_pos0[i] = W.snapBody[5 * (size_t)i + 1];
		W.b_vel[i] = W.snapBody[5 * (size_t)i + 2];
		W.b_xf[i] = W.snapBody[5 * (size_t)i + 3];
		W.b_flags[i] = __float_as_uint(W.snapBody[5 * (size_t)i + 4].x);
		W.b_rowDirty[i] = 1;
	}
	for (int p = t0; p < W.nProxies; p += stride)
	{
		const int body = W.p_body[p];
		if (body >= 0 && failedBody(body)) W.p_fat[p] = W.snapFat[p];
	}
	const int nC = S->c.nContacts;
	for (int i = t0; i < nC; i += stride)
	{
		const int4 ids = A.ids[i];
		if (!failedBody(ids.z) && !failedBody(ids.w)) continue;
		A.flags[i] = B.flags[i];
		A.mat[i] = B.mat[i];
		A.man0[i] = B.man0[i];
		A.man1[i] = B.man1[i];
		A.imp[i] = B.imp[i];
		A.man3[i] = B.man3[i];
	}
	// the pending impacts of those components, as k_toi_first listed them (their flags are back to that state too)
	const int nL = S->c.nToiList < W.capContacts ? S->c.nToiList : W.capContacts;
	for (int k = t0; k < nL; k += stride)
	{
		const int c = W.toiList[k];
		const int4 ids = A.ids[c];
		if (!failedBody(ids.z) && !failedBody(ids.w)) continue;
		W.toiDomList[atomicAdd(&S->c.nToiPartial, 1)] = c;
	}
}

#endif
