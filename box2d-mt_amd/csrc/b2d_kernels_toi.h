// b2d_kernels_toi.h - continuous collision on the device: b2World::SolveTOI (b2World.cpp:1026-1093).
//
// Shape of the work in the reference: (1) one time-of-impact evaluation per TOI-candidate contact,
// embarrassingly parallel (b2FindMinToiContactTask, b2World.cpp:283-360); (2) a strictly serial event
// loop: take the earliest impact, advance the two bodies, build a mini island (<= 64 bodies, <= 32
// contacts), solve it, re-sync its proxies, create the contacts that movement produced, recompute the
// invalidated impacts, repeat (StepSolveTOI :851-1024).
//
// Mapping here: (1) is k_toi_first, one lane per TOI-candidate contact (the manager's slot table). (2) runs inside
// ONE persistent workgroup (k_toi_loop) so that an event costs no launch and no host round trip; inside
// an event everything that commutes is done by all lanes (candidate manifolds, proxy re-sync, pair
// search, impact recomputation) and only the order-defining decisions (which candidate enters the
// mini island, capacity cut-offs) are taken by one lane, in the reference's order: a body's contact list
// is newest first = descending contact index here (contacts live in creation order).
// The arg-min uses b2Contact::ToiLessThan (b2Contact.cpp:326-334): (alpha, proxyLow, proxyHigh).
#ifndef B2D_KERNELS_TOI_H
#define B2D_KERNELS_TOI_H

#include "b2d_kernels_broadphase.h"
#include "b2d_toi.h"

#define TOI_LANES 512
#define TOI_CAND_MAX 512     // candidate contacts of the two seed bodies in one event (one lane each)
#define TOI_MOVES_MAX 128
#define TOI_PAIRS_MAX 512
#define TOI_RECOMP_MAX 512
#define TOI_WOKEN_MAX 256
#define TOI_EVENTS_MAX 100000
#define TOI_NEWPAIR_MAX 256      // new pairs the parallel chains may leave for their close-out to create (more: serial loop)
#define TOI_MOVED_ALL_MAX 32768  // proxies re-inserted during one TOI phase (their grid bins are stale): DW::toiMoved

// Flags are updated with L2 atomics (wake-ups, claims, invalidation) inside the event loop; a plain load could
// be served from a stale L1 line of the same CU, so flag reads in this file go to L2 as relaxed atomic loads.
__device__ __forceinline__ uint32_t ldFlags(const uint32_t* p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ Sweep loadSweep(const DW& W, int body)
{
	const float4 p0 = W.b_pos0[body], p = W.b_pos[body], m = W.b_mass[body];
	Sweep s;
	s.localCenter = v2(m.z, m.w);
	s.c0 = v2(p0.x, p0.y);
	s.a0 = p0.z;
	s.alpha0 = p0.w;
	s.c = v2(p.x, p.y);
	s.a = p.z;
	return s;
}

// Candidate partners of a re-inserted proxy: every proxy whose fat AABB can overlap `a`, found through the hash grid
// (proxies are binned by the cell of their centre and are at most gridLimit wide, so the centres of all partners lie in
// `a` grown by half that limit: gridWindow), plus the proxies that are wider than a cell, plus the proxies whose fat AABB changed since the
// grid was built (`moved`: their bin is stale). f(q) may be called more than once for the same q.
template <typename F>
__device__ __forceinline__ void toiForEachCandidate(const DW& W, AABB a, int lane, int nLanes, const int* moved, int nMoved, F f)
{
	int ix0, iy0, nx, ny;
	if (!gridWindow(W, make_float4(a.lo.x, a.lo.y, a.hi.x, a.hi.y), &ix0, &iy0, &nx, &ny))
	{
		for (int q = lane; q < W.nProxies; q += nLanes) f(q); // degenerate or huge query: look at everything
		return;
	}
	const int nCells = nx * ny;
	for (int c = lane; c < nCells; c += nLanes)
	{
		const uint32_t h = cellHash(ix0 + c % nx, iy0 + c / nx, W.gridMask);
		const int start = W.gridStart[h], n = W.gridCount[h];
		for (int t = 0; t < n; ++t) f(W.gridItems[start + t]);
	}
	const int nLarge = W.st->c.nLargeProxies;
	for (int t = lane; t < nLarge; t += nLanes) f(W.largeProxies[t]);
	for (int t = lane; t < nMoved; t += nLanes) f(moved[t]);
}

// b2World::ComputeToi(c, alpha0) (b2World.cpp:401-444): the sweeps are already on the same interval.
__device__ __forceinline__ float computeToiOf(const GjkProxy& pA, const GjkProxy& pB, const Sweep& sA, const Sweep& sB)
{
	float t = 1.0f;
	const int state = b2dTimeOfImpact(&t, pA, sA, pB, sB, 1.0f);
	const float alpha0 = sA.alpha0;
	return state == TOI_TOUCHING ? b2dMin(alpha0 + (1.0f - alpha0) * t, 1.0f) : 1.0f;
}
__device__ __forceinline__ float computeToi(const DW& W, int4 ids, const Sweep& sA, const Sweep& sB)
{
	const GjkProxy pA = b2dProxy(W.shapes + W.p_shape[ids.x]);
	const GjkProxy pB = b2dProxy(W.shapes + W.p_shape[ids.y]);
	float t = 1.0f;
	const int state = b2dTimeOfImpact(&t, pA, sA, pB, sB, 1.0f);
	const float alpha0 = sA.alpha0;
	return state == TOI_TOUCHING ? b2dMin(alpha0 + (1.0f - alpha0) * t, 1.0f) : 1.0f;
}

// ... with the vertices of both shapes copied into `stage` first (LDS, 2 x B2D_MAX_POLY_VERTS of the calling lane): the root
// finder and its distance iterations read them index by index, hundreds of times - from LDS instead of through the vector L1
// (k_toi_first on the 1 M field: 204 -> 166 us; the same arithmetic on the same values)
__device__ __forceinline__ float computeToiStaged(const DW& W, int4 ids, const Sweep& sA, const Sweep& sB, V2* stage)
{
	const ShapeRec* rA = W.shapes + W.p_shape[ids.x];
	const ShapeRec* rB = W.shapes + W.p_shape[ids.y];
	GjkProxy pA = b2dProxy(rA), pB = b2dProxy(rB);
#pragma unroll
	for (int k = 0; k < B2D_MAX_POLY_VERTS; ++k) { stage[k] = rA->verts[k]; stage[B2D_MAX_POLY_VERTS + k] = rB->verts[k]; }
	pA.verts = stage;
	pB.verts = stage + B2D_MAX_POLY_VERTS;
	return computeToiOf(pA, pB, sA, sB);
}

// IsMinToiCandidate (b2Contact.h:403-418) + !e_inactiveFlag: may this contact take part in the arg-min?
__device__ __forceinline__ bool toiEligible(const DW& W, uint32_t flags, int4 ids)
{
	if ((flags & CF_TOI_CANDIDATE) == 0 || (flags & CF_ENABLED) == 0 || (flags & CF_FOREIGN) != 0) return false;
	if ((int)((flags & CF_TOI_COUNT_MASK) >> CF_TOI_COUNT_SHIFT) > B2D_MAX_SUB_STEPS) return false;
	return bodyActiveForContact(ldFlags(&W.b_flags[ids.z])) || bodyActiveForContact(ldFlags(&W.b_flags[ids.w]));
}

// First pass of b2World::FindMinToiContact: every sweep starts the step at alpha0 = 0, so no contact
// needs the out-of-sync path and all of them are independent.
__global__ __launch_bounds__(256) void k_toi_first(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	int calls = 0;
	if (W.toiContinue)
	{
		// a call that continues a step (sub-stepping): the impacts earlier calls found stay (b2World.cpp:1041-1045), only the
		// pending list is made again; what has no valid impact is computed by the event loop in the reference's order
		for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		{
			const uint32_t flags = C.flags[i] & ~(CF_TOI_LISTED | CF_TOI_PENDING);
			const bool listed = (flags & (CF_TOI | CF_FOREIGN)) == CF_TOI && C.mat[i].w < 1.0f;
			if (listed)
			{
				const int k = atomicAdd(&S->c.nToiList, 1);
				if (k < W.capContacts) W.toiList[k] = i;
			}
			C.flags[i] = flags | (listed ? CF_TOI_LISTED : 0u);
		}
		if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&S->c.toiUnsafe, 1); // (never the parallel chains)
		return;
	}
	// Only TOI candidates can be eligible or carry TOI state (CF_TOI_STATE_MASK: set below and by the event loops for eligible
	// contacts only, wiped by k_edit_* when a contact stops being a candidate), and the candidates are listed: the manager's slot
	// table (DW::toiPos2c[0, nToiOrder), kept by toiOrderDestroy / k_toi_order_create / the edits). A lane per SLOT instead of a
	// lane per contact: the million-body field has 1.2 M contacts and ~40 000 candidates, whose evaluations - thousands of
	// instructions each - were spread one or two to a wave over every wave of the launch (113 us); now they sit side by side.
	// (round 6) The vertices of a candidate's two shapes wait in LDS, a lane's own 128 bytes: the root finder and the distance
	// iterations inside it read them one at a time, index by index, hundreds to thousands of times per candidate - each a
	// round trip to the vector L1 in a chain (the field's million distinct records: the first touch of each comes from memory)
	// - and that chain, not the arithmetic, was the kernel's time.
	__shared__ V2 s_toiVerts[256][2 * B2D_MAX_POLY_VERTS];
	const int nSlots = S->c.nToiOrder < W.capContacts ? S->c.nToiOrder : W.capContacts;
	(void)n;
	for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nSlots; s += gridDim.x * blockDim.x)
	{
		const int i = W.toiPos2c[s];
		uint32_t flags = C.flags[i] & ~CF_TOI_STATE_MASK;
		const int4 ids = C.ids[i];
		if (toiEligible(W, flags, ids))
		{
			const Sweep sA = loadSweep(W, ids.z), sB = loadSweep(W, ids.w);
			const float alpha = computeToiStaged(W, ids, sA, sB, s_toiVerts[threadIdx.x]);
			++calls;
			float4 mat = C.mat[i];
			mat.w = alpha;
			C.mat[i] = mat;
			flags |= CF_TOI;
			if (alpha < 1.0f)
			{
				flags |= CF_TOI_LISTED;
				const int k = atomicAdd(&S->c.nToiList, 1);
				if (k < W.capContacts) W.toiList[k] = i;
				// the parallel chains need (dynamic, static) pairs; bullets / kinematic partners go to the serial loop
				const uint32_t fA = W.b_flags[ids.z], fB = W.b_flags[ids.w];
				const uint32_t tA = fA & BF_TYPE_MASK, tB = fB & BF_TYPE_MASK;
				const bool simple = ((tA == BT_DYNAMIC && tB == BT_STATIC) || (tB == BT_DYNAMIC && tA == BT_STATIC)) && ((fA | fB) & BF_BULLET) == 0;
				if (!simple) atomicOr(&S->c.toiUnsafe, 1);
			}
		}
		C.flags[i] = flags;
	}
	// (one add for the launch, carried by the arrival of the workgroups: b2d_world.h)
	b2dBlockTreeAdd2(W, ARRIVE_TOI_FIRST, &S->c.nToiCalls, calls, &S->c.nToiCalls, 0, (unsigned)W.capContacts <= TREE_SUM_MAX);
}

// ---- adjacency of ALL contacts by non-static body (the island CSR only holds solid touching ones) ----
__global__ __launch_bounds__(256) void k_toi_adj_clear(DW W)
{
	b2dPhaseStamp(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= W.nBodies; i += gridDim.x * blockDim.x)
	{
		W.deg[i] = 0;
		if (i < W.nBodies) W.adjCursor[i] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) W.st->c.toiBase = W.st->c.nContacts;
}

__global__ __launch_bounds__(256) void k_toi_adj_count(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		if ((W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC) atomicAdd(&W.deg[ids.z], 1);
		if ((W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC) atomicAdd(&W.deg[ids.w], 1);
	}
}

__global__ __launch_bounds__(256) void k_toi_adj_fill(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		if ((W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC) W.adj[W.adjStart[ids.z] + atomicAdd(&W.adjCursor[ids.z], 1)] = i;
		if ((W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC) W.adj[W.adjStart[ids.w] + atomicAdd(&W.adjCursor[ids.w], 1)] = i;
	}
}

// b2ClearBodySolveTOIFlags (b2World.cpp:239-259): sweeps go back to alpha0 = 0 for the next step
__global__ __launch_bounds__(256) void k_toi_clear(DW W)
{
	b2dPhaseStamp(W);
	// (ClearPostSolveTOI, b2World.cpp:1467-1504: when the step is complete, and only if anything was touched - by this call
	// or, sub-stepping, by the calls before it)
	if (W.st->c.toiIncomplete != 0) return;
	if (W.st->c.nToiEvents == 0 && !W.toiContinue) return;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		W.b_pos0[i].w = 0.0f;
	}
}

// ---- the event loop ----------------------------------------------------------------------------------
struct ToiCand
{
	int contact;
	int other;     // body on the far side
	int side;      // 0: from seed A's list, 1: from seed B's list
	int info;      // bit0 touching after the update, bit1 was touching, bit2 visited, bit3 adds `other` to the island
};

struct ToiPair
{
	uint64_t key;
	int lo, hi;
};

// b2Contact::UpdateImpl (b2Contact.cpp:173-298) for a non-sensor contact, evaluated at given transforms.
struct ToiUpdate
{
	float4 man0, man1, imp;
	int4 man3;
	float4 old0, old1, oldImp; // the manifold before the update (b2ContactListener::PreSolve is handed it)
	int4 old3;
	bool touching, wasTouching;
};

// The listener calls b2Contact::Update makes (b2Contact.cpp:253-297) as ONE log record: BeginContact when the contact starts
// touching, EndContact when it stops, PreSolve while it touches (TOI candidates are never sensors). `slot` is the record's
// place in DW::toiLog - the call order of the reference. The reported state (CF_REPORTED) follows what was logged.
// assumedOff: the sub-step went on as if this Update's PreSolve had switched the contact off (toiPreSolveOutcome): bit 4 of
// the record's kind, so that the host can tell whether the answer it gets now changes anything.
__device__ __forceinline__ void toiLogUpdate(const DW& W, const ContactArrays& C, int slot, int contact, int4 ids, const ToiUpdate& u, bool assumedOff)
{
	if (W.toiLog == nullptr || slot < 0 || slot >= W.capToiLog) return;
	int kind = 0;
	if (W.eventsOn && !u.wasTouching && u.touching) kind |= 1;
	if (W.eventsOn && u.wasTouching && !u.touching) kind |= 2;
	if (W.preSolveOn && u.touching) kind |= 4 | (assumedOff ? 16 : 0);
	if (W.eventsOn)
	{
		if (u.touching) atomicOr(&C.flags[contact], CF_REPORTED); else atomicAnd(&C.flags[contact], ~CF_REPORTED);
	}
	ToiLogRec r;
	r.info = make_int4(kind, contact, ids.x, ids.y);
	r.o0 = u.old0; r.o1 = u.old1; r.oimp = u.oldImp; r.o3 = u.old3;
	r.n0 = u.man0; r.n1 = u.man1; r.nimp = u.imp; r.n3 = u.man3;
	r.mat = C.mat[contact];
	W.toiLog[slot] = r;
}

// What the host's PreSolve answered for the Update logged at `slot` (DW::toiVerdict, see b2hip.hip: toiPreSolveRounds): the
// contact's material as the callback left it, and whether the callback switched the contact off (b2Contact::SetEnabled(false),
// b2Contact.h:117-123). Slots at or above DW::nToiVerdict have not been asked yet: there the sub-step ASSUMES the answer
// the listener gave for this contact the last time (CF_PRESOLVE_OFF, kept by the Collide phase's PreSolve and by the answers
// applied here) and says so in its log record - a listener that keeps answering the same costs no second run of the phase.
__device__ __forceinline__ bool toiPreSolveAsked(const DW& W, int slot)
{
	return W.toiVerdict != nullptr && slot >= 0 && slot < W.nToiVerdict && (W.toiVerdict[slot].x & 1) != 0;
}

// (read-only form for the order-defining walk, which runs before the lanes commit their Updates)
__device__ __forceinline__ bool toiPreSolveOff(const DW& W, const ContactArrays& C, int slot, int contact)
{
	if (toiPreSolveAsked(W, slot)) return (W.toiVerdict[slot].x & 2) != 0;
	return (ldFlags(&C.flags[contact]) & CF_PRESOLVE_OFF) != 0;
}

// (committing form: an answer's material and its off state go to the contact; CF_ENABLED is the caller's)
__device__ __forceinline__ bool toiPreSolveOutcome(const DW& W, const ContactArrays& C, int slot, int contact)
{
	if (!toiPreSolveAsked(W, slot)) return (ldFlags(&C.flags[contact]) & CF_PRESOLVE_OFF) != 0;
	const int4 v = W.toiVerdict[slot];
	float4 m = C.mat[contact];
	m.x = __int_as_float(v.y);
	m.y = __int_as_float(v.z);
	m.z = __int_as_float(v.w);
	C.mat[contact] = m;
	if (v.x & 2) atomicOr(&C.flags[contact], CF_PRESOLVE_OFF); else atomicAnd(&C.flags[contact], ~CF_PRESOLVE_OFF);
	return (v.x & 2) != 0;
}

__device__ __forceinline__ void toiEvaluate(const DW& W, const ContactArrays& C, int i, int4 ids, Xf xfA, Xf xfB, ToiUpdate* u)
{
	const int4 m3 = C.man3[i];
	const float4 oldImp = C.imp[i];
	const float4 o0 = C.man0[i], o1 = C.man1[i];
	const uint32_t oldId0 = (uint32_t)m3.x, oldId1 = (uint32_t)m3.y;
	const int oldCount = m3.w;
	Manifold mf;
	mf.pointCount = 0;
	mf.type = m3.z;
	mf.localNormal = v2(o0.x, o0.y);
	mf.localPoint = v2(o0.z, o0.w);
	mf.p[0] = v2(o1.x, o1.y);
	mf.p[1] = v2(o1.z, o1.w);
	mf.id[0] = oldId0;
	mf.id[1] = oldId1;
	b2dEvaluate(&mf, W.shapes + W.p_shape[ids.x], xfA, W.shapes + W.p_shape[ids.y], xfB);
	float ni[2], ti[2];
	ni[0] = oldImp.x; ti[0] = oldImp.y; ni[1] = oldImp.z; ti[1] = oldImp.w;
	for (int k = 0; k < mf.pointCount; ++k)
	{
		float n = 0.0f, t = 0.0f;
		const uint32_t id2 = mf.id[k];
		if (oldCount > 0 && oldId0 == id2) { n = oldImp.x; t = oldImp.y; }
		else if (oldCount > 1 && oldId1 == id2) { n = oldImp.z; t = oldImp.w; }
		ni[k] = n;
		ti[k] = t;
	}
	u->man0 = make_float4(mf.localNormal.x, mf.localNormal.y, mf.localPoint.x, mf.localPoint.y);
	u->man1 = make_float4(mf.p[0].x, mf.p[0].y, mf.p[1].x, mf.p[1].y);
	u->imp = make_float4(ni[0], ti[0], ni[1], ti[1]);
	u->man3 = make_int4((int)mf.id[0], (int)mf.id[1], mf.type, mf.pointCount);
	u->old0 = o0; u->old1 = o1; u->oldImp = oldImp; u->old3 = m3;
	u->touching = mf.pointCount > 0;
}

__device__ __forceinline__ void toiCommitUpdate(const ContactArrays& C, int i, const ToiUpdate& u)
{
	C.man0[i] = u.man0;
	C.man1[i] = u.man1;
	C.imp[i] = u.imp;
	C.man3[i] = u.man3;
	if (u.touching) atomicOr(&C.flags[i], CF_TOUCHING | CF_ENABLED);
	else
	{
		atomicAnd(&C.flags[i], ~CF_TOUCHING);
		atomicOr(&C.flags[i], CF_ENABLED);
	}
}

// Pose of a body advanced to `alpha` (b2Body::Advance, b2Body.h:964-972) without storing it.
__device__ __forceinline__ Sweep advancedSweep(const DW& W, int body, float alpha)
{
	Sweep s = loadSweep(W, body);
	b2dSweepAdvance(s, alpha);
	s.c = s.c0;
	s.a = s.a0;
	return s;
}

__device__ __forceinline__ void storeAdvanced(const DW& W, int body, const Sweep& s)
{
	const float sleepTime = W.b_pos[body].w;
	W.b_rowDirty[body] = 1;
	W.b_pos0[body] = make_float4(s.c0.x, s.c0.y, s.a0, s.alpha0);
	W.b_pos[body] = make_float4(s.c.x, s.c.y, s.a, sleepTime);
	const Xf xf = b2dXfFromSweep(s.c, s.a, s.localCenter);
	W.b_xf[body] = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
}

#define TOI_MOVED_MAX 4096     // proxies re-inserted by the parallel TOI paths in one step (more: serial loop)

// unsafe bits of the domain mode (shared with the chains: b2d_kernels_toi_chains.h)
#define TOI_DOM_UNSAFE_WOKE 2
#define TOI_DOM_UNSAFE_PAIR 4
#define TOI_DOM_UNSAFE_CAPACITY 8
#define TOI_DOM_MOVED_LOCAL 64

// The event loop of b2World::SolveTOI. DOMAIN = false: the whole world in one workgroup (k_toi_loop), the reference's
// order replayed event by event. DOMAIN = true: the same loop restricted to ONE connected component of the graph
// {non-static bodies, contacts between them} (k_toi_domains runs the components side by side, see
// b2d_kernels_toi_domains.h): its pending list is the component's slice of DW::toiDomList, static bodies are read-only
// partners that are always "in sync" (their alpha0 is neither read nor written), and the three things that would couple
// components - a new contact, a sleeping body woken, a capacity cut - raise Counters::toiUnsafe instead of being done.
// partial (DOMAIN = false only): the serial loop over the components that could not finish on their own (see
// k_toi_dom_rollback): its pending list is DW::toiDomList[0, nToiPartial), the proxies the finished components moved
// are already in DW::toiMoved, and a contact created with a body of a finished component means that component did not
// see it in time: Counters::toiUnsafe, and the whole phase is redone serially.
template <bool DOMAIN, int LANES>
__device__ __forceinline__ void toiLoopRun(const DW& W, const StepParams& sp, int domain, int partial)
{
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int tid = threadIdx.x;
	const int nC0 = S->c.toiBase; // contacts covered by the adjacency; [nC0, s_nC) is the tail created by events
	int* const list = DOMAIN ? W.toiDomList + W.toiDomBase[domain] : (partial ? W.toiDomList : W.toiList);
	const int listCap = DOMAIN ? W.toiDomCount[domain] : W.capContacts;
	const int myRoot = DOMAIN ? W.toiDomRoot[domain] : -1;
	__shared__ int s_unsafe, s_failed, s_movedLocal[TOI_DOM_MOVED_LOCAL], s_nMovedLocal;
	__shared__ int s_incomplete; // the call ends after its event cap with the step unfinished (sub-stepping)
	__shared__ int s_stepCalls;  // b2World::StepSolveTOI calls of this launch (solid or not: the cap counts calls)

	__shared__ int s_nC, s_nL, s_events, s_calls, s_overflow;
	__shared__ int s_logCursor; // next free record of DW::toiLog (listener calls of the sub-steps, in call order)
	__shared__ int s_minIdx;
	__shared__ float s_minAlpha;
	__shared__ unsigned long long s_best[LANES];
	__shared__ uint32_t s_bestAlpha[LANES];
	__shared__ int s_bestIdx[LANES];
	__shared__ int s_solid;
	__shared__ int s_bodies[B2D_MAX_TOI_BODIES], s_nBodies;
	__shared__ int s_contacts[B2D_MAX_TOI_CONTACTS], s_nContacts;
	__shared__ int s_level[B2D_MAX_TOI_CONTACTS], s_maxLevel;
	__shared__ float4 s_pos[B2D_MAX_TOI_BODIES], s_vel[B2D_MAX_TOI_BODIES];
	__shared__ uint32_t s_pen;
	__shared__ ToiCand s_cand[TOI_CAND_MAX], s_sorted[TOI_CAND_MAX];
	__shared__ int s_nCand;
	__shared__ int s_moves[TOI_MOVES_MAX], s_nMoves;
	__shared__ ToiPair s_pairs[TOI_PAIRS_MAX];
	__shared__ int s_pairFirst[TOI_PAIRS_MAX], s_pairRank[TOI_PAIRS_MAX];
	__shared__ int s_nPairs, s_nNew;
	__shared__ int s_recomp[TOI_RECOMP_MAX], s_nRecomp;
	__shared__ int s_rSorted[TOI_RECOMP_MAX], s_rSlot[TOI_RECOMP_MAX];
	__shared__ int s_flatBody[2 * TOI_RECOMP_MAX], s_flatFirst[2 * TOI_RECOMP_MAX];
	__shared__ float s_flatAlpha[2 * TOI_RECOMP_MAX];
	__shared__ int s_advFlat[TOI_RECOMP_MAX], s_nAdv;
	__shared__ float s_advAlpha[TOI_RECOMP_MAX];
	__shared__ int s_pairCand[TOI_PAIRS_MAX];
	__shared__ int s_toiOrder, s_nMovedAll;
	__shared__ int s_woken[TOI_WOKEN_MAX], s_nWoken;

	if (tid == 0)
	{
		s_nC = S->c.nContacts;
		const int nPending = partial ? S->c.nToiPartial : S->c.nToiList;
		s_nL = DOMAIN ? (W.toiDomFill[domain] < listCap ? W.toiDomFill[domain] : listCap) : (nPending < W.capContacts ? nPending : W.capContacts);
		s_unsafe = 0;
		s_failed = 0;
		s_nMovedLocal = 0;
		s_logCursor = S->c.nToiLog;
		s_events = 0;
		s_calls = 0;
		s_overflow = 0;
		s_toiOrder = S->c.nToiOrder;
		s_nMovedAll = partial ? (S->c.nToiMoved < TOI_MOVED_ALL_MAX ? S->c.nToiMoved : TOI_MOVED_ALL_MAX) : 0;
		s_nRecomp = 0;
		s_incomplete = 0;
		s_stepCalls = 0;
	}
	__syncthreads();
	// A call that continues a step (b2World::SetSubStepping, b2World.cpp:1041-1086: m_stepComplete false): nothing was computed
	// by a first pass - the impacts found by earlier calls are still valid (CF_TOI), the contacts of the bodies the last
	// event displaced, the contacts it created and whatever Collide made eligible since are not. The single-threaded
	// FindMinToiContact computes them on its way through the contact array: all of them form the first batch here.
	if (!DOMAIN && !partial && W.toiContinue)
	{
		for (int c = tid; c < s_nC; c += LANES)
		{
			const uint32_t flags = ldFlags(&C.flags[c]);
			if ((flags & (CF_TOI | CF_TOI_PENDING)) != 0 || !toiEligible(W, flags, C.ids[c])) continue;
			atomicOr(&C.flags[c], CF_TOI_PENDING);
			const int k = atomicAdd(&s_nRecomp, 1);
			if (k < TOI_RECOMP_MAX) s_recomp[k] = c; else atomicOr(&s_overflow, 8);
		}
		__syncthreads();
	}

	// SetAwake(true) (b2Body.h:690-718) + remember bodies that were asleep: their dormant contacts become eligible
	auto wake = [&](int body)
	{
		const uint32_t old = atomicOr(&W.b_flags[body], BF_AWAKE);
		W.b_pos[body].w = 0.0f;
		W.b_rowDirty[body] = 1;
		if ((old & BF_AWAKE) == 0)
		{
			// (a woken body resting on a static one would read that body's alpha0, which other components also advance)
			if (DOMAIN) atomicOr(&s_unsafe, TOI_DOM_UNSAFE_WOKE);
			else if (partial) atomicOr(&S->c.toiUnsafe, TOI_DOM_UNSAFE_WOKE);
			const int k = atomicAdd(&s_nWoken, 1);
			if (k < TOI_WOKEN_MAX) s_woken[k] = body; else atomicOr(&s_overflow, 8);
		}
	};

	// phase timers (100 MHz ticks, accumulated by lane 0; written to DW::hubList[0..15], which is idle during the TOI phase)
	__shared__ unsigned long long s_t[12];
	unsigned long long tPrev = wall_clock64();
	if (tid < 12) s_t[tid] = 0;
#define TOI_T(k) do { if (tid == 0) { const unsigned long long now = wall_clock64(); s_t[k] += now - tPrev; tPrev = now; } } while (0)
	for (;;)
	{
		// ---- the impacts FindMinToiContact has to (re)compute before it can look for the minimum (b2World.cpp:1525-1611: a
		// contact without e_toiFlag): collected at the end of the previous event, or - in a call that continues a step
		// (sub-stepping) - by the scan above
		{
			const int nRecomp = s_nRecomp < TOI_RECOMP_MAX ? s_nRecomp : TOI_RECOMP_MAX;
			// Put the sweeps of each pair on the same interval (b2World.cpp:385-396): the lagging body advances to
			// the other's alpha0. A body can meet partners at different times within one pass, so the advances are
			// replayed in the reference's visiting order: the slot order of its contact array (ContactArrays::mgr).
			for (int r = tid; r < nRecomp; r += LANES) s_rSlot[r] = C.mgr[s_recomp[r]];
			__syncthreads();
			for (int r = tid; r < nRecomp; r += LANES)
			{
				const int m = s_rSlot[r];
				int rank = 0;
				for (int j = 0; j < nRecomp; ++j) rank += s_rSlot[j] < m ? 1 : 0;
				s_rSorted[rank] = s_recomp[r];
			}
			__syncthreads();
			for (int r = tid; r < nRecomp; r += LANES)
			{
				const int4 ids = C.ids[s_rSorted[r]];
				s_flatBody[2 * r] = ids.z;
				s_flatBody[2 * r + 1] = ids.w;
			}
			if (tid == 0) s_nAdv = 0;
			__syncthreads();
			for (int q = tid; q < 2 * nRecomp; q += LANES)
			{
				const int b = s_flatBody[q];
				int first = q;
				for (int j = 0; j < q; ++j)
				{
					if (s_flatBody[j] == b)
					{
						first = j;
						break;
					}
				}
				s_flatFirst[q] = first;
				if (first == q) s_flatAlpha[q] = W.b_pos0[b].w;
			}
			__syncthreads();
			if (tid == 0)
			{
				int nAdv = 0;
				for (int r = 0; r < nRecomp; ++r)
				{
					const int fa = s_flatFirst[2 * r], fb = s_flatFirst[2 * r + 1];
					if (DOMAIN)
					{
						// a static partner is always level with or behind the body it is paired with here (every body of the mini
						// island was advanced to the event time, and event times never decrease): advancing it changes nothing
						const bool stA = (ldFlags(&W.b_flags[s_flatBody[fa]]) & BF_TYPE_MASK) == BT_STATIC;
						const bool stB = (ldFlags(&W.b_flags[s_flatBody[fb]]) & BF_TYPE_MASK) == BT_STATIC;
						if (stA || stB) continue;
					}
					const float aA = s_flatAlpha[fa], aB = s_flatAlpha[fb];
					if (aA < aB)
					{
						s_advFlat[nAdv] = fa;
						s_advAlpha[nAdv++] = aB;
						s_flatAlpha[fa] = aB;
					}
					else if (aB < aA)
					{
						s_advFlat[nAdv] = fb;
						s_advAlpha[nAdv++] = aA;
						s_flatAlpha[fb] = aA;
					}
				}
				s_nAdv = nAdv;
			}
			__syncthreads();
			for (int q = tid; q < 2 * nRecomp; q += LANES)
			{
				if (s_flatFirst[q] != q) continue;
				const int nAdv = s_nAdv;
				bool any = false;
				Sweep sw;
				for (int k = 0; k < nAdv; ++k)
				{
					if (s_advFlat[k] != q) continue;
					if (!any) sw = loadSweep(W, s_flatBody[q]);
					any = true;
					b2dSweepAdvance(sw, s_advAlpha[k]);
				}
				if (any) { W.b_pos0[s_flatBody[q]] = make_float4(sw.c0.x, sw.c0.y, sw.a0, sw.alpha0); W.b_rowDirty[s_flatBody[q]] = 1; }
			}
			__syncthreads();
			for (int r = tid; r < nRecomp; r += LANES)
			{
				const int c = s_rSorted[r];
				const int4 ids = C.ids[c];
				Sweep sA = loadSweep(W, ids.z), sB = loadSweep(W, ids.w);
				if (DOMAIN)
				{
					if ((ldFlags(&W.b_flags[ids.z]) & BF_TYPE_MASK) == BT_STATIC) sA.alpha0 = sB.alpha0;
					else if ((ldFlags(&W.b_flags[ids.w]) & BF_TYPE_MASK) == BT_STATIC) sB.alpha0 = sA.alpha0;
				}
				const float alpha = computeToi(W, ids, sA, sB); // (staged through LDS as in k_toi_first: measured, no difference to the step - the event loops wait for one another's events, not for this)
				atomicAdd(&s_calls, 1);
				float4 mat = C.mat[c];
				mat.w = alpha;
				C.mat[c] = mat;
				uint32_t flags = (ldFlags(&C.flags[c]) & ~CF_TOI_PENDING) | CF_TOI;
				if (alpha < 1.0f && (flags & CF_TOI_LISTED) == 0)
				{
					flags |= CF_TOI_LISTED;
					const int k = atomicAdd(&s_nL, 1);
					if (k < listCap) list[k] = c; else atomicOr(&s_overflow, 16);
				}
				C.flags[c] = flags;
			}
			__syncthreads();
			if (tid == 0 && s_nL > listCap) s_nL = listCap;
			__syncthreads();
			if (tid == 0) s_nRecomp = 0;
			__syncthreads();
		}
		// ---- FindMinToiContact: lexicographic min of (alpha, proxyLow, proxyHigh) over the pending list ----
		{
			uint32_t bestA = 0xffffffffu;
			unsigned long long bestK = ~0ull;
			int bestI = -1;
			const int nL = s_nL;
			for (int k = tid; k < nL; k += LANES)
			{
				const int i = list[k];
				const uint32_t flags = ldFlags(&C.flags[i]);
				const int4 ids = C.ids[i];
				if ((flags & CF_TOI) == 0 || !toiEligible(W, flags, ids)) continue;
				const uint32_t a = __float_as_uint(C.mat[i].w); // alpha >= 0: bit order == value order
				const unsigned long long key = C.key[i];
				if (a < bestA || (a == bestA && key < bestK))
				{
					bestA = a;
					bestK = key;
					bestI = i;
				}
			}
			s_bestAlpha[tid] = bestA;
			s_best[tid] = bestK;
			s_bestIdx[tid] = bestI;
			__syncthreads();
			for (int off = LANES / 2; off > 0; off >>= 1)
			{
				if (tid < off)
				{
					const uint32_t a = s_bestAlpha[tid + off];
					const unsigned long long k2 = s_best[tid + off];
					if (a < s_bestAlpha[tid] || (a == s_bestAlpha[tid] && k2 < s_best[tid]))
					{
						s_bestAlpha[tid] = a;
						s_best[tid] = k2;
						s_bestIdx[tid] = s_bestIdx[tid + off];
					}
				}
				__syncthreads();
			}
			if (tid == 0)
			{
				s_minIdx = s_bestIdx[0];
				s_minAlpha = s_bestIdx[0] >= 0 ? __uint_as_float(s_bestAlpha[0]) : 1.0f;
			}
			__syncthreads();
		}
		const int minIdx = s_minIdx;
		const float minAlpha = s_minAlpha;
		if (minIdx < 0 || 1.0f - 10.0f * B2D_EPSILON < minAlpha) break;
		if (DOMAIN && (s_unsafe || s_overflow || s_failed)) break;
		if (s_events >= TOI_EVENTS_MAX)
		{
			// safety net against a runaway loop (the reference bounds it through b2_maxSubSteps per contact)
			if (tid == 0) s_overflow |= 32;
			break;
		}

		TOI_T(0);
		// ---- StepSolveTOI (b2World.cpp:851-1024) ------------------------------------------------------------
		const int4 minIds = C.ids[minIdx];
		const int seedA = minIds.z, seedB = minIds.w;
		if (tid == 0)
		{
			s_nWoken = 0;
			s_stepCalls += 1;
			s_nCand = 0;
			s_nMoves = 0;
			s_nPairs = 0;
			s_nNew = 0;
			Sweep a = loadSweep(W, seedA), b = loadSweep(W, seedB);
			b2dSweepAdvance(a, minAlpha); a.c = a.c0; a.a = a.a0;
			b2dSweepAdvance(b, minAlpha); b.c = b.c0; b.a = b.a0;
			const Xf xfA = b2dXfFromSweep(a.c, a.a, a.localCenter), xfB = b2dXfFromSweep(b.c, b.a, b.localCenter);
			// the TOI contact likely has some new contact points
			ToiUpdate u;
			u.wasTouching = (ldFlags(&C.flags[minIdx]) & CF_TOUCHING) != 0;
			toiEvaluate(W, C, minIdx, minIds, xfA, xfB, &u);
			toiCommitUpdate(C, minIdx, u);
			bool switchedOff = false; // (by the listener's PreSolve, asked in an earlier round of this phase)
			if (W.toiLog != nullptr)
			{
				const int slot = s_logCursor++;
				const bool assumedOff = u.touching && W.preSolveOn && !toiPreSolveAsked(W, slot) && toiPreSolveOff(W, C, slot, minIdx);
				toiLogUpdate(W, C, slot, minIdx, minIds, u, assumedOff);
				if (u.touching && W.preSolveOn) switchedOff = toiPreSolveOutcome(W, C, slot, minIdx);
			}
			uint32_t f = ldFlags(&C.flags[minIdx]);
			const uint32_t cnt = ((f & CF_TOI_COUNT_MASK) >> CF_TOI_COUNT_SHIFT) + 1u;
			f = (f & ~(CF_TOI | CF_TOI_COUNT_MASK)) | (cnt << CF_TOI_COUNT_SHIFT);
			if (u.touching != u.wasTouching)
			{
				wake(seedA);
				wake(seedB);
			}
			if (!u.touching || switchedOff)
			{
				// not solid (b2World.cpp:873-881): disable the contact and restore the sweeps (xf was derived from the same c, a)
				f &= ~CF_ENABLED;
				s_solid = 0;
			}
			else
			{
				s_solid = 1;
				if (!DOMAIN || (ldFlags(&W.b_flags[seedA]) & BF_TYPE_MASK) != BT_STATIC) storeAdvanced(W, seedA, a);
				if (!DOMAIN || (ldFlags(&W.b_flags[seedB]) & BF_TYPE_MASK) != BT_STATIC) storeAdvanced(W, seedB, b);
				wake(seedA);
				wake(seedB);
				s_bodies[0] = seedA;
				s_bodies[1] = seedB;
				s_nBodies = 2;
				s_contacts[0] = minIdx;
				s_nContacts = 1;
			}
			C.flags[minIdx] = f;
		}
		__syncthreads();
		// A contact that is not solid at its time of impact (no manifold points) is disabled and its bodies keep
		// their sweeps; only the wake-ups of that update remain to be accounted for below.
		const bool solid = s_solid != 0;
		if (solid)
		{

		TOI_T(1);
		// ---- gather the candidate contacts of the two seeds (dynamic seeds only) -------------------------------
		for (int side = 0; side < 2; ++side)
		{
			const int X = side == 0 ? seedA : seedB;
			const uint32_t fX = ldFlags(&W.b_flags[X]);
			if ((fX & BF_TYPE_MASK) != BT_DYNAMIC) continue;
			const int e0 = W.adjStart[X], e1 = W.adjStart[X + 1];
			const int nTail = s_nC - nC0;
			for (int e = tid; e < (e1 - e0) + nTail; e += LANES)
			{
				int c;
				if (e < e1 - e0) c = W.adj[e0 + e];
				else c = nC0 + (e - (e1 - e0));
				if (c == minIdx) continue;
				const int4 ids = C.ids[c];
				if (ids.z != X && ids.w != X) continue;
				const int other = ids.z == X ? ids.w : ids.z;
				const uint32_t fO = ldFlags(&W.b_flags[other]);
				// only static, kinematic or bullet bodies join
				if ((fO & BF_TYPE_MASK) == BT_DYNAMIC && ((fX | fO) & BF_BULLET) == 0) continue;
				if (ldFlags(&C.flags[c]) & CF_SENSOR) continue;
				const int k = atomicAdd(&s_nCand, 1);
				if (k < TOI_CAND_MAX)
				{
					s_cand[k].contact = c;
					s_cand[k].other = other;
					s_cand[k].side = side;
					s_cand[k].info = 0;
				}
				else atomicOr(&s_overflow, 1);
			}
		}
		__syncthreads();
		// (one lane per candidate below: a narrow workgroup - the components' single wave - takes as many as it has lanes, more
		// is a capacity cut like any other: the serial loop's 512 lanes get the phase)
		const int candCap = TOI_CAND_MAX < LANES ? TOI_CAND_MAX : LANES;
		if (tid == 0 && s_nCand > candCap) s_overflow |= 1;
		const int nCand = s_nCand < candCap ? s_nCand : candCap;
		// order: seed A's list first, each list newest contact first
		if (tid < nCand)
		{
			const ToiCand me = s_cand[tid];
			int rank = 0;
			for (int j = 0; j < nCand; ++j)
			{
				const ToiCand o = s_cand[j];
				if (o.side < me.side || (o.side == me.side && o.contact > me.contact)) ++rank;
			}
			s_sorted[rank] = me;
		}
		__syncthreads();

		TOI_T(2);
		// ---- tentative update of every candidate with both bodies at the time of impact ------------------------
		ToiUpdate upd;
		Sweep otherAdv;
		int myContact = -1, myOther = -1;
		if (tid < nCand)
		{
			const ToiCand me = s_sorted[tid];
			myContact = me.contact;
			myOther = me.other;
			const int4 ids = C.ids[myContact];
			const int X = me.side == 0 ? seedA : seedB;
			const bool otherIsSeed = myOther == seedA || myOther == seedB;
			Xf xfX = loadXf(W.b_xf, X), xfO;
			if (otherIsSeed)
			{
				xfO = loadXf(W.b_xf, myOther);
				otherAdv = loadSweep(W, myOther);
			}
			else
			{
				otherAdv = advancedSweep(W, myOther, minAlpha);
				xfO = b2dXfFromSweep(otherAdv.c, otherAdv.a, otherAdv.localCenter);
			}
			upd.wasTouching = (ldFlags(&C.flags[myContact]) & CF_TOUCHING) != 0;
			toiEvaluate(W, C, myContact, ids, ids.z == X ? xfX : xfO, ids.z == X ? xfO : xfX, &upd);
			s_sorted[tid].info = (upd.touching ? 1 : 0) | (upd.wasTouching ? 2 : 0);
		}
		__syncthreads();

		TOI_T(3);
		// ---- the order-defining walk (b2World.cpp:899-985) -------------------------------------------------------
		if (tid == 0)
		{
			int nB = s_nBodies, nK = s_nContacts;
			int blockedSide = -1;
			for (int r = 0; r < nCand; ++r)
			{
				ToiCand cd = s_sorted[r];
				if (cd.side == blockedSide) continue;
				if (nB == B2D_MAX_TOI_BODIES || nK == B2D_MAX_TOI_CONTACTS)
				{
					blockedSide = cd.side;
					continue;
				}
				bool inIsland = false;
				for (int j = 0; j < nK; ++j) inIsland = inIsland || s_contacts[j] == cd.contact;
				if (inIsland) continue;
				cd.info |= 4; // visited: its update is committed
				bool switchedOff = false;
				if (W.toiLog != nullptr)
				{
					const int slot = s_logCursor++; // (... and its listener calls logged, in this order)
					cd.info |= slot << 8;
					// a contact its PreSolve switched off stays out of the sub-step's island, its partner where it was (b2World.cpp:948-954)
					switchedOff = (cd.info & 1) != 0 && W.preSolveOn && toiPreSolveOff(W, C, slot, cd.contact);
				}
				if ((cd.info & 1) && !switchedOff)
				{
					s_contacts[nK++] = cd.contact;
					bool bodyIn = false;
					for (int j = 0; j < nB; ++j) bodyIn = bodyIn || s_bodies[j] == cd.other;
					if (!bodyIn)
					{
						s_bodies[nB++] = cd.other;
						cd.info |= 8;
					}
				}
				s_sorted[r].info = cd.info;
			}
			s_nBodies = nB;
			s_nContacts = nK;
		}
		__syncthreads();
		if (tid < nCand)
		{
			const int info = s_sorted[tid].info;
			if (info & 4)
			{
				toiCommitUpdate(C, myContact, upd);
				if (W.toiLog != nullptr)
				{
					const int slot = info >> 8;
					const bool assumedOff = upd.touching && W.preSolveOn && !toiPreSolveAsked(W, slot) && toiPreSolveOff(W, C, slot, myContact);
					toiLogUpdate(W, C, slot, myContact, C.ids[myContact], upd, assumedOff);
					if (upd.touching && W.preSolveOn && toiPreSolveOutcome(W, C, slot, myContact)) atomicAnd(&C.flags[myContact], ~CF_ENABLED);
				}
				if (upd.touching != upd.wasTouching)
				{
					const int4 ids = C.ids[myContact];
					wake(ids.z);
					wake(ids.w);
				}
			}
		}
		__syncthreads();
		if (tid < nCand)
		{
			const int info = s_sorted[tid].info;
			if (info & 8)
			{
				const bool otherStatic = (ldFlags(&W.b_flags[myOther]) & BF_TYPE_MASK) == BT_STATIC;
				if (!DOMAIN || !otherStatic) storeAdvanced(W, myOther, otherAdv);
				if (!otherStatic) wake(myOther);
			}
		}
		__syncthreads();

		TOI_T(4);
		// ---- b2Island::SolveTOI (b2Island.cpp:398-530): one constraint per lane, dependency levels -------------
		const int nB = s_nBodies, nK = s_nContacts; // (shadows nothing: the outer nB is declared after this block)
		const float h = (1.0f - minAlpha) * sp.dt;
		if (tid < nB)
		{
			const int b = s_bodies[tid];
			const float4 p = W.b_pos[b], v = W.b_vel[b];
			s_pos[tid] = make_float4(p.x, p.y, p.z, 0.0f);
			s_vel[tid] = make_float4(v.x, v.y, v.z, 0.0f);
		}
		if (tid == 0)
		{
			int maxLevel = 0;
			for (int i = 0; i < nK; ++i)
			{
				const int4 a = C.ids[s_contacts[i]];
				int lv = 1;
				for (int j = 0; j < i; ++j)
				{
					const int4 b = C.ids[s_contacts[j]];
					const bool share = a.z == b.z || a.z == b.w || a.w == b.z || a.w == b.w;
					if (share && s_level[j] + 1 > lv) lv = s_level[j] + 1;
				}
				s_level[i] = lv;
				if (lv > maxLevel) maxLevel = lv;
			}
			s_maxLevel = maxLevel;
		}
		__syncthreads();
		ContactConstraint cc;
		int ci = -1, la = 0, lb = 0, level = 0;
		Manifold mf;
		float4 cmat = make_float4(0, 0, 0, 0), mA4 = cmat, mB4 = cmat;
		float radiusA = 0.0f, radiusB = 0.0f;
		if (tid < nK)
		{
			ci = s_contacts[tid];
			level = s_level[tid];
			const int4 ids = C.ids[ci];
			for (int j = 0; j < nB; ++j)
			{
				if (s_bodies[j] == ids.z) la = j;
				if (s_bodies[j] == ids.w) lb = j;
			}
			mA4 = W.b_mass[ids.z];
			mB4 = W.b_mass[ids.w];
			radiusA = W.shapes[W.p_shape[ids.x]].radius;
			radiusB = W.shapes[W.p_shape[ids.y]].radius;
			cmat = C.mat[ci];
			const float4 m0 = C.man0[ci], m1 = C.man1[ci], im = C.imp[ci];
			const int4 m3 = C.man3[ci];
			mf.localNormal = v2(m0.x, m0.y);
			mf.localPoint = v2(m0.z, m0.w);
			mf.p[0] = v2(m1.x, m1.y);
			mf.p[1] = v2(m1.z, m1.w);
			mf.ni[0] = im.x; mf.ti[0] = im.y; mf.ni[1] = im.z; mf.ti[1] = im.w;
			mf.id[0] = (uint32_t)m3.x; mf.id[1] = (uint32_t)m3.y;
			mf.type = m3.z;
			mf.pointCount = m3.w;
		}
		const int maxLevel = s_maxLevel;
		auto initConstraint = [&]()
		{
			BodyPos pA, pB;
			BodyVel vA, vB;
			const float4 pa = s_pos[la], va = s_vel[la], pb = s_pos[lb], vb = s_vel[lb];
			pA.c = v2(pa.x, pa.y); pA.a = pa.z; vA.v = v2(va.x, va.y); vA.w = va.z;
			pB.c = v2(pb.x, pb.y); pB.a = pb.z; vB.v = v2(vb.x, vb.y); vB.w = vb.z;
			b2dInitConstraint(&cc, &mf, cmat.x, cmat.y, cmat.z,
				mA4.x, mA4.y, v2(mA4.z, mA4.w), radiusA,
				mB4.x, mB4.y, v2(mB4.z, mB4.w), radiusB,
				pA, vA, pB, vB, false, 1.0f);
		};
		// The mini island has at most 64 bodies and 32 constraints: everything below runs in the first wave, with wave-level
		// ordering instead of workgroup barriers (the LDS executes one wave's instructions in order); the other waves wait.
#define TOI_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
		if (tid < 64)
		{
		if (ci >= 0) initConstraint();
		TOI_WAVE_SYNC();
		// SolveTOIPositionConstraints (b2ContactSolver.cpp:755-843): only the two TOI bodies (island slots 0, 1) move
		for (int it = 0; it < 20; ++it)
		{
			if (tid == 0) s_pen = 0;
			TOI_WAVE_SYNC();
			for (int L = 1; L <= maxLevel; ++L)
			{
				if (ci >= 0 && level == L)
				{
					ContactConstraint pc = cc;
					if (la > 1) { pc.invMassA = 0.0f; pc.invIA = 0.0f; }
					if (lb > 1) { pc.invMassB = 0.0f; pc.invIB = 0.0f; }
					BodyPos pA, pB;
					const float4 pa = s_pos[la], pb = s_pos[lb];
					pA.c = v2(pa.x, pa.y); pA.a = pa.z;
					pB.c = v2(pb.x, pb.y); pB.a = pb.z;
					float minSep = 0.0f;
					b2dSolvePosition(&pc, &pA, &pB, B2D_TOI_BAUMGARTE, &minSep);
					s_pos[la] = make_float4(pA.c.x, pA.c.y, pA.a, 0.0f);
					s_pos[lb] = make_float4(pB.c.x, pB.c.y, pB.a, 0.0f);
					atomicMax(&s_pen, floatBits(0.0f - minSep));
				}
				TOI_WAVE_SYNC();
			}
			const float minSeparation = -__uint_as_float(s_pen);
			TOI_WAVE_SYNC();
			if (minSeparation >= -1.5f * B2D_LINEAR_SLOP) break;
		}
		// leap of faith to the new safe state (b2Island.cpp:466-470)
		if (tid < 2)
		{
			const int b = s_bodies[tid];
			const float4 p = s_pos[tid];
			if (!DOMAIN || (ldFlags(&W.b_flags[b]) & BF_TYPE_MASK) != BT_STATIC) { W.b_pos0[b] = make_float4(p.x, p.y, p.z, minAlpha); W.b_rowDirty[b] = 1; }
		}
		if (ci >= 0) initConstraint();
		TOI_WAVE_SYNC();
		for (int it = 0; it < sp.velIters; ++it)
		{
			for (int L = 1; L <= maxLevel; ++L)
			{
				if (ci >= 0 && level == L)
				{
					BodyVel vA, vB;
					const float4 va = s_vel[la], vb = s_vel[lb];
					vA.v = v2(va.x, va.y); vA.w = va.z;
					vB.v = v2(vb.x, vb.y); vB.w = vb.z;
					b2dSolveVelocity(&cc, &vA, &vB);
					s_vel[la] = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
					s_vel[lb] = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
				}
				TOI_WAVE_SYNC();
			}
		}
		// b2Island::Report (b2Island.cpp:527, 532-570): PostSolve with the sub-step's impulses, contact after contact
		if (W.toiLog != nullptr && W.postSolveOn && ci >= 0 && s_logCursor + tid < W.capToiLog)
		{
			ToiLogRec r;
			const int4 ids = C.ids[ci];
			r.info = make_int4(8, ci, ids.x, ids.y);
			r.o0 = r.o1 = r.oimp = make_float4(0, 0, 0, 0);
			r.o3 = make_int4(0, 0, 0, 0);
			r.n0 = C.man0[ci]; r.n1 = C.man1[ci];
			r.nimp = make_float4(cc.normalImpulse[0], cc.tangentImpulse[0], cc.pointCount > 1 ? cc.normalImpulse[1] : 0.0f, cc.pointCount > 1 ? cc.tangentImpulse[1] : 0.0f);
			r.n3 = C.man3[ci];
			r.n3.w = cc.pointCount;
			r.mat = cmat;
			W.toiLog[s_logCursor + tid] = r;
		}
		} // first wave
#undef TOI_WAVE_SYNC
		__syncthreads();
		if (tid == 0 && W.toiLog != nullptr && W.postSolveOn) s_logCursor += nK;
		// integrate positions, sync bodies (b2Island.cpp:483-527); TOI impulses are not stored
		if (tid < nB)
		{
			const int b = s_bodies[tid];
			const float4 p = s_pos[tid], v = s_vel[tid];
			V2 c = v2(p.x, p.y), vv = v2(v.x, v.y);
			float a = p.z, w = v.z;
			b2dIntegratePosition(&c, &a, &vv, &w, h);
			const float4 m = W.b_mass[b];
			const float sleepTime = W.b_pos[b].w;
			W.b_pos[b] = make_float4(c.x, c.y, a, sleepTime);
			W.b_vel[b] = make_float4(vv.x, vv.y, w, 0.0f);
			W.b_rowDirty[b] = 1;
			const Xf xf = b2dXfFromSweep(c, a, v2(m.z, m.w));
			W.b_xf[b] = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
		}
		__syncthreads();

		TOI_T(5);
		// ---- b2Body::SynchronizeFixtures of the island's dynamic bodies (b2World.cpp:1000-1011) -----------------
		if (tid < nB)
		{
			const int b = s_bodies[tid];
			if ((ldFlags(&W.b_flags[b]) & BF_TYPE_MASK) == BT_DYNAMIC)
			{
				const float4 m = W.b_mass[b], p0 = W.b_pos0[b];
				const Xf xf1 = b2dXfFromSweep(v2(p0.x, p0.y), p0.z, v2(m.z, m.w));
				const Xf xf2 = loadXf(W.b_xf, b);
				for (int p = W.b_proxyHead[b]; p >= 0; p = W.p_next[p])
				{
					const ShapeRec* shape = W.shapes + W.p_shape[p];
					const AABB aabb = b2dAabbCombine(b2dShapeAABB(shape, xf1), b2dShapeAABB(shape, xf2));
					if (b2dAabbContains(loadAabb(W.p_fat, p), aabb)) continue;
					const V2 d = B2D_AABB_MULTIPLIER * (xf2.p - xf1.p);
					AABB f = aabb;
					f.lo = v2(f.lo.x - B2D_AABB_EXTENSION, f.lo.y - B2D_AABB_EXTENSION);
					f.hi = v2(f.hi.x + B2D_AABB_EXTENSION, f.hi.y + B2D_AABB_EXTENSION);
					if (d.x < 0.0f) f.lo.x += d.x; else f.hi.x += d.x;
					if (d.y < 0.0f) f.lo.y += d.y; else f.hi.y += d.y;
					W.p_fat[p] = make_float4(f.lo.x, f.lo.y, f.hi.x, f.hi.y);
					const int k = atomicAdd(&s_nMoves, 1);
					if (k < TOI_MOVES_MAX) s_moves[k] = p; else atomicOr(&s_overflow, 2);
				}
			}
		}
		__syncthreads();

		TOI_T(6);
		// ---- FindNewContacts for the moved proxies (b2World.cpp:1013-1023) through the hash grid -------------------------
		const int nMoves = s_nMoves < TOI_MOVES_MAX ? s_nMoves : TOI_MOVES_MAX;
		if (tid == 0)
		{
			// every proxy re-inserted during this TOI phase has a stale grid bin from now on
			for (int mI = 0; mI < nMoves; ++mI)
			{
				if (!DOMAIN)
				{
					if (s_nMovedAll < TOI_MOVED_ALL_MAX) W.toiMoved[s_nMovedAll++] = s_moves[mI]; else s_overflow |= 64;
					continue;
				}
				// component mode: the proxies this component has moved (for its own searches) ...
				const int p = s_moves[mI];
				bool mine = false;
				for (int j = 0; j < s_nMovedLocal; ++j) mine = mine || s_movedLocal[j] == p;
				if (!mine)
				{
					if (s_nMovedLocal < TOI_DOM_MOVED_LOCAL) s_movedLocal[s_nMovedLocal++] = p; else s_unsafe |= TOI_DOM_UNSAFE_CAPACITY;
					// ... and the world-wide list with the hull of every fat AABB the proxy has had in this phase, for the
					// cross-component check afterwards (k_toi_domains_end)
					const int k = atomicAdd(&S->c.nToiMoved, 1);
					if (k < TOI_MOVED_MAX) W.toiMoved[k] = p; else s_unsafe |= TOI_DOM_UNSAFE_CAPACITY;
					W.toiHull[p] = W.snapFat[p];
				}
				const float4 f = W.p_fat[p], hcur = W.toiHull[p];
				W.toiHull[p] = make_float4(fminf(hcur.x, f.x), fminf(hcur.y, f.y), fmaxf(hcur.z, f.z), fmaxf(hcur.w, f.w));
			}
		}
		__syncthreads();
		const int nMovedAll = DOMAIN ? s_nMovedLocal : s_nMovedAll;
		const int* const movedList = DOMAIN ? s_movedLocal : W.toiMoved;
		for (int mI = 0; mI < nMoves; ++mI)
		{
			const int p = s_moves[mI];
			const AABB fp = loadAabb(W.p_fat, p);
			const int bodyP = W.p_body[p];
			toiForEachCandidate(W, fp, tid, LANES, movedList, nMovedAll, [&](int q)
			{
				const int bodyQ = W.p_body[q];
				if (bodyQ < 0 || p == q || bodyP == bodyQ) return;
				// component mode: a proxy of another component (or a static one) is compared as it was when the phase began -
				// the grid was built from exactly those boxes; what the others move meanwhile is k_toi_domains_end's business
				const bool foreign = DOMAIN && W.toiParent[bodyQ] != myRoot;
				if (!b2dAabbOverlap(fp, loadAabb(foreign ? W.snapFat : W.p_fat, q))) return;
				const int keyP = W.p_key[p], keyQ = W.p_key[q];
				const int lo = keyP < keyQ ? p : q, hi = keyP < keyQ ? q : p;
				const uint64_t key = ((uint64_t)(uint32_t)W.p_key[lo] << 32) | (uint32_t)W.p_key[hi];
				// does a contact already exist? (b2ContactManager.cpp:262-287) p's body is dynamic: walk its adjacency
				bool exists = false;
				const int e0 = W.adjStart[bodyP], e1 = W.adjStart[bodyP + 1];
				for (int e = e0; e < e1 && !exists; ++e) exists = C.key[W.adj[e]] == key;
				for (int c = nC0; c < s_nC && !exists; ++c) exists = C.key[c] == key;
				if (exists) return;
				if (!bodiesShouldCollide(W, W.p_body[hi], W.p_body[lo])) return;
				if (!filterShouldCollide(W.p_filter0[lo], W.p_filter1[lo], W.p_filter0[hi], W.p_filter1[hi])) return;
				if (b2dContactSwap(W.shapes[W.p_shape[lo]].type, W.shapes[W.p_shape[hi]].type) < 0) return;
				if (DOMAIN)
				{
					// the new contact ties this component to q's: neither can finish on its own (k_toi_dom_rollback)
					if (foreign && (ldFlags(&W.b_flags[bodyQ]) & BF_TYPE_MASK) != BT_STATIC)
					{
						const int d2 = W.toiDomOf[W.toiParent[bodyQ]] - 1;
						if (d2 >= 0) { W.toiDomFailed[d2] = 1; S->c.toiAnyFailed = 1; }
					}
					s_failed = 1;
				}
				const int k = atomicAdd(&s_nPairs, 1);
				if (k < TOI_PAIRS_MAX)
				{
					s_pairs[k].key = key;
					s_pairs[k].lo = lo;
					s_pairs[k].hi = hi;
				}
				else atomicOr(&s_overflow, 4);
			});
		}
		__syncthreads();
		if (DOMAIN && s_nPairs > 0)
		{
			// the creation ORDER of contacts is defined by the global event order: not for a component to decide. The
			// component is put back to the snapshot and replayed by the serial loop together with the others like it.
			break;
		}
		const int nPairs = s_nPairs < TOI_PAIRS_MAX ? s_nPairs : TOI_PAIRS_MAX;
		for (int i = tid; i < nPairs; i += LANES)
		{
			const uint64_t key = s_pairs[i].key;
			int first = 1;
			for (int j = 0; j < i; ++j) if (s_pairs[j].key == key) first = 0;
			s_pairFirst[i] = first;
		}
		__syncthreads();
		for (int i = tid; i < nPairs; i += LANES)
		{
			const uint64_t key = s_pairs[i].key;
			int rank = 0;
			for (int j = 0; j < nPairs; ++j) if (s_pairFirst[j] && s_pairs[j].key < key) ++rank;
			s_pairRank[i] = rank;
			if (s_pairFirst[i]) atomicAdd(&s_nNew, 1);
		}
		__syncthreads();
		// OnContactCreate (b2ContactManager.cpp:507-564), in (proxyLow, proxyHigh) order at the end of the array
		const int base = s_nC;
		for (int i = tid; i < nPairs; i += LANES)
		{
			if (!s_pairFirst[i]) continue;
			const int dst = base + s_pairRank[i];
			if (dst >= W.capContacts)
			{
				atomicOr(&S->c.overflow, 1);
				continue;
			}
			int pA = s_pairs[i].lo, pB = s_pairs[i].hi;
			if (b2dContactSwap(W.shapes[W.p_shape[pA]].type, W.shapes[W.p_shape[pB]].type) == 1)
			{
				const int t = pA;
				pA = pB;
				pB = t;
			}
			const int bodyA = W.p_body[pA], bodyB = W.p_body[pB];
			if (partial)
			{
				for (int side = 0; side < 2; ++side)
				{
					const int b = side ? bodyB : bodyA;
					if ((ldFlags(&W.b_flags[b]) & BF_TYPE_MASK) == BT_STATIC) continue;
					const int d2 = W.toiDomOf[W.toiParent[b]] - 1;
					if (d2 >= 0 && W.toiDomFailed[d2] == 0) atomicOr(&S->c.toiUnsafe, TOI_DOM_UNSAFE_PAIR);
				}
			}
			const bool sensor = ((W.p_filter1[pA] | W.p_filter1[pB]) & PF_SENSOR) != 0;
			uint32_t flags = CF_ENABLED | (sensor ? CF_SENSOR : 0u);
			const bool cand = isToiCandidate(W, pA, pB, bodyA, bodyB);
			if (cand) flags |= CF_TOI_CANDIDATE;
			s_pairCand[i] = cand ? 1 : 0;
			const float2 mA = W.p_mat[pA], mB = W.p_mat[pB];
			C.ids[dst] = make_int4(pA, pB, bodyA, bodyB);
			C.key[dst] = s_pairs[i].key;
			C.flags[dst] = flags;
			if (W.spatial)
			{
				// the event that created it: its place in the creation order of ALL ranks (b2d_kernels_spatial.h: k_sp_merge_tails)
				const uint64_t evKey = C.key[minIdx];
				W.spTailKey[dst] = make_int4((int)__float_as_uint(minAlpha), (int)(uint32_t)(evKey >> 32), (int)(uint32_t)evKey, 0);
				const uint32_t tA = ldFlags(&W.b_flags[bodyA]) & BF_TYPE_MASK, tB = ldFlags(&W.b_flags[bodyB]) & BF_TYPE_MASK;
				if ((tA != BT_STATIC && W.b_owner[bodyA] != (uint8_t)W.shardRank) || (tB != BT_STATIC && W.b_owner[bodyB] != (uint8_t)W.shardRank))
					atomicAdd(&S->c.spToiStraddle, 1);
			}
			C.mat[dst] = make_float4(b2dSqrt(mA.x * mB.x), mA.y > mB.y ? mA.y : mB.y, 0.0f, 1.0f);
			C.man0[dst] = make_float4(0, 0, 0, 0);
			C.man1[dst] = make_float4(0, 0, 0, 0);
			C.imp[dst] = make_float4(0, 0, 0, 0);
			C.man3[dst] = make_int4(0, 0, 0, 0);
			C.color[dst] = -1;
			C.mgr[dst] = -1;
			if (!sensor)
			{
				wake(bodyA);
				wake(bodyB);
			}
		}
		__syncthreads();
		// b2ContactManager::AddToContactArray: new TOI candidates take the next slots in creation order
		for (int i = tid; i < nPairs; i += LANES)
		{
			if (!s_pairFirst[i] || !s_pairCand[i] || base + s_pairRank[i] >= W.capContacts) continue;
			int before = 0;
			for (int j = 0; j < nPairs; ++j)
			{
				if (s_pairFirst[j] && s_pairCand[j] && s_pairRank[j] < s_pairRank[i]) ++before;
			}
			const int slot = s_toiOrder + before;
			C.mgr[base + s_pairRank[i]] = slot;
			W.toiPos2c[slot] = base + s_pairRank[i];
		}
		__syncthreads();
		if (tid == 0)
		{
			int total = s_nC + s_nNew;
			if (total > W.capContacts) total = W.capContacts;
			s_nC = total;
			s_events += 1;
			int cands = 0;
			for (int j = 0; j < nPairs; ++j) cands += (s_pairFirst[j] && s_pairCand[j] && base + s_pairRank[j] < W.capContacts) ? 1 : 0;
			s_toiOrder += cands;
		}
		__syncthreads();

		} // solid
		const int nB = solid ? s_nBodies : 0;
		TOI_T(7);
		// ---- invalidate the impacts of the displaced bodies (b2World.cpp:1005-1010) ---------------------------------
		const int nTail = s_nC - nC0;
		for (int bi = 0; bi < nB; ++bi)
		{
			const int b = s_bodies[bi];
			if ((ldFlags(&W.b_flags[b]) & BF_TYPE_MASK) != BT_DYNAMIC) continue;
			const int e0 = W.adjStart[b], e1 = W.adjStart[b + 1];
			for (int e = tid; e < (e1 - e0) + nTail; e += LANES)
			{
				int c;
				if (e < e1 - e0) c = W.adj[e0 + e];
				else
				{
					c = nC0 + (e - (e1 - e0));
					const int4 ids = C.ids[c];
					if (ids.z != b && ids.w != b) continue;
				}
				atomicAnd(&C.flags[c], ~CF_TOI);
			}
		}
		__syncthreads();
		TOI_T(8);
		// b2World::SetSubStepping (b2World.cpp:1082-1086): the call ends after its event, the step stays open - the next call
		// finds what has to be recomputed by itself (m_stepComplete = false)
		if (!DOMAIN && !partial && W.toiEventCap > 0 && s_stepCalls >= W.toiEventCap)
		{
			if (tid == 0) s_incomplete = 1;
			__syncthreads();
			break;
		}
		// ---- contacts that the next FindMinToiContact would have to (re)compute ----------------------------------------
		const int nWoken = s_nWoken < TOI_WOKEN_MAX ? s_nWoken : TOI_WOKEN_MAX;
		for (int bi = 0; bi < nB + nWoken; ++bi)
		{
			const int b = bi < nB ? s_bodies[bi] : s_woken[bi - nB];
			if ((ldFlags(&W.b_flags[b]) & BF_TYPE_MASK) == BT_STATIC) continue;
			const int e0 = W.adjStart[b], e1 = W.adjStart[b + 1];
			for (int e = tid; e < (e1 - e0) + nTail; e += LANES)
			{
				int c;
				if (e < e1 - e0) c = W.adj[e0 + e];
				else c = nC0 + (e - (e1 - e0));
				const int4 ids = C.ids[c];
				if (ids.z != b && ids.w != b) continue;
				const uint32_t flags = ldFlags(&C.flags[c]);
				if ((flags & (CF_TOI | CF_TOI_PENDING)) != 0 || !toiEligible(W, flags, ids)) continue;
				const uint32_t old = atomicOr(&C.flags[c], CF_TOI_PENDING);
				if (old & CF_TOI_PENDING) continue;
				const int k = atomicAdd(&s_nRecomp, 1);
				if (k < TOI_RECOMP_MAX) s_recomp[k] = c; else atomicOr(&s_overflow, 8);
			}
		}
		__syncthreads();
		TOI_T(9);
	}
	if (!DOMAIN && !partial && tid < 12) W.hubList[tid] = (int)s_t[tid];
	// (diagnostics: a component's phase ticks, events, TOI calls and pending contacts in 16 words behind those - tools/gpu_toi_domains_probe.py)
	if (DOMAIN && tid < 15 && 16 + 16 * (domain + 1) <= W.capContacts)
		W.hubList[16 + 16 * domain + tid] = tid < 12 ? (int)s_t[tid] : tid == 12 ? s_events : tid == 13 ? s_calls : s_nL;
#undef TOI_T

	if (tid == 0)
	{
		if (DOMAIN)
		{
			if (s_failed) { W.toiDomFailed[domain] = 1; S->c.toiAnyFailed = 1; }
			W.toiDomEvents[domain] = s_events;
			if (s_events) atomicAdd(&S->c.nToiEvents, s_events);
			if (s_calls) atomicAdd(&S->c.nToiCalls, s_calls);
			// a capacity cut inside a component is not an error yet: the serial loop gets the phase (and reports it if it
			// hits the same limit)
			const int unsafe = s_unsafe | (s_overflow ? TOI_DOM_UNSAFE_CAPACITY : 0);
			if (unsafe) atomicOr(&S->c.toiUnsafe, unsafe);
		}
		else
		{
			S->c.nContacts = s_nC;
			if (!partial) S->c.nToiList = s_nL;
			S->c.nToiEvents = (partial ? S->c.nToiEvents : 0) + s_events;
			atomicAdd(&S->c.nToiCalls, s_calls);
			S->c.toiOverflow = s_overflow | (s_logCursor > W.capToiLog && W.toiLog != nullptr ? 64 : 0);
			S->c.nToiOrder = s_toiOrder;
			S->c.nToiLog = s_logCursor < W.capToiLog ? s_logCursor : W.capToiLog;
			if (!partial) S->c.toiIncomplete = s_incomplete;
		}
	}
}

// The whole world in one persistent workgroup: the reference's serial order.
__global__ __launch_bounds__(TOI_LANES) void k_toi_loop(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	toiLoopRun<false, TOI_LANES>(W, sp, 0, 0);
}

// The components that met a new contact (or each other), replayed in the reference's global order by one workgroup.
__global__ __launch_bounds__(TOI_LANES) void k_toi_loop_partial(DW W, StepParams sp)
{
	b2dPhaseStamp(W);
	if (W.st->c.nToiPartial == 0 || W.st->c.toiUnsafe != 0) return;
	toiLoopRun<false, TOI_LANES>(W, sp, 0, 1);
}

#endif
