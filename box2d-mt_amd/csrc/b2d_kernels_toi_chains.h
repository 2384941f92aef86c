// b2d_kernels_toi_chains.h - the common case of b2World::SolveTOI in parallel.
//
// In a world without bullets / kinematic bodies every TOI candidate pairs ONE dynamic body D with a static
// body, and everything a TOI event on such a contact reads or writes is owned by D: D's sweep and velocity,
// D's contacts with static bodies (manifold update, mini island, impact recomputation) and D's proxies.
// Static bodies are only read (the reference also "advances" them, which changes nothing but their alpha0
// bookkeeping, and that only ever makes them in-sync with D again - see DESIGN.md). Events of different dynamic
// bodies therefore commute, unless one of them
//   * creates a contact (the creation ORDER across bodies is defined by the global event order), or
//   * wakes a sleeping body (its dormant contacts join the global arg-min), or
//   * involves a non-static partner (bullet, kinematic).
// k_toi_chains runs the whole event chain of each such body in its own wave; any of the conditions above raises
// Counters::toiUnsafe, in which case the host restores the snapshot taken before the chains (k_toi_snapshot /
// k_toi_restore) and lets the serial event loop (k_toi_loop) redo the phase. Nothing is approximated: either the
// chains are provably order independent, or the reference's sequential order is replayed.
#ifndef B2D_KERNELS_TOI_CHAINS_H
#define B2D_KERNELS_TOI_CHAINS_H

#include "b2d_kernels_toi.h"

#define CHAIN_LANES 64
#define CHAIN_CAND_MAX 64
#define CHAIN_EVENTS_MAX 4096
#define CHAIN_ADJ_MAX 32        // contacts of one chain body (more: serial loop)
#define TOI_GROUPS_MAX 262144   // chains per step (more: serial loop)

// unsafe bits
#define TOI_UNSAFE_PARTNER 1    // a pending impact involves a non-static partner
#define TOI_UNSAFE_WOKE 2       // a sleeping body was woken
#define TOI_UNSAFE_PAIR 4       // a moved proxy found a new pair
#define TOI_UNSAFE_CAPACITY 8   // a per-chain capacity was exceeded
#define TOI_UNSAFE_MOVED 16     // two moved proxies of different chains overlap without a contact

// One entry per dynamic body that owns at least one pending impact.
__global__ __launch_bounds__(256) void k_toi_groups_begin(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nToiList < W.capContacts ? S->c.nToiList : W.capContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[W.toiList[k]];
		const uint32_t tA = W.b_flags[ids.z] & BF_TYPE_MASK, tB = W.b_flags[ids.w] & BF_TYPE_MASK;
		int D = -1;
		if (tA == BT_DYNAMIC && tB == BT_STATIC) D = ids.z;
		else if (tB == BT_DYNAMIC && tA == BT_STATIC) D = ids.w;
		if (D < 0 || ((W.b_flags[ids.z] | W.b_flags[ids.w]) & BF_BULLET))
		{
			atomicOr(&S->c.toiUnsafe, TOI_UNSAFE_PARTNER);
			continue;
		}
		if (atomicCAS(&W.b_toiGroup[D], 0, -1) == 0)
		{
			const int g = atomicAdd(&S->c.nToiGroups, 1);
			if (g < TOI_GROUPS_MAX)
			{
				W.toiGroups[g] = D;
				W.toiGroupCount[g] = 0;
				atomicExch(&W.b_toiGroup[D], g + 1);
			}
			else atomicOr(&S->c.toiUnsafe, TOI_UNSAFE_CAPACITY);
		}
	}
}

// Everything the chains may touch, copied once (contacts go to the idle half of the double buffer).
__device__ __forceinline__ void toiSnapshotSave(const DW& W)
{
	DState* S = W.st;
	const int nC = S->c.nContacts;
	const ContactArrays& A = W.ca[S->cur];
	const ContactArrays& B = W.ca[1 - S->cur];
	const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
	if (t0 == 0)
	{
		S->c.nContactsSnap = S->c.nContacts;
		S->c.nToiOrderSnap = S->c.nToiOrder;
	}
	// (a spatially sharded world: the phase touches this rank's bodies, their proxies and contacts only - b2d_kernels_spatial.h)
	for (int i = t0; i < W.nBodies; i += stride)
	{
		if (W.spatial && (W.b_flags[i] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[i] != (uint8_t)W.shardRank) continue;
		W.snapBody[5 * (size_t)i + 0] = W.b_pos[i];
		W.snapBody[5 * (size_t)i + 1] = W.b_pos0[i];
		W.snapBody[5 * (size_t)i + 2] = W.b_vel[i];
		W.snapBody[5 * (size_t)i + 3] = W.b_xf[i];
		W.snapBody[5 * (size_t)i + 4] = make_float4(__uint_as_float(W.b_flags[i]), 0, 0, 0);
	}
	for (int p = t0; p < W.nProxies; p += stride) W.snapFat[p] = W.p_fat[p];
	for (int i = t0; i < nC; i += stride)
	{
		if (A.flags[i] & CF_FOREIGN) continue;
		B.flags[i] = A.flags[i];
		B.mat[i] = A.mat[i];
		B.man0[i] = A.man0[i];
		B.man1[i] = A.man1[i];
		B.imp[i] = A.imp[i];
		B.man3[i] = A.man3[i];
	}
}

// What k_toi_first wrote while the snapshot was being taken beside it: flags and time of impact of the TOI candidates.
__global__ __launch_bounds__(256) void k_toi_snap_cands(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& A = W.ca[S->cur];
	const ContactArrays& B = W.ca[1 - S->cur];
	const int nSlots = S->c.nToiOrder < W.capContacts ? S->c.nToiOrder : W.capContacts;
	for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nSlots; s += gridDim.x * blockDim.x)
	{
		const int i = W.toiPos2c[s];
		if (A.flags[i] & CF_FOREIGN) continue;
		B.flags[i] = A.flags[i];
		B.mat[i] = A.mat[i];
	}
}

// The contacts of every chain body, gathered in one pass over the contact array (b_toiGroup = chain index + 1).
// withSnapshot: the state the chains may touch is saved by the same launch (k_toi_snapshot, restore = 0, was one more).
__global__ __launch_bounds__(256) void k_toi_group_contacts(DW W, int withSnapshot)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (withSnapshot && S->c.nToiList != 0) toiSnapshotSave(W);
	if (S->c.toiUnsafe || S->c.nToiGroups == 0) return;
	const int nC = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nC; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		for (int side = 0; side < 2; ++side)
		{
			const int g = W.b_toiGroup[side ? ids.w : ids.z] - 1;
			if (g < 0) continue;
			const int k = atomicAdd(&W.toiGroupCount[g], 1);
			if (k < CHAIN_ADJ_MAX) W.toiGroupList[(size_t)g * CHAIN_ADJ_MAX + k] = i;
			else atomicOr(&S->c.toiUnsafe, TOI_UNSAFE_CAPACITY);
		}
	}
}

// restore = 0: toiSnapshotSave as a launch of its own; 1: everything back; 2: saved whether or not an impact is pending - the
// launch runs BESIDE k_toi_first, which finds that out (b2hip_host_phases.h: phaseToiSync; k_toi_snap_cands completes it).
__global__ __launch_bounds__(256) void k_toi_snapshot(DW W, int restore)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int nC = S->c.nContacts;
	const ContactArrays& A = W.ca[S->cur];
	const ContactArrays& B = W.ca[1 - S->cur];
	const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
	// (launched before the host knows whether any impact is pending. A call that CONTINUES a sub-stepped step has events to
	// run although nothing is pending yet - the event loop's first batch computes the impacts - so its snapshot is always due:
	// a PreSolve answer from that call's sub-step otherwise took the phase back to an EARLIER call's snapshot, materials included.)
	if (!restore && S->c.nToiList == 0 && !W.toiContinue) return;
	if (restore != 1)
	{
		toiSnapshotSave(W);
	}
	else
	{
		for (int i = t0; i < W.nBodies; i += stride)
		{
			if (W.spatial && (W.b_flags[i] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[i] != (uint8_t)W.shardRank) continue;
			W.b_pos[i] = W.snapBody[5 * (size_t)i + 0];
			W.b_pos0[i] = W.snapBody[5 * (size_t)i + 1];
			W.b_vel[i] = W.snapBody[5 * (size_t)i + 2];
			W.b_xf[i] = W.snapBody[5 * (size_t)i + 3];
			W.b_flags[i] = __float_as_uint(W.snapBody[5 * (size_t)i + 4].x);
			W.b_rowDirty[i] = 1; // (whatever the phase made of the row: the read-back compares it again)
			W.b_toiGroup[i] = 0;
		}
		for (int p = t0; p < W.nProxies; p += stride) W.p_fat[p] = W.snapFat[p];
		for (int i = t0; i < nC; i += stride)
		{
			if (A.flags[i] & CF_FOREIGN) continue;
			A.flags[i] = B.flags[i];
			A.mat[i] = B.mat[i];
			A.man0[i] = B.man0[i];
			A.man1[i] = B.man1[i];
			A.imp[i] = B.imp[i];
			A.man3[i] = B.man3[i];
		}
		if (t0 == 0)
		{
			S->c.nToiEvents = 0;
			S->c.nToiMoved = 0;
			S->c.nToiNewPairs = 0;
			S->c.nToiChainCreated = 0;
			S->c.nToiLog = 0;
			S->c.toiIncomplete = 0;
			S->c.spToiStraddle = 0;
			// (the serial replay of tied components may have created contacts)
			S->c.nContacts = S->c.nContactsSnap;
			S->c.nToiOrder = S->c.nToiOrderSnap;
		}
	}
}

// The event chain of one dynamic body: b2World::SolveTOI restricted to the contacts of D (all with static partners).
// haveGrid = 0: the hash grid was not rebuilt for this phase, so a proxy leaving its fat AABB sends the phase to the serial loop.
__device__ __forceinline__ void toiChainRun(const DW& W, const StepParams& sp, int haveGrid, int group, int* census)
{
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int lane = threadIdx.x;
	const int D = W.toiGroups[group];
	const int* adjL = W.toiGroupList + (size_t)group * CHAIN_ADJ_MAX;
	const int nAdj = W.toiGroupCount[group] < CHAIN_ADJ_MAX ? W.toiGroupCount[group] : CHAIN_ADJ_MAX;
	const float4 massD = W.b_mass[D];

	__shared__ int s_cand[CHAIN_CAND_MAX], s_sorted[CHAIN_CAND_MAX], s_info[CHAIN_CAND_MAX];
	__shared__ int s_nCand;
	__shared__ int s_contacts[B2D_MAX_TOI_CONTACTS], s_cBodyA[B2D_MAX_TOI_CONTACTS], s_cBodyB[B2D_MAX_TOI_CONTACTS], s_nK;
	__shared__ int s_bodies[B2D_MAX_TOI_BODIES], s_nB;
	__shared__ float4 s_pos[B2D_MAX_TOI_BODIES], s_vel[B2D_MAX_TOI_BODIES];
	__shared__ uint32_t s_pen;
	__shared__ int s_unsafe, s_events, s_calls, s_solid, s_minIdx;
	__shared__ float s_minAlpha;
	__shared__ int s_moved[16], s_nMoved;
	__shared__ int s_listed[16], s_nListed;
	if (lane == 0)
	{
		s_unsafe = 0;
		s_events = 0;
		s_calls = 0;
		s_nListed = 0;
	}
	__syncthreads();

	// SetAwake(true); a body that really was asleep changes the eligibility of contacts outside this chain
	auto wake = [&](int body)
	{
		const uint32_t old = atomicOr(&W.b_flags[body], BF_AWAKE);
		W.b_pos[body].w = 0.0f;
		W.b_rowDirty[body] = 1;
		if ((old & BF_AWAKE) == 0 && (old & BF_TYPE_MASK) != BT_STATIC) atomicOr(&s_unsafe, TOI_UNSAFE_WOKE);
	};
	// static partners are always in sync with D (see the header): give both sweeps D's alpha0
	auto chainToi = [&](int c, int4 ids) -> float
	{
		Sweep sA = loadSweep(W, ids.z), sB = loadSweep(W, ids.w);
		const float a0 = ids.z == D ? sA.alpha0 : sB.alpha0;
		sA.alpha0 = a0;
		sB.alpha0 = a0;
		(void)c;
		return computeToi(W, ids, sA, sB);
	};

	for (;;)
	{
		// ---- arg-min over D's candidate contacts -----------------------------------------------------------------
		uint32_t bestA = 0xffffffffu;
		unsigned long long bestK = ~0ull;
		int bestI = -1;
		for (int e = lane; e < nAdj; e += CHAIN_LANES)
		{
			const int c = adjL[e];
			const uint32_t flags = ldFlags(&C.flags[c]);
			if ((flags & CF_TOI) == 0) continue;
			const int4 ids = C.ids[c];
			if (!toiEligible(W, flags, ids)) continue;
			const uint32_t a = __float_as_uint(C.mat[c].w);
			const unsigned long long key = C.key[c];
			if (a < bestA || (a == bestA && key < bestK))
			{
				bestA = a;
				bestK = key;
				bestI = c;
			}
		}
		for (int off = 32; off > 0; off >>= 1)
		{
			const uint32_t oa = (uint32_t)__shfl_xor((int)bestA, off);
			const unsigned long long ok = ((unsigned long long)(uint32_t)__shfl_xor((int)(bestK >> 32), off) << 32) |
				(uint32_t)__shfl_xor((int)(uint32_t)bestK, off);
			const int oi = __shfl_xor(bestI, off);
			if (oa < bestA || (oa == bestA && ok < bestK))
			{
				bestA = oa;
				bestK = ok;
				bestI = oi;
			}
		}
		const int minIdx = bestI;
		const float minAlpha = bestI >= 0 ? __uint_as_float(bestA) : 1.0f;
		const uint32_t evAlphaBits = bestA;        // this event's place in the reference's global order (b2Contact::ToiLessThan)
		const unsigned long long evKey = bestK;
		if (minIdx < 0 || 1.0f - 10.0f * B2D_EPSILON < minAlpha) break;
		if (s_unsafe) break;
		if (s_events >= CHAIN_EVENTS_MAX)
		{
			if (lane == 0) s_unsafe |= TOI_UNSAFE_CAPACITY;
			break;
		}

		// ---- StepSolveTOI for (static, D) ----------------------------------------------------------------------------
		const int4 minIds = C.ids[minIdx];
		const int partner = minIds.z == D ? minIds.w : minIds.z;
		if (lane == 0)
		{
			s_nCand = 0;
			s_nMoved = 0;
			if ((ldFlags(&W.b_flags[partner]) & BF_TYPE_MASK) != BT_STATIC) s_unsafe |= TOI_UNSAFE_PARTNER;
			Sweep d = loadSweep(W, D);
			b2dSweepAdvance(d, minAlpha);
			d.c = d.c0;
			d.a = d.a0;
			const Xf xfD = b2dXfFromSweep(d.c, d.a, d.localCenter);
			const Xf xfS = loadXf(W.b_xf, partner);
			ToiUpdate u;
			u.wasTouching = (ldFlags(&C.flags[minIdx]) & CF_TOUCHING) != 0;
			toiEvaluate(W, C, minIdx, minIds, minIds.z == D ? xfD : xfS, minIds.z == D ? xfS : xfD, &u);
			toiCommitUpdate(C, minIdx, u);
			uint32_t f = ldFlags(&C.flags[minIdx]);
			const uint32_t cnt = ((f & CF_TOI_COUNT_MASK) >> CF_TOI_COUNT_SHIFT) + 1u;
			f = (f & ~(CF_TOI | CF_TOI_COUNT_MASK)) | (cnt << CF_TOI_COUNT_SHIFT);
			if (u.touching != u.wasTouching)
			{
				wake(minIds.z);
				wake(minIds.w);
			}
			if (!u.touching)
			{
				f &= ~CF_ENABLED;
				s_solid = 0;
			}
			else
			{
				s_solid = 1;
				storeAdvanced(W, D, d);
				wake(minIds.z);
				wake(minIds.w);
				s_bodies[0] = minIds.z;
				s_bodies[1] = minIds.w;
				s_nB = 2;
				s_contacts[0] = minIdx;
				s_nK = 1;
			}
			C.flags[minIdx] = f;
		}
		__syncthreads();
		if (s_unsafe) break;
		if (!s_solid) continue;

		// ---- D's other contacts with non-dynamic partners, newest first --------------------------------------------------
		for (int e = lane; e < nAdj; e += CHAIN_LANES)
		{
			const int c = adjL[e];
			if (c == minIdx) continue;
			const int4 ids = C.ids[c];
			const int other = ids.z == D ? ids.w : ids.z;
			const uint32_t fO = ldFlags(&W.b_flags[other]);
			if ((fO & BF_TYPE_MASK) == BT_DYNAMIC) continue; // neither is a bullet in this mode
			if (ldFlags(&C.flags[c]) & CF_SENSOR) continue;
			if ((fO & BF_TYPE_MASK) != BT_STATIC)
			{
				atomicOr(&s_unsafe, TOI_UNSAFE_PARTNER);
				continue;
			}
			const int k = atomicAdd(&s_nCand, 1);
			if (k < CHAIN_CAND_MAX) s_cand[k] = c; else atomicOr(&s_unsafe, TOI_UNSAFE_CAPACITY);
		}
		__syncthreads();
		if (s_unsafe) break;
		const int nCand = s_nCand;
		if (lane < nCand)
		{
			const int me = s_cand[lane];
			int rank = 0;
			for (int j = 0; j < nCand; ++j) rank += s_cand[j] > me ? 1 : 0;
			s_sorted[rank] = me;
		}
		__syncthreads();
		ToiUpdate upd;
		int myContact = -1;
		if (lane < nCand)
		{
			myContact = s_sorted[lane];
			const int4 ids = C.ids[myContact];
			const int other = ids.z == D ? ids.w : ids.z;
			const Xf xfD = loadXf(W.b_xf, D), xfO = loadXf(W.b_xf, other);
			upd.wasTouching = (ldFlags(&C.flags[myContact]) & CF_TOUCHING) != 0;
			toiEvaluate(W, C, myContact, ids, ids.z == D ? xfD : xfO, ids.z == D ? xfO : xfD, &upd);
			s_info[lane] = upd.touching ? 1 : 0;
		}
		__syncthreads();
		if (lane == 0)
		{
			int nB = s_nB, nK = s_nK;
			bool blocked = false;
			for (int r = 0; r < nCand; ++r)
			{
				if (blocked) break;
				if (nB == B2D_MAX_TOI_BODIES || nK == B2D_MAX_TOI_CONTACTS)
				{
					blocked = true;
					break;
				}
				int info = s_info[r] | 4;
				if (info & 1)
				{
					const int c = s_sorted[r];
					s_contacts[nK++] = c;
					const int4 ids = C.ids[c];
					const int other = ids.z == D ? ids.w : ids.z;
					bool bodyIn = false;
					for (int j = 0; j < nB; ++j) bodyIn = bodyIn || s_bodies[j] == other;
					if (!bodyIn) s_bodies[nB++] = other;
				}
				s_info[r] = info;
			}
			s_nB = nB;
			s_nK = nK;
		}
		__syncthreads();
		if (lane < nCand && (s_info[lane] & 4))
		{
			toiCommitUpdate(C, myContact, upd);
			if (upd.touching != upd.wasTouching)
			{
				const int4 ids = C.ids[myContact];
				wake(ids.z);
				wake(ids.w);
			}
		}
		__syncthreads();

		// ---- b2Island::SolveTOI: every constraint involves D, so the sweep is one constraint after the other -----------------
		const int nB = s_nB, nK = s_nK;
		const float h = (1.0f - minAlpha) * sp.dt;
		if (lane < nB)
		{
			const int b = s_bodies[lane];
			const float4 p = W.b_pos[b], v = W.b_vel[b];
			s_pos[lane] = make_float4(p.x, p.y, p.z, 0.0f);
			s_vel[lane] = make_float4(v.x, v.y, v.z, 0.0f);
		}
		ContactConstraint cc;
		int ci = -1, la = 0, lb = 0;
		Manifold mf;
		float4 cmat = make_float4(0, 0, 0, 0), mA4 = cmat, mB4 = cmat;
		float radiusA = 0.0f, radiusB = 0.0f;
		if (lane < nK)
		{
			ci = s_contacts[lane];
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
		__syncthreads();
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
		if (ci >= 0) initConstraint();
		__syncthreads();
		for (int it = 0; it < 20; ++it)
		{
			if (lane == 0) s_pen = 0;
			__syncthreads();
			for (int L = 0; L < nK; ++L)
			{
				if (lane == L)
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
				__syncthreads();
			}
			const float minSeparation = -__uint_as_float(s_pen);
			__syncthreads();
			if (minSeparation >= -1.5f * B2D_LINEAR_SLOP) break;
		}
		// leap of faith: only D moves (the static seed keeps c0 = c)
		if (lane < 2 && s_bodies[lane] == D)
		{
			const float4 p = s_pos[lane];
			W.b_pos0[D] = make_float4(p.x, p.y, p.z, minAlpha);
			W.b_rowDirty[D] = 1;
		}
		if (ci >= 0) initConstraint();
		__syncthreads();
		for (int it = 0; it < sp.velIters; ++it)
		{
			for (int L = 0; L < nK; ++L)
			{
				if (lane == L)
				{
					BodyVel vA, vB;
					const float4 va = s_vel[la], vb = s_vel[lb];
					vA.v = v2(va.x, va.y); vA.w = va.z;
					vB.v = v2(vb.x, vb.y); vB.w = vb.z;
					b2dSolveVelocity(&cc, &vA, &vB);
					s_vel[la] = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
					s_vel[lb] = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
				}
				__syncthreads();
			}
		}
		if (lane < nB && s_bodies[lane] == D)
		{
			const float4 p = s_pos[lane], v = s_vel[lane];
			V2 c = v2(p.x, p.y), vv = v2(v.x, v.y);
			float a = p.z, w = v.z;
			b2dIntegratePosition(&c, &a, &vv, &w, h);
			const float sleepTime = W.b_pos[D].w;
			W.b_pos[D] = make_float4(c.x, c.y, a, sleepTime);
			W.b_vel[D] = make_float4(vv.x, vv.y, w, 0.0f);
			W.b_rowDirty[D] = 1;
			const Xf xf = b2dXfFromSweep(c, a, v2(massD.z, massD.w));
			W.b_xf[D] = make_float4(xf.p.x, xf.p.y, xf.q.s, xf.q.c);
		}
		__syncthreads();

		// ---- SynchronizeFixtures(D) + pair search -----------------------------------------------------------------------------
		if (lane == 0)
		{
			const float4 p0 = W.b_pos0[D];
			const Xf xf1 = b2dXfFromSweep(v2(p0.x, p0.y), p0.z, v2(massD.z, massD.w));
			const Xf xf2 = loadXf(W.b_xf, D);
			for (int p = W.b_proxyHead[D]; p >= 0; p = W.p_next[p])
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
				if (s_nMoved < 16) s_moved[s_nMoved++] = p; else s_unsafe |= TOI_UNSAFE_CAPACITY;
				// world-wide list (once per proxy) with the hull of every box the proxy has had in this phase, for k_toi_chains_end
				bool listed = false;
				for (int j = 0; j < s_nListed; ++j) listed = listed || s_listed[j] == p;
				if (!listed)
				{
					if (s_nListed < 16) s_listed[s_nListed++] = p; else s_unsafe |= TOI_UNSAFE_CAPACITY;
					const int k = atomicAdd(&S->c.nToiMoved, 1);
					// (list and hulls are read by the workgroup of this launch that finishes last: past the L2, b2dStoreAgent*)
					if (k < TOI_MOVED_MAX) b2dStoreAgentI(&W.toiMoved[k], p); else s_unsafe |= TOI_UNSAFE_CAPACITY;
					b2dStoreAgent4(&W.toiHull[p], W.snapFat[p]);
				}
				const float4 hcur = b2dLoadAgent4(&W.toiHull[p]);
				b2dStoreAgent4(&W.toiHull[p], make_float4(fminf(hcur.x, f.lo.x), fminf(hcur.y, f.lo.y), fmaxf(hcur.z, f.hi.x), fmaxf(hcur.w, f.hi.y)));
			}
		}
		__syncthreads();
		const int nMoved = s_nMoved;
		if (nMoved > 0 && !haveGrid)
		{
			if (lane == 0) s_unsafe |= TOI_UNSAFE_PAIR;
		}
		else if (nMoved > 0)
		{
			// against the fat AABBs as they were before the chains started (other chains move theirs concurrently; the grid was
			// built from exactly those); moved-vs-moved across chains is checked afterwards by k_toi_chains_end
			for (int mI = 0; mI < nMoved; ++mI)
			{
				const int p = s_moved[mI];
				const AABB fp = loadAabb(W.p_fat, p);
				toiForEachCandidate(W, fp, lane, CHAIN_LANES, W.toiMoved, 0, [&](int q)
				{
					const int bodyQ = W.p_body[q];
					if (bodyQ < 0 || bodyQ == D) return;
					if (!b2dAabbOverlap(fp, loadAabb(W.snapFat, q))) return;
					const int keyP = W.p_key[p], keyQ = W.p_key[q];
					const int lo = keyP < keyQ ? p : q, hi = keyP < keyQ ? q : p;
					const uint64_t key = ((uint64_t)(uint32_t)W.p_key[lo] << 32) | (uint32_t)W.p_key[hi];
					bool exists = false;
					for (int e = 0; e < nAdj && !exists; ++e) exists = C.key[adjL[e]] == key;
					if (exists) return;
					if (!bodiesShouldCollide(W, W.p_body[hi], W.p_body[lo])) return;
					if (!filterShouldCollide(W.p_filter0[lo], W.p_filter1[lo], W.p_filter0[hi], W.p_filter1[hi])) return;
					if (b2dContactSwap(W.shapes[W.p_shape[lo]].type, W.shapes[W.p_shape[hi]].type) < 0) return;
					// A new pair. The reference creates its contact now (b2World.cpp:1015, FindNewContacts after every sub-step),
					// and creation ORDER is global - but a contact between D and another moving, non-bullet, awake body takes no
					// part in the rest of the phase: sub-step islands leave such contacts out (b2World.cpp:905-912), it is no TOI
					// candidate, nobody updates it before the next Collide. So it is only noted, with the key of this event; the
					// chains' close-out (toiChainsEnd) creates what was noted in event order. Anything else stays a case for
					// the serial loop: a static / kinematic / bullet partner (D's later events would use the contact), a sleeper.
					const uint32_t fq = ldFlags(&W.b_flags[bodyQ]);
					const bool inert = (fq & BF_TYPE_MASK) == BT_DYNAMIC && (fq & (BF_BULLET | BF_AWAKE | BF_ACTIVE)) == (BF_AWAKE | BF_ACTIVE);
					if (!inert || W.noChainCreate)
					{
						atomicOr(&s_unsafe, TOI_UNSAFE_PAIR);
						return;
					}
					const int k = atomicAdd(&S->c.nToiNewPairs, 1);
					if (k >= TOI_NEWPAIR_MAX)
					{
						atomicOr(&s_unsafe, TOI_UNSAFE_CAPACITY);
						return;
					}
					int* ent = W.toiNew + 8 * (size_t)k;
					b2dStoreAgentI(&ent[0], (int)evAlphaBits);
					b2dStoreAgentI(&ent[1], (int)(uint32_t)(evKey >> 32));
					b2dStoreAgentI(&ent[2], (int)(uint32_t)evKey);
					b2dStoreAgentI(&ent[3], lo);
					b2dStoreAgentI(&ent[4], hi);
				});
			}
		}
		__syncthreads();
		if (lane == 0) s_events += 1;

		// ---- invalidate + recompute the impacts of D's contacts ------------------------------------------------------------------
		for (int e = lane; e < nAdj; e += CHAIN_LANES) atomicAnd(&C.flags[adjL[e]], ~CF_TOI);
		__syncthreads();
		if (s_unsafe) break;
		for (int e = lane; e < nAdj; e += CHAIN_LANES)
		{
			const int c = adjL[e];
			const uint32_t flags = ldFlags(&C.flags[c]);
			const int4 ids = C.ids[c];
			if (!toiEligible(W, flags, ids)) continue;
			if ((ldFlags(&W.b_flags[ids.z == D ? ids.w : ids.z]) & BF_TYPE_MASK) != BT_STATIC)
			{
				atomicOr(&s_unsafe, TOI_UNSAFE_PARTNER);
				continue;
			}
			const float alpha = chainToi(c, ids);
			atomicAdd(&s_calls, 1);
			float4 mat = C.mat[c];
			mat.w = alpha;
			C.mat[c] = mat;
			C.flags[c] = flags | CF_TOI;
		}
		__syncthreads();
	}

	__syncthreads();
	if (lane == 0)
	{
		// (events and calls: summed by the workgroup over its chains and carried by its arrival - k_toi_chains)
		census[0] += s_events;
		census[1] += s_calls;
		if (s_unsafe) atomicOr(&S->c.toiUnsafe, s_unsafe);
	}
}

// The launch does not know the number of chains (the host no longer waits for the census of k_toi_first): a fixed grid
// of waves takes them in turn.
// Two proxies moved by different chains may have come to overlap without either chain seeing it. Run by the last
// workgroup of k_toi_chains to finish (it was a launch of its own behind it).
__device__ __forceinline__ void toiChainsEnd(const DW& W)
{
	DState* S = W.st;
	const int nMovedAll = b2dLoadAgentI(&S->c.nToiMoved);
	const int n = nMovedAll < TOI_MOVED_MAX ? nMovedAll : TOI_MOVED_MAX;
	const int nG = S->c.nToiGroups < TOI_GROUPS_MAX ? S->c.nToiGroups : TOI_GROUPS_MAX;
	for (int i = threadIdx.x; i < n && n >= 2 && !b2dLoadAgentI(&S->c.toiUnsafe); i += blockDim.x)
	{
		const int p = b2dLoadAgentI(&W.toiMoved[i]);
		const float4 hp4 = b2dLoadAgent4(&W.toiHull[p]); // every box p has had in this phase: covers momentary overlaps too
		AABB fp;
		fp.lo = v2(hp4.x, hp4.y);
		fp.hi = v2(hp4.z, hp4.w);
		for (int j = i + 1; j < n; ++j)
		{
			const int q = b2dLoadAgentI(&W.toiMoved[j]);
			if (q == p || W.p_body[p] == W.p_body[q]) continue;
			const float4 hq4 = b2dLoadAgent4(&W.toiHull[q]);
			AABB fq;
			fq.lo = v2(hq4.x, hq4.y);
			fq.hi = v2(hq4.z, hq4.w);
			if (!b2dAabbOverlap(fp, fq)) continue;
			// an existing contact makes the overlap harmless: look it up on p's (dynamic) body
			const int keyP = W.p_key[p], keyQ = W.p_key[q];
			const int lo = keyP < keyQ ? p : q, hi = keyP < keyQ ? q : p;
			const uint64_t key = ((uint64_t)(uint32_t)W.p_key[lo] << 32) | (uint32_t)W.p_key[hi];
			const ContactArrays& C = W.ca[S->cur];
			const int g = W.b_toiGroup[W.p_body[p]] - 1;
			bool exists = false;
			if (g >= 0)
			{
				const int cnt = W.toiGroupCount[g] < CHAIN_ADJ_MAX ? W.toiGroupCount[g] : CHAIN_ADJ_MAX;
				for (int e = 0; e < cnt && !exists; ++e) exists = C.key[W.toiGroupList[(size_t)g * CHAIN_ADJ_MAX + e]] == key;
			}
			if (!exists) atomicOr(&S->c.toiUnsafe, TOI_UNSAFE_MOVED);
		}
	}
	__syncthreads();
	for (int g = threadIdx.x; g < nG; g += blockDim.x) W.b_toiGroup[W.toiGroups[g]] = 0;
	// ---- the contacts the chains noted (pairs of two moving bodies), created in the reference's order: by event (alpha,
	// proxy ids of the event's contact), within an event by the pair's proxy ids (b2ContactManager::FindNewContacts sorts its
	// pairs); a pair noted twice - by a later event of the same body, or from the other side - exists from its first event on.
	// (Two chains that both moved towards each other were caught above: their hulls overlap without a contact.)
	const int nNoted = b2dLoadAgentI(&S->c.nToiNewPairs);
	if (nNoted > 0 && nNoted <= TOI_NEWPAIR_MAX && b2dLoadAgentI(&S->c.toiUnsafe) == 0)
	{
		const ContactArrays& C = W.ca[S->cur];
		__shared__ int s_first[TOI_NEWPAIR_MAX], s_rank[TOI_NEWPAIR_MAX];
		__shared__ int s_created;
		if (threadIdx.x == 0) s_created = 0;
		auto entry = [&](int i, uint32_t* alpha, unsigned long long* ev, unsigned long long* pair, int* lo, int* hi)
		{
			const int* ent = W.toiNew + 8 * (size_t)i;
			*alpha = (uint32_t)b2dLoadAgentI(&ent[0]);
			*ev = ((unsigned long long)(uint32_t)b2dLoadAgentI(&ent[1]) << 32) | (uint32_t)b2dLoadAgentI(&ent[2]);
			*lo = b2dLoadAgentI(&ent[3]);
			*hi = b2dLoadAgentI(&ent[4]);
			*pair = ((unsigned long long)(uint32_t)W.p_key[*lo] << 32) | (uint32_t)W.p_key[*hi];
		};
		auto before = [](uint32_t a1, unsigned long long e1, unsigned long long p1, uint32_t a2, unsigned long long e2, unsigned long long p2)
		{
			if (a1 != a2) return a1 < a2;
			if (e1 != e2) return e1 < e2;
			return p1 < p2;
		};
		__syncthreads();
		for (int i = threadIdx.x; i < nNoted; i += blockDim.x)
		{
			uint32_t ai; unsigned long long ei, pi; int loi, hii;
			entry(i, &ai, &ei, &pi, &loi, &hii);
			int first = 1;
			for (int j = 0; j < nNoted; ++j)
			{
				if (j == i) continue;
				uint32_t aj; unsigned long long ej, pj; int loj, hij;
				entry(j, &aj, &ej, &pj, &loj, &hij);
				if (pj != pi) continue;
				// the same pair: the note of the earlier event stands (identical notes: the lower index)
				if (before(aj, ej, pj, ai, ei, pi) || (aj == ai && ej == ei && j < i)) first = 0;
			}
			s_first[i] = first;
			if (first) atomicAdd(&s_created, 1);
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nNoted; i += blockDim.x)
		{
			if (!s_first[i]) continue;
			uint32_t ai; unsigned long long ei, pi; int loi, hii;
			entry(i, &ai, &ei, &pi, &loi, &hii);
			int rank = 0;
			for (int j = 0; j < nNoted; ++j)
			{
				if (!s_first[j] || j == i) continue;
				uint32_t aj; unsigned long long ej, pj; int loj, hij;
				entry(j, &aj, &ej, &pj, &loj, &hij);
				if (before(aj, ej, pj, ai, ei, pi)) ++rank;
			}
			s_rank[i] = rank;
		}
		__syncthreads();
		const int base = S->c.nContacts;
		if (base + s_created > W.capContacts)
		{
			// (b2hip_step_end undoes the phase from its snapshot, grows the array and runs the phase again)
			if (threadIdx.x == 0) atomicOr(&S->c.overflow, 1);
		}
		else
		{
			for (int i = threadIdx.x; i < nNoted; i += blockDim.x)
			{
				if (!s_first[i]) continue;
				uint32_t ai; unsigned long long ei, pi; int pA, pB;
				entry(i, &ai, &ei, &pi, &pA, &pB);
				// OnContactCreate (b2ContactManager.cpp:507-564), as k_create_contacts does it
				if (b2dContactSwap(W.shapes[W.p_shape[pA]].type, W.shapes[W.p_shape[pB]].type) == 1)
				{
					const int t = pA;
					pA = pB;
					pB = t;
				}
				const int dst = base + s_rank[i];
				const int bodyA = W.p_body[pA], bodyB = W.p_body[pB];
				const bool sensor = ((W.p_filter1[pA] | W.p_filter1[pB]) & PF_SENSOR) != 0;
				const float2 mA = W.p_mat[pA], mB = W.p_mat[pB];
				C.ids[dst] = make_int4(pA, pB, bodyA, bodyB);
				C.key[dst] = pi;
				if (W.spatial)
				{
					W.spTailKey[dst] = make_int4((int)ai, (int)(uint32_t)(ei >> 32), (int)(uint32_t)ei, 0);
					if (W.b_owner[bodyA] != (uint8_t)W.shardRank || W.b_owner[bodyB] != (uint8_t)W.shardRank) atomicAdd(&S->c.spToiStraddle, 1);
				}
				C.flags[dst] = CF_ENABLED | (sensor ? CF_SENSOR : 0u); // (two dynamic bodies, no bullet: not a TOI candidate)
				C.mat[dst] = make_float4(b2dSqrt(mA.x * mB.x), mA.y > mB.y ? mA.y : mB.y, 0.0f, 1.0f);
				C.man0[dst] = make_float4(0, 0, 0, 0);
				C.man1[dst] = make_float4(0, 0, 0, 0);
				C.imp[dst] = make_float4(0, 0, 0, 0);
				C.man3[dst] = make_int4(0, 0, 0, 0);
				C.color[dst] = -1;
				C.mgr[dst] = -1;
				if (!sensor)
				{
					// SetAwake(true) on both (:525-529): both are awake (the chains note no pair with a sleeper); the timers restart
					W.b_pos[bodyA].w = 0.0f;
					W.b_pos[bodyB].w = 0.0f;
					W.b_rowDirty[bodyA] = 1;
					W.b_rowDirty[bodyB] = 1;
				}
			}
			__syncthreads();
			if (threadIdx.x == 0)
			{
				S->c.nContacts = base + s_created;
				S->c.nToiChainCreated = s_created;
			}
		}
	}
}

__global__ __launch_bounds__(CHAIN_LANES) void k_toi_chains(DW W, StepParams sp, int haveGrid)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	__shared__ int s_chainCensus[2];
	if (threadIdx.x == 0) { s_chainCensus[0] = 0; s_chainCensus[1] = 0; }
	__syncthreads();
	if ((S->c.toiUnsafe & (TOI_UNSAFE_PARTNER | TOI_UNSAFE_CAPACITY)) == 0)
	{
		const int nGroups = S->c.nToiGroups < TOI_GROUPS_MAX ? S->c.nToiGroups : TOI_GROUPS_MAX;
		for (int group = blockIdx.x; group < nGroups; group += gridDim.x)
		{
			toiChainRun(W, sp, haveGrid, group, s_chainCensus);
			__syncthreads();
		}
	}
	// the last workgroup to finish adds the census the arrivals carried (two atomics per chain on the two counters before)
	// and closes the phase
	unsigned events = 0u, calls = 0u;
	if (b2dLastBlockArrive(W, ARRIVE_CHAINS, (unsigned)s_chainCensus[0], (unsigned)s_chainCensus[1], &events, &calls))
	{
		if (threadIdx.x == 0)
		{
			if (events) atomicAdd(&S->c.nToiEvents, (int)events);
			if (calls) atomicAdd(&S->c.nToiCalls, (int)calls);
		}
		toiChainsEnd(W);
	}
}

// ---- fused front of the pair update: (hash table of contact keys + spatial grid) cleared, then built -----------------
// Same bodies as k_ht_clear + k_grid_clear(force 0) and k_ht_build + k_grid_count(force 0): independent arrays, one
// launch each instead of two.
__global__ __launch_bounds__(256) void k_bp_clear(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		S->c.nLargeProxies = 0;
		S->c.nLargeMoves = 0;
		S->c.nPairs = 0;
		S->c.nNewContacts = 0;
		S->c.nMovesSeen = S->c.nMoves;
		S->c.gridFresh = S->c.nMoves != 0 ? 1 : 0; // (this update builds the grid from every proxy's box: k_bp_build, k_grid_fill)
	}
	if (blockIdx.x == 0 && threadIdx.x < 32) S->c.candRounds[threadIdx.x] = 0;
	if (S->c.nMoves == 0) return;
	const uint32_t stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
	// widest proxy that should still go through the grid (see gridLimit): complete when this kernel ends
	{
		const float cap = 4.0f * W.cellSize;
		float ext = 0.0f;
		for (uint32_t p = t0; p < (uint32_t)W.nProxies; p += stride)
		{
			if (W.p_body[p] < 0) continue;
			const float4 a = W.p_fat[p];
			const float e = fmaxf(a.z - a.x, a.w - a.y);
			if (e <= cap && e > ext) ext = e;
		}
		// (a positive float's bits order like the float: one offer per workgroup, b2d_wave.h)
		blockAtomicMaxIfAbove((int*)&S->c.cellExtBits, ext > 0.0f ? (int)__float_as_uint(ext) : 0);
	}
	// (16 bytes per lane and store: the tables are powers of two of at least 64 entries, 256-byte aligned - 4-byte stores
	// were 93 us for the 40 MB of a million-proxy world)
	{
		const int4 z = make_int4(0, 0, 0, 0);
		int4* gc = (int4*)W.gridCount;
		int4* gu = (int4*)W.gridCursor;
		for (uint32_t i = t0, n4 = (W.gridMask + 1u) >> 2; i < n4; i += stride)
		{
			gc[i] = z;
			gu[i] = z;
		}
		int4* hk = (int4*)W.ht_keys;
		for (uint32_t i = t0, n2 = (htLiveMask(W) + 1u) >> 1; i < n2; i += stride) hk[i] = z; // (the part this update uses)
	}
}

__global__ __launch_bounds__(256) void k_bp_build(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	if (S->c.nMoves == 0) return;
	const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
	const int nC = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	// (the pair search asks the set only about pairs of this rank's bodies: another rank's contacts stay out - tryEmitPair)
	for (int i = t0; i < nC; i += stride) if ((C.flags[i] & CF_FOREIGN) == 0) htInsert(W, C.key[i] + 1ull);
	const int n = W.nProxies;
	for (int p = t0; p < n; p += stride)
	{
		if (W.p_body[p] < 0) continue;
		const float4 a = W.p_fat[p];
		if (proxyIsLarge(W, a))
		{
			const int k = atomicAdd(&S->c.nLargeProxies, 1);
			W.largeProxies[k] = p;
		}
		else
		{
			int ix, iy;
			proxyCell(W, a, &ix, &iy);
			atomicAdd(&W.gridCount[cellHash(ix, iy, W.gridMask)], 1);
		}
	}
}

#endif
