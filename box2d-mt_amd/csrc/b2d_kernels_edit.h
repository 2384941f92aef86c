// b2d_kernels_edit.h - world edits between steps that touch the contact array (b2World::DestroyBody, b2Body::DestroyFixture,
// SetBullet, b2Fixture::SetSensor / SetThickShape / SetFilterData / Refilter).
//
// The host queues one op per call (include/b2hip.h: b2hip_destroy_body ...); the next step (or the next read of the
// contacts) applies them in call order with ONE launch of one workgroup. An op concerns the contacts of one body or one
// fixture; the reference walks that body's contact-edge list, which is newest first = descending index in the creation-
// ordered contact array, and wherever its result depends on that order - the slots of the TOI partition
// (b2ContactManager::RemoveFromContactArray / RecalculateToiCandidacy, b2ContactManager.cpp:566-640, 688-714) - one lane
// replays the list in that order.
//   EDIT_DESTROY_BODY / EDIT_DESTROY_FIXTURE   b2ContactManager::Destroy on every contact of the body / fixture
//                                              (b2World.cpp:617-625, b2Body.cpp:262-275, b2ContactManager.cpp:94-160)
//   EDIT_RECALC_BODY / EDIT_RECALC_FIXTURE     b2ContactManager::RecalculateToiCandidacy (:566-640)
//   EDIT_REFILTER_FIXTURE                      b2Fixture::Refilter's contact flags (b2Fixture.cpp:187-210)
//   EDIT_SENSOR_FIXTURE                        the contacts' cached "either fixture is a sensor" bit (b2Contact.cpp:187)
#ifndef B2D_KERNELS_EDIT_H
#define B2D_KERNELS_EDIT_H

#include "b2d_kernels_broadphase.h"

enum
{
	EDIT_DESTROY_BODY = 1,
	EDIT_DESTROY_FIXTURE = 2,
	EDIT_RECALC_BODY = 3,
	EDIT_RECALC_FIXTURE = 4,
	EDIT_REFILTER_FIXTURE = 5,
	EDIT_SENSOR_FIXTURE = 6
};
#define EDIT_LIST_MAX 8192 // contacts per pass (LDS list); an op takes as many passes as it needs

__global__ __launch_bounds__(1024) void k_apply_edits(DW W, const int2* ops, int nOps)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	__shared__ int s_list[EDIT_LIST_MAX];
	__shared__ int s_n, s_waveCount[16];
	const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
	for (int o = 0; o < nOps; ++o)
	{
		const int kind = ops[o].x, id = ops[o].y;
		const bool byBody = kind == EDIT_DESTROY_BODY || kind == EDIT_RECALC_BODY;
		const int n = S->c.nContacts;
		// The matching contacts are listed newest first and worked off in passes of at most EDIT_LIST_MAX (a ground body under
		// a large pile, the Tumbler's container: the reference has no limit): a pass ends where the next chunk of 1024 might
		// not fit any more, the op is applied to the listed contacts, and the next pass goes on below - the newest-first order
		// of the whole walk is kept.
		for (int passTop = n - 1; passTop >= 0;)
		{
		if (t == 0) s_n = 0;
		__syncthreads();
		int hi = passTop;
		// chunks of 1024 from the top, each lane one index, ordered append
		for (; hi >= 0 && s_n + 1024 <= EDIT_LIST_MAX; hi -= 1024)
		{
			const int i = hi - t;
			bool match = false;
			if (i >= 0)
			{
				const int4 ids = C.ids[i];
				match = (C.flags[i] & CF_DESTROY) == 0 && (byBody ? (ids.z == id || ids.w == id) : (ids.x == id || ids.y == id));
			}
			const unsigned long long m = __ballot(match);
			if (lane == 0) s_waveCount[wave] = __popcll(m);
			__syncthreads();
			int base = s_n;
			for (int k = 0; k < wave; ++k) base += s_waveCount[k];
			const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
			if (match) s_list[pos] = i;
			__syncthreads();
			if (t == 0)
			{
				int tot = 0;
				for (int k = 0; k < 16; ++k) tot += s_waveCount[k];
				s_n += tot;
			}
			__syncthreads();
		}
		passTop = hi; // (uniform: every lane ran the same chunks)
		const int cnt = s_n;
		if (kind == EDIT_REFILTER_FIXTURE)
		{
			for (int k = t; k < cnt; k += 1024) C.flags[s_list[k]] |= CF_FILTER;
		}
		else if (kind == EDIT_SENSOR_FIXTURE)
		{
			for (int k = t; k < cnt; k += 1024)
			{
				const int i = s_list[k];
				const int4 ids = C.ids[i];
				const bool sensor = ((W.p_filter1[ids.x] | W.p_filter1[ids.y]) & PF_SENSOR) != 0;
				C.flags[i] = (C.flags[i] & ~CF_SENSOR) | (sensor ? CF_SENSOR : 0u);
			}
		}
		else if (kind == EDIT_DESTROY_BODY || kind == EDIT_DESTROY_FIXTURE)
		{
			// everything that does not depend on the order, in parallel ...
			for (int k = t; k < cnt; k += 1024)
			{
				const int i = s_list[k];
				const uint32_t flags = C.flags[i];
				const int4 ids = C.ids[i];
				if (W.eventsOn && (flags & CF_REPORTED))
				{
					// b2ContactManager::Destroy (:104-107): a touching contact ends when it is destroyed
					const int e = atomicAdd(&S->c.nEvents, 1);
					if (e < W.capContacts)
					{
						W.evKey[e] = C.key[i];
						W.evInfo[e] = make_int4(ids.x, ids.y, 1, -1);
					}
				}
				// b2Contact::Destroy (b2Contact.cpp:100-113): wake both bodies if the manifold had points
				// (at once, as SetAwake(true) does: the flag decides which contacts this step's Collide updates)
				if (C.man3[i].w > 0 && (flags & CF_SENSOR) == 0)
				{
					atomicOr(&W.b_flags[ids.z], BF_AWAKE);
					atomicOr(&W.b_flags[ids.w], BF_AWAKE);
					W.b_pos[ids.z].w = 0.0f;
					W.b_pos[ids.w].w = 0.0f;
				}
				C.flags[i] = flags | CF_DESTROY;
			}
			__syncthreads();
			// ... the TOI partition by one lane, in the reference's order (RemoveFromContactArray, :688-714)
			if (t == 0)
			{
				int nToi = S->c.nToiOrder;
				for (int k = 0; k < cnt; ++k)
				{
					const int i = s_list[k];
					const int slot = C.mgr[i];
					if (slot < 0) continue;
					const int last = nToi - 1;
					const int moved = W.toiPos2c[last];
					W.toiPos2c[slot] = moved;
					C.mgr[moved] = slot;
					C.mgr[i] = -1;
					nToi = last;
				}
				S->c.nToiOrder = nToi;
				S->c.nDestroy += cnt;
			}
		}
		else
		{
			// RecalculateToiCandidacy (:590-640), contact after contact
			if (t == 0)
			{
				int nToi = S->c.nToiOrder;
				for (int k = 0; k < cnt; ++k)
				{
					const int i = s_list[k];
					const int4 ids = C.ids[i];
					uint32_t flags = C.flags[i];
					const bool cand = isToiCandidate(W, ids.x, ids.y, ids.z, ids.w);
					if (cand == ((flags & CF_TOI_CANDIDATE) != 0)) continue;
					flags = (flags ^ CF_TOI_CANDIDATE) & ~CF_TOI_STATE_MASK;
					C.flags[i] = flags;
					float4 mat = C.mat[i];
					mat.w = 1.0f;
					C.mat[i] = mat;
					if (cand)
					{
						C.mgr[i] = nToi;
						W.toiPos2c[nToi] = i;
						++nToi;
					}
					else
					{
						const int slot = C.mgr[i];
						const int last = nToi - 1;
						const int moved = W.toiPos2c[last];
						W.toiPos2c[slot] = moved;
						C.mgr[moved] = slot;
						C.mgr[i] = -1;
						nToi = last;
					}
				}
				S->c.nToiOrder = nToi;
			}
		}
		__syncthreads();
		} // passes
	}
}

// After destroy ops: keep flags for the stable compaction that follows (the tail of phaseCollide)
__global__ __launch_bounds__(256) void k_edit_keepflags(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) W.keepFlag[i] = (C.flags[i] & CF_DESTROY) ? 0 : 1;
}

__global__ void k_edit_finish(DW W)
{
	b2dPhaseStamp(W);
	// the per-step destroy census belongs to Collide
	W.st->c.nDestroy = 0;
}

// b2World::ShiftOrigin (b2World.cpp:1862-1887): every body's transform and sweep, every proxy's fat AABB
// (b2DynamicTree::ShiftOrigin, b2DynamicTree.cpp:768-776) and the world-space anchors joints keep (b2MouseJoint::ShiftOrigin,
// b2PulleyJoint::ShiftOrigin) move by -newOrigin; plain float subtractions, the same bits as the reference's.
__global__ __launch_bounds__(256) void k_shift_origin(DW W, float ox, float oy)
{
	const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
	for (int i = t0; i < W.nBodies; i += stride)
	{
		float4 xf = W.b_xf[i], p = W.b_pos[i], p0 = W.b_pos0[i];
		xf.x -= ox; xf.y -= oy;
		p.x -= ox; p.y -= oy;
		p0.x -= ox; p0.y -= oy;
		W.b_xf[i] = xf;
		W.b_pos[i] = p;
		W.b_pos0[i] = p0;
	}
	for (int q = t0; q < W.nProxies; q += stride)
	{
		float4 f = W.p_fat[q];
		f.x -= ox; f.y -= oy; f.z -= ox; f.w -= oy;
		W.p_fat[q] = f;
	}
	for (int j = t0; j < W.nJoints; j += stride)
	{
		JointRec& jn = W.joints[j];
		if (jn.type == B2D_JOINT_MOUSE) { jn.targetA.x -= ox; jn.targetA.y -= oy; }
		else if (jn.type == B2D_JOINT_PULLEY) { jn.groundAnchorA.x -= ox; jn.groundAnchorA.y -= oy; jn.s1 -= ox; jn.s2 -= oy; }
	}
}

#endif
