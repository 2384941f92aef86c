// b2d_kernels_spatial.h - one world over the GPUs of a node by SPATIAL OWNERSHIP (round 4).
//
// Every rank keeps the replicated tables section 8e names - the body table (pose, velocity, flags), the proxy table (fat
// AABBs) - and the STRUCTURE of the contact array (which contacts exist, in which order, their TOI slots): ids mean the same
// on every rank, so every order the reference's results depend on (proxy keys, creation order, m_nonStaticBodies, TOI slots)
// is the unsharded world's. What is partitioned is the WORK and the contact CONTENT:
//   * every non-static body has ONE owner (DW::b_owner, the same table on every rank), handed out by spatial cell when the
//     world is sharded (b2hip_shard_spatial: strips of equal body count along x);
//   * invariant: a connected component of the graph {non-static bodies, EXISTING contacts between them, joints} is owned by
//     one rank. Islands (touching contacts) and everything a TOI event reaches lie inside such a component, so a rank
//     evaluates manifolds, builds islands, solves, and runs the TOI events of its own bodies only, from complete inputs;
//   * a contact whose bodies belong to another rank is FOREIGN here (CF_FOREIGN): it exists, is destroyed when its fat
//     AABBs part, keeps its slot - and its manifold, impulses and touching bit are not maintained (nobody reads them).
// Per step a rank receives (all-gather, RCCL over xGMI or the caller's collective):
//   E1 after Solve + SynchronizeFixtures: the state rows of the bodies the other ranks stepped and the fat AABBs those moved
//      (k_sp_export_state / k_sp_import_state) - north_star's "all-gather of boundary body velocities", for all moved bodies;
//   E2 inside FindNewContacts: the new pairs every rank found for ITS moved proxies (against the replicated proxy table), so
//      that all ranks create the same contacts in the same proxy-key order (k_sp_export_pairs / k_sp_import_pairs);
//   E3 when a new contact joins components of different owners ("straddles"): every rank labels the components of the
//      replicated structure itself, the component goes to the owner that holds most of its bodies (no vote needed: same
//      inputs, same answer), and the old owners ship the CONTENT of its contacts and joints to the new one
//      (k_sp_resolve_* , k_sp_export_content / k_sp_import_content) - this is where bodies migrate;
//   E4 after SolveTOI: the rows and fat AABBs of the bodies TOI events advanced.
// Reference: the reference partitions exactly these phases over its threads - b2CollideTask b2World.cpp:100,
// b2BroadphaseSyncFixturesTask :120, b2BroadphaseFindNewContactsTask :142, islands :1236-1241, 1322-1330.
#ifndef B2D_KERNELS_SPATIAL_H
#define B2D_KERNELS_SPATIAL_H

#include "b2d_kernels_toi_domains.h"

#define SP_HEADER_WORDS 8      // slab header: [0] bodies, [1] proxies, [2] pairs, [3] contacts, [4] joints, [5] flags, [6..7] -
#define SP_BODY_WORDS 17       // id, pos xyzw, pos0 xyzw, vel xyz, awake, xf xyzw
#define SP_PROXY_WORDS 5       // id, fat AABB
#define SP_TOI_PROXY_WORDS 9   // E4: id, fat AABB, the fat AABB the TOI phase began with (where the proxy has been: k_sp_toi_conflicts)
#define SP_PAIR_WORDS 4        // key hi, key lo, proxy lo, proxy hi
#define SP_CONTENT_WORDS 24    // contact index, flags, mat xyzw, man0 xyzw, man1 xyzw, imp xyzw, man3 xyzw, colour(-1), -
#define SP_JOINT_WORDS 6       // as SHARD_JOINT_WORDS
#define SP_RESOLVE_MAX 65536   // components one resolution can merge

__device__ __forceinline__ bool spSameBits(float4 a, float4 b)
{
	return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
		__float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

__device__ __forceinline__ bool spForeignBody(const DW& W, int body)
{
	return W.spatial && (W.b_flags[body] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[body] != (uint8_t)W.shardRank;
}

// CF_FOREIGN of every contact from the owner table, and the census of STRADDLING contacts (two non-static bodies of
// different owners: a freshly created contact across a boundary). After every change of the structure or of the owners.
__global__ __launch_bounds__(256) void k_sp_flag_contacts(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nContacts;
	const ContactArrays& C = W.ca[S->cur];
	int straddle = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC, nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
		const int oA = nsA ? W.b_owner[ids.z] : -1, oB = nsB ? W.b_owner[ids.w] : -1;
		const bool foreign = (nsA && oA != W.shardRank) || (nsB && oB != W.shardRank);
		const uint32_t f = C.flags[i];
		const uint32_t g = foreign ? (f | CF_FOREIGN) : (f & ~CF_FOREIGN);
		if (g != f) C.flags[i] = g;
		if (nsA && nsB && oA != oB)
		{
			const int k = atomicAdd(&S->c.nStraddle, 1);
			if (k < W.capStraddle) W.spStraddle[k] = i;
			++straddle;
		}
	}
	(void)straddle;
}

// ---- E1 / E4: state rows and fat AABBs --------------------------------------------------------------------------------------
// mode 0: after Solve + SynchronizeFixtures - the bodies this rank stepped (BF_ISLAND: in an island, free bodies included) and
// the proxies its SynchronizeFixtures moved (the move buffer); mode 1: after SolveTOI - the bodies and proxies this rank's TOI
// phase changed: whatever differs from the snapshot the phase started from. Records carry their ids; the header their counts.
__global__ __launch_bounds__(256) void k_sp_export_state(DW W, int* out, int mode, int capBodies, int capProxies)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	int* hdr = out;
	int* ob = out + SP_HEADER_WORDS;
	int* op = ob + (size_t)capBodies * SP_BODY_WORDS;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		const uint32_t f = W.b_flags[i];
		if ((f & BF_TYPE_MASK) == BT_STATIC || W.b_owner[i] != (uint8_t)W.shardRank) continue;
		const float4 p0 = W.b_pos0[i];
		const float4 p = W.b_pos[i], v = W.b_vel[i], xf = W.b_xf[i];
		const uint8_t awakeNow = (f & BF_AWAKE) ? 1 : 0;
		bool rowDue = true;
		if (mode == 0)
		{
			if ((f & BF_ISLAND) == 0) continue;
			// lean: the others' work needs this body's awake bit (the Collide rule for their copies of our contacts), not its row
			rowDue = W.spFullRows || awakeNow != W.spAwake[i];
		}
		else
		{
			// what this rank's TOI phase changed: whatever differs from the snapshot the phase started from (k_toi_snapshot)
			const float4 s0 = W.snapBody[5 * (size_t)i + 0], s1 = W.snapBody[5 * (size_t)i + 1], s2 = W.snapBody[5 * (size_t)i + 2], s3 = W.snapBody[5 * (size_t)i + 3];
			const uint32_t sf = __float_as_uint(W.snapBody[5 * (size_t)i + 4].x);
			const bool same = spSameBits(p, s0) && spSameBits(p0, s1) && spSameBits(v, s2) && spSameBits(xf, s3) && ((f ^ sf) & BF_AWAKE) == 0;
			if (same) continue;
			rowDue = W.spFullRows || awakeNow != W.spAwake[i];
		}
		if (rowDue)
		{
		const int k = atomicAdd(&hdr[0], 1);
		if (k < capBodies) { // (the header keeps the true count: the hosts grow the slab and repeat; k_sp_mark_sent when it has arrived)
		int* o = ob + (size_t)k * SP_BODY_WORDS;
		o[0] = i;
		o[1] = __float_as_int(p.x); o[2] = __float_as_int(p.y); o[3] = __float_as_int(p.z); o[4] = __float_as_int(p.w);
		o[5] = __float_as_int(p0.x); o[6] = __float_as_int(p0.y); o[7] = __float_as_int(p0.z); o[8] = __float_as_int(p0.w);
		o[9] = __float_as_int(v.x); o[10] = __float_as_int(v.y); o[11] = __float_as_int(v.z);
		o[12] = (f & BF_AWAKE) ? 1 : 0;
		o[13] = __float_as_int(xf.x); o[14] = __float_as_int(xf.y); o[15] = __float_as_int(xf.z); o[16] = __float_as_int(xf.w);
		}
		}
		if (mode == 1)
		{
			for (int q = W.b_proxyHead[i]; q >= 0; q = W.p_next[q])
			{
				const float4 fat = W.p_fat[q];
				if (spSameBits(fat, W.snapFat[q])) continue;
				const int kp = atomicAdd(&hdr[1], 1);
				if (kp >= capProxies) { continue; }
				int* r = op + (size_t)kp * SP_TOI_PROXY_WORDS;
				const float4 was = W.snapFat[q];
				r[0] = q;
				r[1] = __float_as_int(fat.x); r[2] = __float_as_int(fat.y); r[3] = __float_as_int(fat.z); r[4] = __float_as_int(fat.w);
				r[5] = __float_as_int(was.x); r[6] = __float_as_int(was.y); r[7] = __float_as_int(was.z); r[8] = __float_as_int(was.w);
			}
		}
	}
	if (mode == 0)
	{
		const int nm = S->c.nMoves < W.capMoves ? S->c.nMoves : W.capMoves;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nm; k += gridDim.x * blockDim.x)
		{
			const int q = W.moveBuf[k];
			if (q < 0 || W.p_body[q] < 0 || spForeignBody(W, W.p_body[q])) continue;
			if ((W.b_flags[W.p_body[q]] & BF_TYPE_MASK) == BT_STATIC) continue; // (a moved static proxy is a host edit: every rank made it)
			const int kp = atomicAdd(&hdr[1], 1);
			if (kp >= capProxies) { atomicOr(&S->c.overflow, 512); continue; }
			int* r = op + (size_t)kp * SP_PROXY_WORDS;
			const float4 fat = W.p_fat[q];
			r[0] = q;
			r[1] = __float_as_int(fat.x); r[2] = __float_as_int(fat.y); r[3] = __float_as_int(fat.z); r[4] = __float_as_int(fat.w);
		}
	}
}

// (behind an exchange that fitted: the other ranks hold the awake bit of every body this rank owns)
__global__ __launch_bounds__(256) void k_sp_mark_sent(DW W)
{
	b2dPhaseStamp(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		const uint32_t f = W.b_flags[i];
		if ((f & BF_TYPE_MASK) == BT_STATIC || W.b_owner[i] != (uint8_t)W.shardRank) continue;
		W.spAwake[i] = (f & BF_AWAKE) ? 1 : 0;
	}
}

// markSent: the lean exchange's k_sp_mark_sent in the same launch (this rank's OWN bodies; the records below are other ranks').
// sendHdr: the header of this rank's send slab, wiped for the next exchange's export (the all-gather has read it: it ran
// before this launch on the stream) - a fill launch less per exchange.
__global__ __launch_bounds__(256) void k_sp_import_state(DW W, const int* in, size_t strideWords, int capBodies, int proxyWords, int markSent, int* sendHdr)
{
	b2dPhaseStamp(W);
	if (blockIdx.x == 0 && threadIdx.x < SP_HEADER_WORDS) sendHdr[threadIdx.x] = 0;
	if (blockIdx.x == 0 && threadIdx.x == 0) W.st->c.gridFresh = 0; // (other ranks' boxes arrive: the grid is stale until a pair update builds it again)
	if (markSent)
	{
		for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
		{
			const uint32_t f = W.b_flags[i];
			if ((f & BF_TYPE_MASK) == BT_STATIC || W.b_owner[i] != (uint8_t)W.shardRank) continue;
			W.spAwake[i] = (f & BF_AWAKE) ? 1 : 0;
		}
	}
	for (int r = 0; r < W.shardCount; ++r)
	{
		if (r == W.shardRank) continue;
		const int* slab = in + (size_t)r * strideWords;
		const int nB = slab[0] < capBodies ? slab[0] : capBodies, nP = slab[1];
		const int* ib = slab + SP_HEADER_WORDS;
		const int* ip = ib + (size_t)capBodies * SP_BODY_WORDS;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nB; k += gridDim.x * blockDim.x)
		{
			const int* o = ib + (size_t)k * SP_BODY_WORDS;
			const int i = o[0];
			if (i < 0 || i >= W.nBodies) continue;
			W.b_pos[i] = make_float4(__int_as_float(o[1]), __int_as_float(o[2]), __int_as_float(o[3]), __int_as_float(o[4]));
			// (alpha0 = 0: what the owner's sweep holds once its step has ended - k_end_step resets the sweeps TOI events advanced)
			W.b_pos0[i] = make_float4(__int_as_float(o[5]), __int_as_float(o[6]), __int_as_float(o[7]), 0.0f);
			W.b_vel[i] = make_float4(__int_as_float(o[9]), __int_as_float(o[10]), __int_as_float(o[11]), 0.0f);
			W.b_xf[i] = make_float4(__int_as_float(o[13]), __int_as_float(o[14]), __int_as_float(o[15]), __int_as_float(o[16]));
			uint32_t f = W.b_flags[i];
			if (o[12]) f |= BF_AWAKE;
			else
			{
				// the island fell asleep (b2Body::SetAwake(false), b2Body.h:704-717)
				f &= ~BF_AWAKE;
				W.b_force[i] = make_float4(0, 0, 0, 0);
			}
			W.b_flags[i] = f;
			W.spAwake[i] = o[12] ? 1 : 0;
		}
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nP; k += gridDim.x * blockDim.x)
		{
			const int* o = ip + (size_t)k * proxyWords;
			const int q = o[0];
			if (q < 0 || q >= W.nProxies) continue;
			W.p_fat[q] = make_float4(__int_as_float(o[1]), __int_as_float(o[2]), __int_as_float(o[3]), __int_as_float(o[4]));
		}
	}
}

// ---- E2: the new pairs ------------------------------------------------------------------------------------------------------
// out: header + this rank's candidate pairs (what k_find_pairs_* stored for ITS moved proxies, filters applied).
__global__ __launch_bounds__(256) void k_sp_export_pairs(DW W, int* out, int capPairsSlab)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nPairs < W.capPairs ? S->c.nPairs : W.capPairs;
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		out[2] = S->c.nPairs;                       // (the true count, also when it exceeds either buffer: the host grows and repeats)
		out[5] = S->c.overflow & 3;
	}
	int* op = out + SP_HEADER_WORDS;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n && i < capPairsSlab; i += gridDim.x * blockDim.x)
	{
		const uint64_t key = W.pairKey[i];
		const int2 pr = W.pairProxy[i];
		{
			// (a pair that will join bodies of different owners: the hosts learn from the headers whether a resolution is due)
			const int bA = W.p_body[pr.x], bB = W.p_body[pr.y];
			if ((W.b_flags[bA] & BF_TYPE_MASK) != BT_STATIC && (W.b_flags[bB] & BF_TYPE_MASK) != BT_STATIC && W.b_owner[bA] != W.b_owner[bB]) atomicAdd(&out[6], 1);
		}
		int* o = op + (size_t)i * SP_PAIR_WORDS;
		o[0] = (int)(uint32_t)(key >> 32);
		o[1] = (int)(uint32_t)key;
		o[2] = pr.x;
		o[3] = pr.y;
	}
}

// The other ranks' pairs behind our own (duplicates - a pair both of whose proxies moved, on two ranks - fall to the first /
// rank kernels like duplicates inside one rank's list do).
__global__ __launch_bounds__(256) void k_sp_import_pairs(DW W, const int* in, size_t strideWords, int capPairsSlab)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	__shared__ int s_base[SHARD_MAX_RANKS + 1];
	if (threadIdx.x == 0)
	{
		int at = S->c.nPairs < W.capPairs ? S->c.nPairs : W.capPairs;
		for (int r = 0; r < W.shardCount; ++r)
		{
			s_base[r] = at;
			if (r != W.shardRank) at += in[(size_t)r * strideWords + 2] < capPairsSlab ? in[(size_t)r * strideWords + 2] : capPairsSlab;
		}
		s_base[W.shardCount] = at;
	}
	__syncthreads();
	for (int r = 0; r < W.shardCount; ++r)
	{
		if (r == W.shardRank) continue;
		const int* slab = in + (size_t)r * strideWords;
		const int n = slab[2] < capPairsSlab ? slab[2] : capPairsSlab;
		const int* ip = slab + SP_HEADER_WORDS;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		{
			const int dst = s_base[r] + k;
			if (dst >= W.capPairs) { atomicOr(&S->c.overflow, 2); continue; }
			const int* o = ip + (size_t)k * SP_PAIR_WORDS;
			W.pairKey[dst] = ((uint64_t)(uint32_t)o[0] << 32) | (uint32_t)o[1];
			W.pairProxy[dst] = make_int2(o[2], o[3]);
		}
	}
}
// (a launch of its own behind the import: every workgroup of the import reads the old count)
__global__ void k_sp_import_pairs_commit(DW W, const int* in, size_t strideWords, int capPairsSlab, int* sendHdr)
{
	DState* S = W.st;
	for (int k = 0; k < SP_HEADER_WORDS; ++k) sendHdr[k] = 0; // (for the next exchange's export: k_sp_import_state)
	int at = S->c.nPairs;
	for (int r = 0; r < W.shardCount; ++r)
		if (r != W.shardRank) at += in[(size_t)r * strideWords + 2] < capPairsSlab ? in[(size_t)r * strideWords + 2] : capPairsSlab;
	S->c.nPairs = at;
}

// ---- E3: a contact that joins components of different owners ------------------------------------------------------------------
// Every rank runs this on the same replicated structure and owner table and gets the same answer - no vote. Components are
// labelled by k_toi_dom_init / k_toi_dom_union / k_sp_union_joints / k_toi_dom_flatten (DW::toiParent: scratch of the TOI
// phase, which starts over from it). Then, for the components that hold a straddling contact:
//   k_sp_resolve_mark   one table row per such component (DW::toiDomOf[label] = row + 1)
//   k_sp_resolve_count  bodies per owner in each row
//   k_sp_resolve_pick   the new owner: whoever holds most bodies, the lowest rank among equals
//   k_sp_export_content what this rank loses: content of the contacts / joints of bodies it owns whose component goes elsewhere
//   k_sp_apply_owners   DW::b_owner of every body of those components
//   k_sp_import_content what this rank gains
__global__ __launch_bounds__(256) void k_sp_union_joints(DW W)
{
	b2dPhaseStamp(W);
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x)
	{
		const JointRec& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		const bool nsA = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC, nsB = (W.b_flags[jn.bodyB] & BF_TYPE_MASK) != BT_STATIC;
		if (nsA && nsB) ufUnion(W.toiParent, jn.bodyA, jn.bodyB);
		if (jn.type == B2D_JOINT_GEAR)
		{
			// (a gear joint couples the bodies of its two joints as well: b2GearJoint.cpp:40-130)
			const GearRec& g = W.gears[jn.enableLimit];
			const int others[2] = { g.bodyC, g.bodyD };
			for (int k = 0; k < 2; ++k)
			{
				const int o = others[k];
				if (o < 0 || (W.b_flags[o] & BF_TYPE_MASK) == BT_STATIC) continue;
				if (nsA) ufUnion(W.toiParent, jn.bodyA, o);
				else if (nsB) ufUnion(W.toiParent, jn.bodyB, o);
			}
		}
	}
}

// (joints between bodies of different owners straddle as well: a joint created after the world was sharded)
__global__ __launch_bounds__(256) void k_sp_flag_joints(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x)
	{
		const JointRec& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		const bool nsA = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC, nsB = (W.b_flags[jn.bodyB] & BF_TYPE_MASK) != BT_STATIC;
		if (nsA && nsB && W.b_owner[jn.bodyA] != W.b_owner[jn.bodyB]) atomicAdd(&S->c.nStraddleJoints, 1);
	}
}

__global__ __launch_bounds__(256) void k_sp_resolve_mark(DW W, const int2* virt, int nVirt)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int n = S->c.nStraddle < W.capStraddle ? S->c.nStraddle : W.capStraddle;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n + W.nJoints + nVirt; k += gridDim.x * blockDim.x)
	{
		int body = -1;
		if (k < n) body = C.ids[W.spStraddle[k]].z;
		else if (k >= n + W.nJoints) body = virt[k - n - W.nJoints].x;
		else
		{
			const JointRec& jn = W.joints[k - n];
			if (jn.type == B2D_JOINT_DEAD) continue;
			const bool nsA = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC, nsB = (W.b_flags[jn.bodyB] & BF_TYPE_MASK) != BT_STATIC;
			if (!(nsA && nsB) || W.b_owner[jn.bodyA] == W.b_owner[jn.bodyB]) continue;
			body = jn.bodyA;
		}
		const int label = W.toiParent[body];
		if (atomicCAS(&W.toiDomOf[label], 0, -1) == 0)
		{
			const int row = atomicAdd(&S->c.nResolve, 1);
			if (row < SP_RESOLVE_MAX)
			{
				for (int r = 0; r < SHARD_MAX_RANKS; ++r) W.spCount[(size_t)row * SHARD_MAX_RANKS + r] = 0;
				W.spTarget[row] = -1;
				atomicExch(&W.toiDomOf[label], row + 1);
			}
			else atomicOr(&S->c.overflow, 1024);
		}
	}
}

__global__ __launch_bounds__(256) void k_sp_resolve_count(DW W)
{
	b2dPhaseStamp(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if ((W.b_flags[i] & BF_TYPE_MASK) == BT_STATIC) continue;
		const int row = W.toiDomOf[W.toiParent[i]] - 1;
		if (row < 0 || row >= SP_RESOLVE_MAX) continue;
		atomicAdd(&W.spCount[(size_t)row * SHARD_MAX_RANKS + W.b_owner[i]], 1);
	}
}

__global__ __launch_bounds__(256) void k_sp_resolve_pick(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nResolve < SP_RESOLVE_MAX ? S->c.nResolve : SP_RESOLVE_MAX;
	for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x)
	{
		int best = 0;
		for (int r = 1; r < W.shardCount; ++r)
			if (W.spCount[(size_t)row * SHARD_MAX_RANKS + r] > W.spCount[(size_t)row * SHARD_MAX_RANKS + best]) best = r;
		W.spTarget[row] = best;
	}
}

// The new owner of `body` (its own when its component is not being resolved).
__device__ __forceinline__ int spNewOwner(const DW& W, int body)
{
	const int row = W.toiDomOf[W.toiParent[body]] - 1;
	return row >= 0 && row < SP_RESOLVE_MAX ? W.spTarget[row] : (int)W.b_owner[body];
}

// What every rank will ship (every rank counts ALL ranks' records: the hosts size the collective without talking).
__global__ __launch_bounds__(256) void k_sp_content_census(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int n = S->c.nContacts;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
		const bool nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
		if (!nsA && !nsB) continue;
		if (nsA && nsB && W.b_owner[ids.z] != W.b_owner[ids.w]) continue; // (a straddling contact: nobody has content for it yet)
		const int body = nsA ? ids.z : ids.w;
		const int from = W.b_owner[body];
		if (spNewOwner(W, body) != from) atomicAdd(&S->c.spContacts[from], 1);
	}
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x)
	{
		const JointRec& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		const bool nsA = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC;
		const bool nsB = (W.b_flags[jn.bodyB] & BF_TYPE_MASK) != BT_STATIC;
		if (!nsA && !nsB) continue;
		if (nsA && nsB && W.b_owner[jn.bodyA] != W.b_owner[jn.bodyB]) continue;
		const int body = nsA ? jn.bodyA : jn.bodyB;
		const int from = W.b_owner[body];
		if (spNewOwner(W, body) != from) atomicAdd(&S->c.spJoints[from], 1);
	}
	// (the rows of the bodies that leave a rank: with the lean exchange nobody else holds them)
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if ((W.b_flags[i] & BF_TYPE_MASK) == BT_STATIC) continue;
		const int from = W.b_owner[i];
		if (spNewOwner(W, i) != from) atomicAdd(&S->c.spMigBodies[from], 1);
	}
}

__global__ __launch_bounds__(256) void k_sp_export_content(DW W, int* out, int capContacts, int capJoints)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int n = S->c.nContacts;
	const int me = W.shardRank;
	int* oc = out + SP_HEADER_WORDS;
	int* oj = oc + (size_t)capContacts * SP_CONTENT_WORDS;
	int* ob = oj + (size_t)capJoints * SP_JOINT_WORDS;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		const uint32_t f = W.b_flags[i];
		if ((f & BF_TYPE_MASK) == BT_STATIC || W.b_owner[i] != me || spNewOwner(W, i) == me) continue;
		const int k = atomicAdd(&out[0], 1);
		int* o = ob + (size_t)k * SP_BODY_WORDS;
		const float4 p = W.b_pos[i], p0 = W.b_pos0[i], v = W.b_vel[i], xf = W.b_xf[i];
		o[0] = i;
		o[1] = __float_as_int(p.x); o[2] = __float_as_int(p.y); o[3] = __float_as_int(p.z); o[4] = __float_as_int(p.w);
		o[5] = __float_as_int(p0.x); o[6] = __float_as_int(p0.y); o[7] = __float_as_int(p0.z); o[8] = __float_as_int(p0.w);
		o[9] = __float_as_int(v.x); o[10] = __float_as_int(v.y); o[11] = __float_as_int(v.z);
		o[12] = (f & BF_AWAKE) ? 1 : 0;
		o[13] = __float_as_int(xf.x); o[14] = __float_as_int(xf.y); o[15] = __float_as_int(xf.z); o[16] = __float_as_int(xf.w);
	}
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int4 ids = C.ids[i];
		const bool nsA = (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC;
		const bool nsB = (W.b_flags[ids.w] & BF_TYPE_MASK) != BT_STATIC;
		if (!nsA && !nsB) continue;
		if (nsA && nsB && W.b_owner[ids.z] != W.b_owner[ids.w]) continue;
		const int body = nsA ? ids.z : ids.w;
		if (W.b_owner[body] != me || spNewOwner(W, body) == me) continue;
		const int k = atomicAdd(&out[3], 1);
		if (k >= capContacts) { atomicOr(&S->c.overflow, 512); continue; }
		int* o = oc + (size_t)k * SP_CONTENT_WORDS;
		const float4 mat = C.mat[i], m0 = C.man0[i], m1 = C.man1[i], im = C.imp[i];
		const int4 m3 = C.man3[i];
		o[0] = i;
		o[1] = (int)C.flags[i];
		o[2] = __float_as_int(mat.x); o[3] = __float_as_int(mat.y); o[4] = __float_as_int(mat.z); o[5] = __float_as_int(mat.w);
		o[6] = __float_as_int(m0.x); o[7] = __float_as_int(m0.y); o[8] = __float_as_int(m0.z); o[9] = __float_as_int(m0.w);
		o[10] = __float_as_int(m1.x); o[11] = __float_as_int(m1.y); o[12] = __float_as_int(m1.z); o[13] = __float_as_int(m1.w);
		o[14] = __float_as_int(im.x); o[15] = __float_as_int(im.y); o[16] = __float_as_int(im.z); o[17] = __float_as_int(im.w);
		o[18] = m3.x; o[19] = m3.y; o[20] = m3.z; o[21] = m3.w;
		o[22] = spNewOwner(W, body);
		o[23] = 0;
	}
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W.nJoints; j += gridDim.x * blockDim.x)
	{
		const JointRec& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		const bool nsA = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC;
		const bool nsB = (W.b_flags[jn.bodyB] & BF_TYPE_MASK) != BT_STATIC;
		if (!nsA && !nsB) continue;
		if (nsA && nsB && W.b_owner[jn.bodyA] != W.b_owner[jn.bodyB]) continue;
		const int body = nsA ? jn.bodyA : jn.bodyB;
		if (W.b_owner[body] != me || spNewOwner(W, body) == me) continue;
		const int k = atomicAdd(&out[4], 1);
		int* o = oj + (size_t)k * SP_JOINT_WORDS;
		o[0] = j;
		o[1] = __float_as_int(jn.type == B2D_JOINT_GEAR ? W.gears[jn.enableLimit].impulse : jn.impulseX);
		o[2] = __float_as_int(jn.impulseY);
		o[3] = __float_as_int(jn.impulseZ);
		o[4] = __float_as_int(jn.motorImpulse);
		o[5] = jn.limitState;
	}
}

// (before the import: it tells a record for this rank by the body's NEW owner)
__global__ __launch_bounds__(256) void k_sp_apply_owners(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if ((W.b_flags[i] & BF_TYPE_MASK) == BT_STATIC) continue;
		const int to = spNewOwner(W, i);
		if (to != (int)W.b_owner[i])
		{
			W.spNewOwner[i] = (uint8_t)to;
			atomicAdd(&S->c.nMigrated, 1);
			// a body that arrives has no home block here (the block partition of the large islands is this rank's own)
			if (to == W.shardRank) { W.b_blk1[i] = 0; }
		}
		else W.spNewOwner[i] = W.b_owner[i];
	}
}
__global__ __launch_bounds__(256) void k_sp_commit_owners(DW W)
{
	b2dPhaseStamp(W);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		W.b_owner[i] = W.spNewOwner[i];
		W.toiDomOf[i] = 0;
	}
}

__global__ __launch_bounds__(256) void k_sp_import_content(DW W, const int* in, size_t strideWords, int capContacts, int capJoints)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	for (int r = 0; r < W.shardCount; ++r)
	{
		if (r == W.shardRank) continue;
		const int* slab = in + (size_t)r * strideWords;
		const int nC = slab[3] < capContacts ? slab[3] : capContacts, nJ = slab[4], nB = slab[0];
		const int* ic = slab + SP_HEADER_WORDS;
		const int* ij = ic + (size_t)capContacts * SP_CONTENT_WORDS;
		const int* ib = ij + (size_t)capJoints * SP_JOINT_WORDS;
		// the rows of the bodies that changed owner (every rank takes them: whoever gains the body needs them, the others' copies
		// are simply brought up to date)
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nB; k += gridDim.x * blockDim.x)
		{
			const int* o = ib + (size_t)k * SP_BODY_WORDS;
			const int i = o[0];
			if (i < 0 || i >= W.nBodies) continue;
			W.b_pos[i] = make_float4(__int_as_float(o[1]), __int_as_float(o[2]), __int_as_float(o[3]), __int_as_float(o[4]));
			W.b_pos0[i] = make_float4(__int_as_float(o[5]), __int_as_float(o[6]), __int_as_float(o[7]), __int_as_float(o[8]));
			W.b_vel[i] = make_float4(__int_as_float(o[9]), __int_as_float(o[10]), __int_as_float(o[11]), 0.0f);
			W.b_xf[i] = make_float4(__int_as_float(o[13]), __int_as_float(o[14]), __int_as_float(o[15]), __int_as_float(o[16]));
			uint32_t f = W.b_flags[i];
			f = o[12] ? (f | BF_AWAKE) : (f & ~BF_AWAKE);
			W.b_flags[i] = f;
			W.spAwake[i] = o[12] ? 1 : 0;
		}
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nC; k += gridDim.x * blockDim.x)
		{
			const int* o = ic + (size_t)k * SP_CONTENT_WORDS;
			const int i = o[0];
			if (i < 0 || i >= S->c.nContacts || o[22] != W.shardRank) continue;
			// (the structural bits are this rank's own - equal on every rank - and CF_FOREIGN is set afresh by k_sp_flag_contacts)
			C.flags[i] = (uint32_t)o[1];
			C.mat[i] = make_float4(__int_as_float(o[2]), __int_as_float(o[3]), __int_as_float(o[4]), __int_as_float(o[5]));
			C.man0[i] = make_float4(__int_as_float(o[6]), __int_as_float(o[7]), __int_as_float(o[8]), __int_as_float(o[9]));
			C.man1[i] = make_float4(__int_as_float(o[10]), __int_as_float(o[11]), __int_as_float(o[12]), __int_as_float(o[13]));
			C.imp[i] = make_float4(__int_as_float(o[14]), __int_as_float(o[15]), __int_as_float(o[16]), __int_as_float(o[17]));
			C.man3[i] = make_int4(o[18], o[19], o[20], o[21]);
			C.color[i] = -1;
		}
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nJ; k += gridDim.x * blockDim.x)
		{
			const int* o = ij + (size_t)k * SP_JOINT_WORDS;
			const int j = o[0];
			if (j < 0 || j >= W.nJoints) continue;
			JointRec& jn = W.joints[j];
			if (jn.type == B2D_JOINT_GEAR) W.gears[jn.enableLimit].impulse = __int_as_float(o[1]);
			else jn.impulseX = __int_as_float(o[1]);
			jn.impulseY = __int_as_float(o[2]);
			jn.impulseZ = __int_as_float(o[3]);
			jn.motorImpulse = __int_as_float(o[4]);
			jn.limitState = o[5];
		}
	}
}

// ---- contacts created inside the TOI phases ----------------------------------------------------------------------------------
// Every rank's event loop appended the contacts ITS events created behind the array all ranks shared when the phase began
// (index `base` on): in the order of its own events, which is the reference's order restricted to its bodies. The reference's
// array holds the contacts of all events in the order of the events - (alpha, proxy ids of the event's contact,
// b2Contact::ToiLessThan), inside an event by the pair's proxy ids (b2ContactManager::FindNewContacts sorts them) - so the
// tails are merged in that order on every rank: a rank's own contacts move (content and all), the others' are created
// (structure only), and the TOI slots (b2Contact::m_managerIndex: AddToContactArray hands them out in creation order) are
// dealt again in the merged order. The descriptors travel with E4: SP_TAIL_WORDS per contact.
#define SP_TAIL_WORDS 6        // alpha bits, event key hi, lo, proxy lo, proxy hi (key order), index in the creating rank's tail
#define SP_TAIL_MAX 4096       // contacts all ranks together may create inside one TOI phase

__global__ __launch_bounds__(256) void k_sp_export_tail(DW W, int* out, int* hdr, int base, int capTail, int chains, int* virtCount)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) *virtCount = 0; // (k_sp_tail_pairs counts into it, behind the all-gather)
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int n = S->c.nContacts - base;
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		// what the host would have read back before the exchange: it reads every rank's header behind it anyway
		hdr[4] = S->c.nToiMoved;
		hdr[5] = n;
		hdr[6] = S->c.spToiStraddle;
		hdr[7] = (chains ? S->c.toiUnsafe : 0) | ((S->c.overflow & 1) ? 0x40000000 : 0);
	}
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n && k < capTail; k += gridDim.x * blockDim.x)
	{
		const int4 key = W.spTailKey[base + k];
		const int4 ids = C.ids[base + k];
		const bool swap = W.p_key[ids.x] > W.p_key[ids.y];
		int* o = out + (size_t)k * SP_TAIL_WORDS;
		o[0] = key.x; o[1] = key.y; o[2] = key.z;
		o[3] = swap ? ids.y : ids.x;
		o[4] = swap ? ids.x : ids.y;
		o[5] = k;
	}
}

// A TOI event that reached over an ownership boundary (a contact created inside a sub-step with a body of another rank): the
// phase is taken back on every rank, the components of such pairs merge as if the contact existed (the pairs are "virtual
// edges" of the resolution: DW::spVirt), and the phase runs again - the pair then lies inside one rank.
// ... and two proxies that events of DIFFERENT ranks moved so that they came to overlap: in the unsharded world the later of
// the two events finds the pair and creates its contact inside the phase; here neither rank saw the other's move. Every
// rank has every rank's E4 records, so every rank finds the same conflicts: boxes = where a proxy has been in the phase (the
// hull of the box it began with and the one it ended with). A conflict is a virtual edge like a straddling tail contact
// (no contact can exist between the two bodies yet: they would have one owner).
__global__ __launch_bounds__(256) void k_sp_tail_pairs(DW W, const int* in, size_t strideWords, size_t tailAt, int capBodies, int capProxies, int capTail, int2* out, int* nOut)
{
	b2dPhaseStamp(W);
	{
		// cross-rank overlaps of moved proxies: record (r, k) against every record of the ranks behind r
		int start[SHARD_MAX_RANKS + 1];
		int at = 0;
		for (int r = 0; r < W.shardCount; ++r) { start[r] = at; const int n = in[(size_t)r * strideWords + 1]; at += n < capProxies ? n : capProxies; }
		start[W.shardCount] = at;
		const size_t proxAt = SP_HEADER_WORDS + (size_t)capBodies * SP_BODY_WORDS;
		auto rec = [&](int e, int* r) -> const int*
		{
			int rr = 0;
			while (e >= start[rr + 1]) ++rr;
			*r = rr;
			return in + (size_t)rr * strideWords + proxAt + (size_t)(e - start[rr]) * SP_TOI_PROXY_WORDS;
		};
		for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < at; e += gridDim.x * blockDim.x)
		{
			int re;
			const int* a = rec(e, &re);
			const float ax0 = fminf(__int_as_float(a[1]), __int_as_float(a[5])), ay0 = fminf(__int_as_float(a[2]), __int_as_float(a[6]));
			const float ax1 = fmaxf(__int_as_float(a[3]), __int_as_float(a[7])), ay1 = fmaxf(__int_as_float(a[4]), __int_as_float(a[8]));
			const int bodyA = W.p_body[a[0]];
			for (int j = start[re + 1]; j < at; ++j)
			{
				int rj;
				const int* b = rec(j, &rj);
				const float bx0 = fminf(__int_as_float(b[1]), __int_as_float(b[5])), by0 = fminf(__int_as_float(b[2]), __int_as_float(b[6]));
				const float bx1 = fmaxf(__int_as_float(b[3]), __int_as_float(b[7])), by1 = fmaxf(__int_as_float(b[4]), __int_as_float(b[8]));
				// b2TestOverlap (b2Collision.h:273-286): separated iff a gap is positive
				if (bx0 - ax1 > 0.0f || by0 - ay1 > 0.0f || ax0 - bx1 > 0.0f || ay0 - by1 > 0.0f) continue;
				const int bodyB = W.p_body[b[0]];
				if (bodyA == bodyB || W.b_owner[bodyA] == W.b_owner[bodyB]) continue;
				const int k = atomicAdd(nOut, 1);
				if (k < SP_TAIL_MAX) out[k] = make_int2(bodyA, bodyB);
			}
		}
	}
	for (int r = 0; r < W.shardCount; ++r)
	{
		const int* slab = in + (size_t)r * strideWords;
		// (header word 5 is the rank's TRUE count - the host grows capTail from it and repeats the exchange; the slab holds
		// capTail descriptors, what lies behind them is the next rank's slab: ADVICE round 4)
		const int n = slab[5] < capTail ? slab[5] : capTail;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		{
			const int* o = slab + tailAt + (size_t)k * SP_TAIL_WORDS;
			if ((unsigned)o[3] >= (unsigned)W.nProxies || (unsigned)o[4] >= (unsigned)W.nProxies) continue; // (ids from another rank's slab: checked like the import kernels')
			const int bA = W.p_body[o[3]], bB = W.p_body[o[4]];
			const bool nsA = (W.b_flags[bA] & BF_TYPE_MASK) != BT_STATIC, nsB = (W.b_flags[bB] & BF_TYPE_MASK) != BT_STATIC;
			if (!nsA || !nsB || W.b_owner[bA] == W.b_owner[bB]) continue;
			const int slot = atomicAdd(nOut, 1);
			if (slot < SP_TAIL_MAX) out[slot] = make_int2(bA, bB);
		}
	}
}

__global__ __launch_bounds__(256) void k_sp_union_virtual(DW W, const int2* pairs, int n)
{
	b2dPhaseStamp(W);
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) ufUnion(W.toiParent, pairs[k].x, pairs[k].y);
}

// One workgroup. in: all ranks' slabs (header word 5 = contacts created, descriptors at `tailAt` words into the slab).
__global__ __launch_bounds__(1024) void k_sp_merge_tails(DW W, const int* in, size_t strideWords, size_t tailAt, int base, int toiOrderBase)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const ContactArrays& T = W.ca[1 - S->cur]; // (scratch between compactions: this rank's tail is parked here while it moves)
	__shared__ int s_start[SHARD_MAX_RANKS + 1];
	__shared__ int s_cand[SP_TAIL_MAX];
	__shared__ int s_dst[SP_TAIL_MAX];
	const int tid = (int)threadIdx.x;
	if (tid == 0)
	{
		int at = 0;
		for (int r = 0; r < W.shardCount; ++r) { s_start[r] = at; at += in[(size_t)r * strideWords + 5]; }
		s_start[W.shardCount] = at;
	}
	__syncthreads();
	const int total = s_start[W.shardCount];
	if (total == 0) return;
	if (total > SP_TAIL_MAX || base + total > W.capContacts) { if (tid == 0) atomicOr(&S->c.overflow, 2048); return; }
	auto entry = [&](int e, int* r) -> const int*
	{
		int rr = 0;
		while (e >= s_start[rr + 1]) ++rr;
		*r = rr;
		return in + (size_t)rr * strideWords + tailAt + (size_t)(e - s_start[rr]) * SP_TAIL_WORDS;
	};
	// park this rank's tail
	const int mine = in[(size_t)W.shardRank * strideWords + 5];
	for (int k = tid; k < mine; k += 1024)
	{
		const int i = base + k;
		T.ids[i] = C.ids[i]; T.key[i] = C.key[i]; T.flags[i] = C.flags[i]; T.mat[i] = C.mat[i]; T.man0[i] = C.man0[i];
		T.man1[i] = C.man1[i]; T.imp[i] = C.imp[i]; T.man3[i] = C.man3[i]; T.color[i] = C.color[i]; T.mgr[i] = C.mgr[i];
	}
	__syncthreads();
	// the merged order: rank of every entry by (alpha, event key, pair key); equal keys cannot come from two ranks (an event
	// belongs to one rank; a pair created twice would straddle, which is refused)
	for (int e = tid; e < total; e += 1024)
	{
		int r;
		const int* o = entry(e, &r);
		const uint32_t a = (uint32_t)o[0];
		const unsigned long long ev = ((unsigned long long)(uint32_t)o[1] << 32) | (uint32_t)o[2];
		const unsigned long long pk = ((unsigned long long)(uint32_t)W.p_key[o[3]] << 32) | (uint32_t)W.p_key[o[4]];
		int rank = 0;
		for (int j = 0; j < total; ++j)
		{
			if (j == e) continue;
			int rj;
			const int* q = entry(j, &rj);
			const uint32_t aj = (uint32_t)q[0];
			const unsigned long long evj = ((unsigned long long)(uint32_t)q[1] << 32) | (uint32_t)q[2];
			const unsigned long long pkj = ((unsigned long long)(uint32_t)W.p_key[q[3]] << 32) | (uint32_t)W.p_key[q[4]];
			const bool before = aj != a ? aj < a : (evj != ev ? evj < ev : (pkj != pk ? pkj < pk : j < e));
			if (before) ++rank;
		}
		s_dst[e] = rank;
	}
	__syncthreads();
	for (int e = tid; e < total; e += 1024)
	{
		int r;
		const int* o = entry(e, &r);
		const int dst = base + s_dst[e];
		int pA = o[3], pB = o[4];
		if (b2dContactSwap(W.shapes[W.p_shape[pA]].type, W.shapes[W.p_shape[pB]].type) == 1) { const int t = pA; pA = pB; pB = t; }
		const int bodyA = W.p_body[pA], bodyB = W.p_body[pB];
		bool cand;
		if (r == W.shardRank)
		{
			const int src = base + o[5];
			C.ids[dst] = T.ids[src]; C.key[dst] = T.key[src]; C.flags[dst] = T.flags[src]; C.mat[dst] = T.mat[src]; C.man0[dst] = T.man0[src];
			C.man1[dst] = T.man1[src]; C.imp[dst] = T.imp[src]; C.man3[dst] = T.man3[src]; C.color[dst] = T.color[src];
			cand = (T.flags[src] & CF_TOI_CANDIDATE) != 0;
		}
		else
		{
			// OnContactCreate (b2ContactManager.cpp:507-564) as k_create_contacts does it; the bodies' wake-up came with their rows
			const bool sensor = ((W.p_filter1[pA] | W.p_filter1[pB]) & PF_SENSOR) != 0;
			cand = isToiCandidate(W, pA, pB, bodyA, bodyB);
			const float2 mA = W.p_mat[pA], mB = W.p_mat[pB];
			C.ids[dst] = make_int4(pA, pB, bodyA, bodyB);
			C.key[dst] = ((uint64_t)(uint32_t)W.p_key[o[3]] << 32) | (uint32_t)W.p_key[o[4]];
			C.flags[dst] = CF_ENABLED | (sensor ? CF_SENSOR : 0u) | (cand ? CF_TOI_CANDIDATE : 0u) | CF_FOREIGN;
			C.mat[dst] = make_float4(b2dSqrt(mA.x * mB.x), mA.y > mB.y ? mA.y : mB.y, 0.0f, 1.0f);
			C.man0[dst] = make_float4(0, 0, 0, 0);
			C.man1[dst] = make_float4(0, 0, 0, 0);
			C.imp[dst] = make_float4(0, 0, 0, 0);
			C.man3[dst] = make_int4(0, 0, 0, 0);
			C.color[dst] = -1;
		}
		C.mgr[dst] = -1;
		s_cand[s_dst[e]] = cand ? 1 : 0;
	}
	__syncthreads();
	// b2ContactManager::AddToContactArray: the new TOI candidates take the next slots in (merged) creation order
	for (int g = tid; g < total; g += 1024)
	{
		if (!s_cand[g]) continue;
		int before = 0;
		for (int j = 0; j < g; ++j) before += s_cand[j];
		C.mgr[base + g] = toiOrderBase + before;
		W.toiPos2c[toiOrderBase + before] = base + g;
	}
	__syncthreads();
	if (tid == 0)
	{
		int cands = 0;
		for (int j = 0; j < total; ++j) cands += s_cand[j];
		S->c.nContacts = base + total;
		S->c.nToiOrder = toiOrderBase + cands;
	}
}

// census of who owns what (for the hosts' slab sizes and b2hip_get_counters): bodies and proxies per rank
__global__ __launch_bounds__(256) void k_sp_owner_census(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	// (a workgroup counts in LDS and sends one atomic per rank and counter: a million bodies on sixteen words is otherwise
	// a million same-address atomics - 5 ms of the 1 M-body field's resolution)
	__shared__ int s_bodies[SHARD_MAX_RANKS], s_proxies[SHARD_MAX_RANKS];
	if (threadIdx.x < SHARD_MAX_RANKS) { s_bodies[threadIdx.x] = 0; s_proxies[threadIdx.x] = 0; }
	__syncthreads();
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		if ((W.b_flags[i] & BF_TYPE_MASK) == BT_STATIC) continue;
		const int o = W.b_owner[i];
		int np = 0;
		for (int q = W.b_proxyHead[i]; q >= 0; q = W.p_next[q]) ++np;
		atomicAdd(&s_bodies[o], 1);
		atomicAdd(&s_proxies[o], np);
	}
	__syncthreads();
	if (threadIdx.x < SHARD_MAX_RANKS)
	{
		if (s_bodies[threadIdx.x]) atomicAdd(&S->c.spBodies[threadIdx.x], s_bodies[threadIdx.x]);
		if (s_proxies[threadIdx.x]) atomicAdd(&S->c.spProxies[threadIdx.x], s_proxies[threadIdx.x]);
	}
}

// The headers of all ranks' slabs (and, for E4, the count of conflicts k_sp_tail_pairs found) to the host in one go: one wave
// writes them into pinned memory, the sequence number last - the host polls it (b2hip.hip: spReadHeaders), as it does for the
// state read-back. Replaces one small copy per rank and a stream synchronisation per exchange.
__global__ __launch_bounds__(64) void k_sp_collect_headers(const int* __restrict__ recv, size_t strideWords, int ranks, const int* __restrict__ extra, int* out, int seq)
{
	const int tid = (int)threadIdx.x;
	for (int k = tid; k < ranks * SP_HEADER_WORDS; k += 64)
		out[2 + k] = recv[(size_t)(k / SP_HEADER_WORDS) * strideWords + (size_t)(k % SP_HEADER_WORDS)];
	if (tid == 0) out[1] = extra ? *extra : 0;
	__threadfence_system();
	__syncthreads();
	if (tid == 0) __hip_atomic_store(&out[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#endif
