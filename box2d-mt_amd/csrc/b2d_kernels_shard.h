// b2d_kernels_shard.h - one world over the GPUs of a node, sharded by island.
//
// Every rank holds the WHOLE world - the same bodies, proxies and contacts with the same ids - and runs Collide, the island
// build, SynchronizeFixtures, the pair update and the TOI phase on all of it: those are deterministic functions of the
// shared state, so the ranks stay bit-identical without talking. What is sharded is b2Island::Solve: islands share only
// static bodies (b2World.cpp:1236-1241), so each island is solved by ONE rank - the others mark its bodies "in an island"
// (they will have moved) and wait for the result. Ownership is a pure function of the state every rank has:
//   * islands with more than SHARD_BIG_BODIES bodies (the pyramids of config 4) are dealt round robin in the order of their
//     root ids (k_shard_big), so that N big islands keep N ranks busy;
//   * every other island goes to hash(root) % ranks (k_island_classify).
// After the solve every rank packs the records of what it owns - every record carries its id - into its SLAB; the slabs are
// all-gathered (RCCL over xGMI: ncclAllGather on the world's stream, driven from this library, b2hip_shard_connect; gloo on
// CPUs in the tests), and k_shard_import writes the other ranks' records into the world. A slab holds exactly what a rank
// owns (52 B per body + 20 B per contact + 24 B per joint of ITS islands), and since the island build is replicated every
// rank has counted every rank's slab (Counters::shardBodies / Contacts / Joints): the hosts size the collective from the
// island census they read anyway - no size exchange, no host synchronisation around the collective.
// Reference: b2World::Solve's per-island independence (b2World.cpp:1166-1431); north_star's "RCCL all-gather of boundary
// body velocities" is this exchange (there are no boundary bodies between islands: what is gathered is whole islands).
#ifndef B2D_KERNELS_SHARD_H
#define B2D_KERNELS_SHARD_H

#include "b2d_kernels_island.h"

// One workgroup: rank the big islands by root id, deal them round robin, keep ours on the large-island list.
__global__ __launch_bounds__(1024) void k_shard_big(DW W)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const int n = S->c.nBigIslands < SHARD_BIG_MAX ? S->c.nBigIslands : SHARD_BIG_MAX;
	const int t = (int)threadIdx.x;
	if (t >= n) return;
	const int root = W.bigRoots[t];
	int rank = 0;
	for (int k = 0; k < n; ++k) rank += W.bigRoots[k] < root ? 1 : 0;
	const int owner = rank % W.shardCount;
	atomicAdd(&S->c.shardBodies[owner], W.rootBodies[root]);
	atomicAdd(&S->c.shardContacts[owner], W.rootContacts[root]);
	atomicAdd(&S->c.shardJoints[owner], W.rootJoints[root]);
	if (owner == W.shardRank)
	{
		const int k = atomicAdd(&S->c.nLIslands, 1);
		W.li_roots[k] = root;
	}
	else
	{
		W.rootIsland[root] = ROOT_REMOTE;
		atomicAdd(&S->c.nRemoteIslands, 1);
	}
}

__device__ __forceinline__ bool shardOwnsBody(const DW& W, int body)
{
	const uint32_t f = W.b_flags[body];
	if ((f & BF_TYPE_MASK) == BT_STATIC || (f & BF_ISLAND) == 0) return false;
	const int tier = W.rootIsland[W.parent[body]];
	return tier == ROOT_SMALL || tier == ROOT_LARGE;
}

// This rank's slab: [bodies x SHARD_BODY_WORDS][contacts x SHARD_CONTACT_WORDS][joints x SHARD_JOINT_WORDS], the three counts as
// the island build left them in Counters::shard*[rank]; records are appended in no particular order (they carry their ids).
__global__ __launch_bounds__(256) void k_shard_export(DW W, int* out)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int nb = W.nBodies, nc = S->c.nContacts, nj = W.nJoints;
	const int me = W.shardRank;
	const int capB = S->c.shardBodies[me], capC = S->c.shardContacts[me], capJ = S->c.shardJoints[me];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
	{
		if (!shardOwnsBody(W, i)) continue;
		const int k = atomicAdd(&S->c.shardCursor[0], 1);
		if (k >= capB) { atomicOr(&S->c.overflow, 512); continue; } // (cannot happen: the census counted the same bodies)
		int* o = out + (size_t)k * SHARD_BODY_WORDS;
		const float4 p = W.b_pos[i], v = W.b_vel[i], xf = W.b_xf[i];
		o[0] = i;
		o[1] = __float_as_int(p.x); o[2] = __float_as_int(p.y); o[3] = __float_as_int(p.z); o[4] = __float_as_int(p.w);
		o[5] = __float_as_int(v.x); o[6] = __float_as_int(v.y); o[7] = __float_as_int(v.z);
		o[8] = (W.b_flags[i] & BF_AWAKE) ? 1 : 0;
		o[9] = __float_as_int(xf.x); o[10] = __float_as_int(xf.y); o[11] = __float_as_int(xf.z); o[12] = __float_as_int(xf.w);
	}
	int* oc = out + (size_t)capB * SHARD_BODY_WORDS;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x)
	{
		if (!contactSolid(C.flags[i])) continue;
		const int4 ids = C.ids[i];
		if (!shardOwnsBody(W, (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC ? ids.z : ids.w)) continue;
		const int k = atomicAdd(&S->c.shardCursor[1], 1);
		if (k >= capC) { atomicOr(&S->c.overflow, 512); continue; }
		int* o = oc + (size_t)k * SHARD_CONTACT_WORDS;
		const float4 im = C.imp[i];
		o[0] = i;
		o[1] = __float_as_int(im.x); o[2] = __float_as_int(im.y); o[3] = __float_as_int(im.z); o[4] = __float_as_int(im.w);
	}
	int* oj = oc + (size_t)capC * SHARD_CONTACT_WORDS;
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x)
	{
		const JointRec& jn = W.joints[j];
		if (jn.type == B2D_JOINT_DEAD) continue;
		if (((W.b_flags[jn.bodyA] & W.b_flags[jn.bodyB]) & BF_ACTIVE) == 0) continue; // (not in any island: b2World.cpp:1303-1307)
		if (!shardOwnsBody(W, (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC ? jn.bodyA : jn.bodyB)) continue;
		const int k = atomicAdd(&S->c.shardCursor[2], 1);
		if (k >= capJ) { atomicOr(&S->c.overflow, 512); continue; }
		int* o = oj + (size_t)k * SHARD_JOINT_WORDS;
		o[0] = j;
		o[1] = __float_as_int(jn.type == B2D_JOINT_GEAR ? W.gears[jn.enableLimit].impulse : jn.impulseX);
		o[2] = __float_as_int(jn.impulseY);
		o[3] = __float_as_int(jn.impulseZ);
		o[4] = __float_as_int(jn.motorImpulse);
		o[5] = jn.limitState;
	}
}

// in: the slabs of all ranks, rank r's at in + r * strideWords. The other ranks' records are written into the world.
__global__ __launch_bounds__(256) void k_shard_import(DW W, const int* in, size_t strideWords)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	for (int r = 0; r < W.shardCount; ++r)
	{
		if (r == W.shardRank) continue;
		const int* slab = in + (size_t)r * strideWords;
		const int nB = S->c.shardBodies[r], nC = S->c.shardContacts[r], nJ = S->c.shardJoints[r];
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nB; k += gridDim.x * blockDim.x)
		{
			const int* o = slab + (size_t)k * SHARD_BODY_WORDS;
			const int i = o[0];
			if (i < 0 || i >= W.nBodies) continue;
			// what the owner's solve did to the body: the sweep origin is where it stood (b2Island.cpp:200-204), then the results
			const float4 old = W.b_pos[i];
			W.b_pos0[i] = make_float4(old.x, old.y, old.z, 0.0f);
			W.b_pos[i] = make_float4(__int_as_float(o[1]), __int_as_float(o[2]), __int_as_float(o[3]), __int_as_float(o[4]));
			W.b_vel[i] = make_float4(__int_as_float(o[5]), __int_as_float(o[6]), __int_as_float(o[7]), 0.0f);
			W.b_xf[i] = make_float4(__int_as_float(o[9]), __int_as_float(o[10]), __int_as_float(o[11]), __int_as_float(o[12]));
			uint32_t f = W.b_flags[i];
			if (o[8]) f |= BF_AWAKE;
			else
			{
				// the island fell asleep (b2Body::SetAwake(false), b2Body.h:704-717)
				f &= ~BF_AWAKE;
				W.b_force[i] = make_float4(0, 0, 0, 0);
			}
			W.b_flags[i] = f;
		}
		const int* ic = slab + (size_t)nB * SHARD_BODY_WORDS;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nC; k += gridDim.x * blockDim.x)
		{
			const int* o = ic + (size_t)k * SHARD_CONTACT_WORDS;
			const int i = o[0];
			if (i < 0 || i >= S->c.nContacts) continue;
			C.imp[i] = make_float4(__int_as_float(o[1]), __int_as_float(o[2]), __int_as_float(o[3]), __int_as_float(o[4]));
		}
		const int* ij = ic + (size_t)nC * SHARD_CONTACT_WORDS;
		for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nJ; k += gridDim.x * blockDim.x)
		{
			const int* o = ij + (size_t)k * SHARD_JOINT_WORDS;
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

#endif
