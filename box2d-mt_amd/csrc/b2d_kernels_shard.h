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
// After the solve every rank writes the records of what it owns into one int32 buffer - SHARD_NOBODY (INT32_MIN) elsewhere -
// the caller reduces the buffers with MAX over the ranks (RCCL all-reduce over xGMI on GPUs, gloo on CPUs: any bit
// pattern is >= INT32_MIN as an int32 and every record has exactly one owner, so MAX selects the owner's bits), and
// k_shard_import writes the other ranks' results into the world. One collective per step, 52 B per body + 20 B per contact.
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
	if (rank % W.shardCount == W.shardRank)
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

// out: [nBodies x SHARD_BODY_WORDS][nContacts x SHARD_CONTACT_WORDS][nJoints x SHARD_JOINT_WORDS]
__global__ __launch_bounds__(256) void k_shard_export(DW W, int* out)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int nb = W.nBodies, nc = S->c.nContacts, nj = W.nJoints;
	const int me = W.shardRank + 1;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
	{
		int* o = out + (size_t)i * SHARD_BODY_WORDS;
		if (!shardOwnsBody(W, i))
		{
			for (int k = 0; k < SHARD_BODY_WORDS; ++k) o[k] = SHARD_NOBODY;
			continue;
		}
		const float4 p = W.b_pos[i], v = W.b_vel[i], xf = W.b_xf[i];
		o[0] = __float_as_int(p.x); o[1] = __float_as_int(p.y); o[2] = __float_as_int(p.z); o[3] = __float_as_int(p.w);
		o[4] = __float_as_int(v.x); o[5] = __float_as_int(v.y); o[6] = __float_as_int(v.z);
		o[7] = (W.b_flags[i] & BF_AWAKE) ? 1 : 0;
		o[8] = __float_as_int(xf.x); o[9] = __float_as_int(xf.y); o[10] = __float_as_int(xf.z); o[11] = __float_as_int(xf.w);
		o[12] = me;
	}
	int* oc = out + (size_t)nb * SHARD_BODY_WORDS;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x)
	{
		int* o = oc + (size_t)i * SHARD_CONTACT_WORDS;
		bool mine = false;
		if (contactSolid(C.flags[i]))
		{
			const int4 ids = C.ids[i];
			mine = shardOwnsBody(W, (W.b_flags[ids.z] & BF_TYPE_MASK) != BT_STATIC ? ids.z : ids.w);
		}
		if (!mine)
		{
			for (int k = 0; k < SHARD_CONTACT_WORDS; ++k) o[k] = SHARD_NOBODY;
			continue;
		}
		const float4 im = C.imp[i];
		o[0] = __float_as_int(im.x); o[1] = __float_as_int(im.y); o[2] = __float_as_int(im.z); o[3] = __float_as_int(im.w);
		o[4] = me;
	}
	int* oj = oc + (size_t)nc * SHARD_CONTACT_WORDS;
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x)
	{
		int* o = oj + (size_t)j * SHARD_JOINT_WORDS;
		const JointRec& jn = W.joints[j];
		bool mine = false;
		if (jn.type != B2D_JOINT_DEAD)
		{
			const int b = (W.b_flags[jn.bodyA] & BF_TYPE_MASK) != BT_STATIC ? jn.bodyA : jn.bodyB;
			mine = shardOwnsBody(W, b);
		}
		if (!mine)
		{
			for (int k = 0; k < SHARD_JOINT_WORDS; ++k) o[k] = SHARD_NOBODY;
			continue;
		}
		o[0] = __float_as_int(jn.type == B2D_JOINT_GEAR ? W.gears[jn.enableLimit].impulse : jn.impulseX);
		o[1] = __float_as_int(jn.impulseY);
		o[2] = __float_as_int(jn.impulseZ);
		o[3] = __float_as_int(jn.motorImpulse);
		o[4] = jn.limitState;
		o[5] = me;
	}
}

// in: the MAX over the ranks of what k_shard_export wrote. Records owned by another rank are written into the world.
__global__ __launch_bounds__(256) void k_shard_import(DW W, const int* in)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int nb = W.nBodies, nc = S->c.nContacts, nj = W.nJoints;
	const int me = W.shardRank + 1;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
	{
		const int* o = in + (size_t)i * SHARD_BODY_WORDS;
		const int owner = o[12];
		if (owner <= 0 || owner == me) continue;
		// what the owner's solve did to the body: the sweep origin is where it stood (b2Island.cpp:200-204), then the results
		const float4 old = W.b_pos[i];
		W.b_pos0[i] = make_float4(old.x, old.y, old.z, 0.0f);
		W.b_pos[i] = make_float4(__int_as_float(o[0]), __int_as_float(o[1]), __int_as_float(o[2]), __int_as_float(o[3]));
		W.b_vel[i] = make_float4(__int_as_float(o[4]), __int_as_float(o[5]), __int_as_float(o[6]), 0.0f);
		W.b_xf[i] = make_float4(__int_as_float(o[8]), __int_as_float(o[9]), __int_as_float(o[10]), __int_as_float(o[11]));
		uint32_t f = W.b_flags[i];
		if (o[7]) f |= BF_AWAKE;
		else
		{
			// the island fell asleep (b2Body::SetAwake(false), b2Body.h:704-717)
			f &= ~BF_AWAKE;
			W.b_force[i] = make_float4(0, 0, 0, 0);
		}
		W.b_flags[i] = f;
	}
	const int* ic = in + (size_t)nb * SHARD_BODY_WORDS;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x)
	{
		const int* o = ic + (size_t)i * SHARD_CONTACT_WORDS;
		if (o[4] <= 0 || o[4] == me) continue;
		C.imp[i] = make_float4(__int_as_float(o[0]), __int_as_float(o[1]), __int_as_float(o[2]), __int_as_float(o[3]));
	}
	const int* ij = ic + (size_t)nc * SHARD_CONTACT_WORDS;
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x)
	{
		const int* o = ij + (size_t)j * SHARD_JOINT_WORDS;
		if (o[5] <= 0 || o[5] == me) continue;
		JointRec& jn = W.joints[j];
		if (jn.type == B2D_JOINT_GEAR) W.gears[jn.enableLimit].impulse = __int_as_float(o[0]);
		else jn.impulseX = __int_as_float(o[0]);
		jn.impulseY = __int_as_float(o[1]);
		jn.impulseZ = __int_as_float(o[2]);
		jn.motorImpulse = __int_as_float(o[3]);
		jn.limitState = o[4];
	}
}

#endif
