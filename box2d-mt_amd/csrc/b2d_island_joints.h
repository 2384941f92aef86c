// b2d_island_joints.h - the joints of ONE island, walked in the island's joint order by one lane
// (b2Island::Solve, b2Island.cpp:256-335: InitVelocityConstraints after the contacts' warm start, SolveVelocityConstraints
// before the contacts in every velocity iteration, SolvePositionConstraints after them in every position iteration).
//
// Shared by the two places an island's joints are solved: the launch-per-colour solver (body rows in HBM,
// b2d_kernels_solve_large.h) and the small-island solver (body rows in LDS, b2d_kernels_solve_small.h). `Bodies` hides where
// the rows live:
//     float4 pos(int body)   c.x, c.y, a, (sleep time or nothing)        void setPos(int body, V2 c, float a)
//     float4 vel(int body)   v.x, v.y, w                                 void setVel(int body, V2 v, float w)
// Writes are only ever made for non-static bodies.
#ifndef B2D_ISLAND_JOINTS_H
#define B2D_ISLAND_JOINTS_H

#include "b2d_kernels_island.h"

enum { JOINTS_INIT = 0, JOINTS_VELOCITY = 1, JOINTS_POSITION = 2 };

// Returns jointsOkay (meaningful for JOINTS_POSITION).
template <class Bodies>
__device__ inline int b2dSolveIslandJoints(const DW& W, const StepParams& sp, int mode, int start, int nj, Bodies& B)
{
	int okay = 1;
	for (int t = 0; t < nj; ++t)
	{
		JointRec* j = &W.joints[W.lj_list[start + t]];
		if (j->type == B2D_JOINT_GEAR)
		{
			// four bodies; loaded into separate copies and written back A, B, C, D like the reference does
			GearRec* g = &W.gears[j->enableLimit];
			const int ids[4] = { j->bodyA, j->bodyB, g->bodyC, g->bodyD };
			bool ns[4];
			GearBodies gb;
			BodyPos* gp[4] = { &gb.pA, &gb.pB, &gb.pC, &gb.pD };
			BodyVel* gv[4] = { &gb.vA, &gb.vB, &gb.vC, &gb.vD };
			for (int q = 0; q < 4; ++q)
			{
				ns[q] = (W.b_flags[ids[q]] & BF_TYPE_MASK) != BT_STATIC;
				const float4 p4 = B.pos(ids[q]);
				const float4 v4 = B.vel(ids[q]);
				gp[q]->c = v2(p4.x, p4.y); gp[q]->a = p4.z;
				gv[q]->v = ns[q] ? v2(v4.x, v4.y) : v2(0, 0); gv[q]->w = ns[q] ? v4.z : 0.0f;
			}
			if (mode == JOINTS_POSITION)
			{
				b2dGearSolvePosition(g, &gb);
				for (int q = 0; q < 4; ++q)
					if (ns[q]) B.setPos(ids[q], gp[q]->c, gp[q]->a);
			}
			else
			{
				if (mode == JOINTS_INIT)
				{
					float im[4], ii[4];
					V2 lc[4];
					for (int q = 0; q < 4; ++q)
					{
						const float4 m = W.b_mass[ids[q]];
						im[q] = m.x; ii[q] = m.y; lc[q] = v2(m.z, m.w);
					}
					b2dGearInit(g, &gb, im, ii, lc, sp.warmStarting != 0);
				}
				else
					b2dGearSolveVelocity(g, &gb);
				for (int q = 0; q < 4; ++q)
					if (ns[q]) B.setVel(ids[q], gv[q]->v, gv[q]->w);
			}
			continue;
		}
		const int bA = j->bodyA, bB = j->bodyB;
		const bool nsA = (W.b_flags[bA] & BF_TYPE_MASK) != BT_STATIC;
		const bool nsB = (W.b_flags[bB] & BF_TYPE_MASK) != BT_STATIC;
		const float4 pa = B.pos(bA), pb = B.pos(bB);
		BodyPos pA, pB;
		pA.c = v2(pa.x, pa.y); pA.a = pa.z;
		pB.c = v2(pb.x, pb.y); pB.a = pb.z;
		if (mode == JOINTS_POSITION)
		{
			const bool ok = b2dJointSolvePosition(j, &pA, &pB);
			okay = okay && ok;
			if (nsA) B.setPos(bA, pA.c, pA.a);
			if (nsB) B.setPos(bB, pB.c, pB.a);
		}
		else
		{
			const float4 va = B.vel(bA), vb = B.vel(bB);
			BodyVel vA, vB;
			vA.v = nsA ? v2(va.x, va.y) : v2(0, 0); vA.w = nsA ? va.z : 0.0f;
			vB.v = nsB ? v2(vb.x, vb.y) : v2(0, 0); vB.w = nsB ? vb.z : 0.0f;
			if (mode == JOINTS_INIT)
			{
				const float4 mA = W.b_mass[bA], mB = W.b_mass[bB];
				b2dJointInit(j, mA.x, mA.y, v2(mA.z, mA.w), mB.x, mB.y, v2(mB.z, mB.w), pA, &vA, pB, &vB,
					sp.warmStarting != 0, sp.dtRatio, sp.dt);
			}
			else
			{
				b2dJointSolveVelocity(j, &vA, &vB, sp.dt, sp.inv_dt);
			}
			if (nsA) B.setVel(bA, vA.v, vA.w);
			if (nsB) B.setVel(bB, vB.v, vB.w);
		}
	}
	return okay;
}

// Body rows in HBM (the launch-per-colour solver)
struct JointBodiesGlobal
{
	const DW& W;
	__device__ JointBodiesGlobal(const DW& w) : W(w) {}
	__device__ float4 pos(int body) const { return W.b_pos[body]; }
	__device__ float4 vel(int body) const { return W.b_vel[body]; }
	__device__ void setPos(int body, V2 c, float a) const
	{
		const float sleepTime = W.b_pos[body].w;
		W.b_pos[body] = make_float4(c.x, c.y, a, sleepTime);
	}
	__device__ void setVel(int body, V2 v, float w) const { W.b_vel[body] = make_float4(v.x, v.y, w, 0.0f); }
};

// Body rows of one small-island chunk in LDS; anything outside the chunk (static bodies, a gear's far bodies) in HBM
struct JointBodiesLds
{
	const DW& W;
	float4* s_pos;
	float4* s_vel;
	int bStart, nB;
	__device__ JointBodiesLds(const DW& w, float4* p, float4* v, int b0, int n) : W(w), s_pos(p), s_vel(v), bStart(b0), nB(n) {}
	__device__ int slotOf(int body) const
	{
		if ((W.b_flags[body] & BF_TYPE_MASK) == BT_STATIC) return -1;
		const int s = W.b_slot[body] - bStart;
		return (s >= 0 && s < nB && W.si_bodies[bStart + s] == body) ? s : -1;
	}
	__device__ float4 pos(int body) const { const int s = slotOf(body); return s >= 0 ? s_pos[s] : W.b_pos[body]; }
	__device__ float4 vel(int body) const { const int s = slotOf(body); return s >= 0 ? s_vel[s] : W.b_vel[body]; }
	__device__ void setPos(int body, V2 c, float a) const { const int s = slotOf(body); if (s >= 0) s_pos[s] = make_float4(c.x, c.y, a, 0.0f); }
	__device__ void setVel(int body, V2 v, float w) const { const int s = slotOf(body); if (s >= 0) s_vel[s] = make_float4(v.x, v.y, w, 0.0f); }
};

#endif
