// b2d_kernels_sweep_end.h - the END of one Gauss-Seidel sweep over the large islands, in ONE single-workgroup launch.
//
// A large island that no resident block solver can take (the settled 100 000-box Tumbler: one island of 370 000
// constraints) is swept launch per colour, and a step is priced by the NUMBER of dependent launches (~5.8 us each, whatever
// they hold: DESIGN.md section 4). Measured on that island (profiles/r05_a): 19 colours whose census falls off steeply -
// 41 000, 40 600, ... 8 300, 5 600, 3 400, 1 800, 760, 224, 59, 11 rows - then the hub's 900 constraints swept by one
// workgroup in 15 chained chunks (82 us), then one lane walking the island's joints: 21 launches per sweep, 12 sweeps per step.
// Everything behind the big colours has no use for 256 CUs; this kernel does it in one workgroup of 512 lanes with workgroup
// barriers where the launches had kernel boundaries (512, not 1024: a 1024-lane workgroup caps a lane at 128 registers, the
// velocity solve of the hub's fixed point needs ~166 and spilled inside its rounds - measured with the kernel's own stamps,
// B2HIP_SWEEP_STAMPS=1: the fixed point over the Tumbler's 900 rows 34.5 us as one pass of 1024 lanes, 21.5 us as two passes
// of 512, 28.7 us as four of 256):
//   A  the TAIL colours (those the host found small in this step's census), colour after colour - the arithmetic of
//      k_large_velocity / k_large_position row for row, so the result is the launch-per-colour one bit for bit;
//   B  the hub's constraints as fixed points over 512 lanes at a time (the chunks of 64 of k_large_hub were a chain of 15
//      fixed points, now of two; b2d_kernels_solve_large.h explains the scheme), prefix sums over the workgroup through LDS;
//   C  the hub rows B cannot take (a partner that occurs twice, a second hub, constraints swept in order for lack of a home
//      block) by one wave, turn by turn or chunk-wise: hubSweep<1> over the rest of the list;
//   D  what the host launched between two sweeps anyway: the island's joints (k_large_joints), the verdict of a position
//      iteration (k_large_pos_end) and the reset for the next one (k_large_pos_begin).
// Reference: b2Island::Solve, Box2D/Dynamics/b2Island.cpp:256-336 (joints before contacts in a velocity iteration, after them
// in a position iteration), b2ContactSolver.cpp:293-603, 676-752.
#ifndef B2D_KERNELS_SWEEP_END_H
#define B2D_KERNELS_SWEEP_END_H

#include "b2d_handover.h"

#define SWEEP_END_LANES 512
#define SWEEP_END_WAVES (SWEEP_END_LANES / 64)
// what a launch does behind its tail colours (bits of `what`)
#define SE_HUB 1          // the hub rows (B, C)
#define SE_GUESS 2        // ... starting from the changes the previous sweep of the same kind found
#define SE_JOINTS_INIT 4  // k_large_joints mode 0
#define SE_JOINTS_VEL 8   // k_large_joints mode 1
#define SE_JOINTS_POS 16  // k_large_joints mode 2
#define SE_POS_END 32     // k_large_pos_end
#define SE_POS_BEGIN 64   // k_large_pos_begin (for the next iteration)
#define SE_HUB_WIDE_ONLY 128 // B without C: the host launches k_large_hub for the rows behind Counters::nHubWide

// ---- the small colours of a sweep in ONE launch: data flow per body --------------------------------------------------------------
// The colour census of a big pile falls off steeply (the settled Tumbler: 12 colours hold 97 % of the 370 000 rows, the other
// 7 - whatever a few thousand bodies with 13 to 17 contacts need - 12 000), and every colour is a launch of every sweep
// whatever it holds. The rows of the colours [restFirst, nColors) - the REST rows - only have to keep their order ON EVERY
// BODY: colour ascending. One lane per rest row; a body's rest rows hand its row on through a tagged 16-byte row in memory
// (b_cutv / b_posv: value + tag in one store, read past the L2 - the hand-over of k_blocks_sweep's cut constraints,
// b2d_handover.h): the row of rank k among the body's rest rows (DW::bodyRest: the body's rest colours of this step, k_color_fill)
// waits for tag + k, solves, stores tag + k + 1; rank 0 starts from the body table as the colours before left it, the last
// one writes the body table. A chain is as long as a body has rest colours - a few hops of ~3 us where the launches were
// 6 us each - and chains of different bodies do not wait for one another. The same arithmetic in the same order on every
// body as the launches: the same bits (tests/test_gpu_sweep_end.py).
// The workgroups of the launch wait for one another: a lane on a later trip of the stride loop may wait for a first-trip row
// of a workgroup with a HIGHER index only if that workgroup is running, so ALL workgroups of the launch must be resident
// together. The host clamps the grid to the occupancy query times the CUs (restMaxWG, restHubMaxWG: b2hip_api_world.h;
// ADVICE round 5) and lets the rows beyond it be walked in strides.
#define REST_ROWS_MAX 196608
// A body that the hub workgroup of a fused launch (k_rest_hub below) touches - a partner of a hub row, a body of a joint of a
// large island - carries this bit in DW::bodyRest (k_color_fill, k_joints_fill; HUB_COLOR is never a rest colour). In a fused
// launch the LAST rest row of such a body hands its row on like the others (a tagged row) instead of writing the body table:
// the hub workgroup waits for that tag, puts the row into the table itself and goes on from there - so the table row of such
// a body has ONE writer per launch, and the hub's rows, the joints and the verdict need no launch of their own.
#define REST_SERIAL_BIT (1ull << HUB_COLOR)
// `wg` of `nWG` workgroups of blockDim.x lanes sweep the rest rows; `serialTagged`: a fused launch (see REST_SERIAL_BIT)
template <int MODE>
__device__ __forceinline__ bool restSweepRows(const DW& W, int restFirst, int nColors, int* bar, int epoch, int wg, int nWG, bool serialTagged)
{
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int begin = W.colorStart[restFirst], end = W.colorStart[nColors];
	const int tag = (epoch & 0x7fff) << 16;
	float4* const rows = MODE == 2 ? W.b_pos : W.b_vel;
	float4* const xch = MODE == 2 ? W.b_posv : W.b_cutv;
	// The host sizes the grid from the step's colour census, which the colouring of new contacts can outgrow: a lane takes the
	// rows begin + its index, + the grid size, ... in turn. That cannot deadlock: a row only waits for rows of lower colours,
	// i.e. of lower index (the rows are sorted by colour), and those are taken earlier by their lanes or are being waited for
	// by lanes that are running - the lowest unfinished row can always go ahead.
	const int stride = (int)(nWG * blockDim.x);
	for (int base = begin; base < end; base += stride) // (uniform: every lane of the launch makes the same trips)
	{
		const int row = base + (int)(wg * blockDim.x + threadIdx.x);
		const bool have = row < end;
		LargeRef r;
		r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
		ContactConstraint cc;
		memset(&cc, 0, sizeof(cc));
		bool active = have;
		int rankA = 0, degA = 0, rankB = 0, degB = 0;
		bool tagLastA = false, tagLastB = false;
		float4 startA = make_float4(0, 0, 0, 0), startB = startA;
		if (have)
		{
			int col = restFirst;
			while (col + 1 < nColors && row >= W.colorStart[col + 1]) ++col;
			r = largeRef(W, C, row);
			if (MODE == 2)
			{
				active = W.rootDone[r.root] == 0;
				lcLoad(W, row, cc, LC_MASS_FIRST, LC_MASS_FIRST + 4);
				lcLoad(W, row, cc, LC_POS_FIRST, LC_WORDS);
				startA = rows[r.bodyA]; // (static bodies have a position too)
				startB = rows[r.bodyB];
			}
			else
			{
				lcLoad(W, row, cc, 0, LC_VEL_WORDS);
				if (r.nsA) startA = rows[r.bodyA];
				if (r.nsB) startB = rows[r.bodyB];
			}
			const unsigned long long below = (1ull << col) - 1ull;
			if (r.nsA) { const unsigned long long m = W.bodyRest[r.bodyA]; degA = __popcll(m & ~REST_SERIAL_BIT); rankA = __popcll(m & below); tagLastA = serialTagged && (m & REST_SERIAL_BIT) != 0ull; }
			if (r.nsB) { const unsigned long long m = W.bodyRest[r.bodyB]; degB = __popcll(m & ~REST_SERIAL_BIT); rankB = __popcll(m & below); tagLastB = serialTagged && (m & REST_SERIAL_BIT) != 0ull; }
		}
		const bool nsA = have && r.nsA, nsB = have && r.nsB;
		// (rank 0: nothing to wait for - the row comes from the body table)
		float4* const xA = nsA ? &xch[r.bodyA] : nullptr;
		float4* const xB = nsB ? &xch[r.bodyB] : nullptr;
		const int needA = tag + rankA, needB = tag + rankB;
		float minSep = 0.0f;
		const bool ok = dataflowRun(have && active, (nsA && rankA > 0) ? xA : nullptr, needA, (nsB && rankB > 0) ? xB : nullptr, needB, bar, &S->c.overflow, W.restPoll, [&](f4v ra, f4v rb)
		{
			float4 qa = startA, qb = startB;
			if (nsA && rankA > 0) qa = make_float4(ra.x, ra.y, ra.z, startA.w);
			if (nsB && rankB > 0) qb = make_float4(rb.x, rb.y, rb.z, startB.w);
			if (MODE == 2)
			{
				BodyPos pA, pB;
				pA.c = v2(qa.x, qa.y); pA.a = qa.z;
				pB.c = v2(qb.x, qb.y); pB.a = qb.z;
				b2dSolvePosition(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
				qa = make_float4(pA.c.x, pA.c.y, pA.a, startA.w);
				qb = make_float4(pB.c.x, pB.c.y, pB.a, startB.w);
			}
			else
			{
				BodyVel vA, vB;
				vA.v = v2(qa.x, qa.y); vA.w = qa.z;
				vB.v = v2(qb.x, qb.y); vB.w = qb.z;
				if (!nsA) { vA.v = v2(0, 0); vA.w = 0.0f; }
				if (!nsB) { vB.v = v2(0, 0); vB.w = 0.0f; }
				if (MODE == 0) b2dWarmStart(&cc, &vA, &vB); else b2dSolveVelocity(&cc, &vA, &vB);
				qa = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
				qb = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
			}
			// the body's last rest row puts it back into the body table (for the launches that follow), the others hand it on
			// (... to the hub workgroup of a fused launch, if that is who touches the body next: REST_SERIAL_BIT)
			if (nsA) { if (rankA + 1 < degA || tagLastA) stRow(xA, qa.x, qa.y, qa.z, needA + 1); else rows[r.bodyA] = qa; }
			if (nsB) { if (rankB + 1 < degB || tagLastB) stRow(xB, qb.x, qb.y, qb.z, needB + 1); else rows[r.bodyB] = qb; }
		});
		if (!ok) return false;
		if (MODE == 1 && have && active) lcStore(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
		if (MODE == 2) waveAtomicMaxU32Guarded(W.rootPen, r.root, floatBits(0.0f - minSep), have && active);
	}
	return true;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_large_rest(DW W, int restFirst, int nColors, int* bar, int epoch)
{
	b2dPhaseStamp(W);
	if (MODE == 2 && W.st->c.allLargeDone) return;
	(void)restSweepRows<MODE>(W, restFirst, nColors, bar, epoch, (int)blockIdx.x, (int)gridDim.x, false);
}

// ---- A: one tail colour -------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void sweepEndColour(const DW& W, const ContactArrays& C, int begin, int end)
{
	for (int base = begin; base < end; base += SWEEP_END_LANES)
	{
		const int row = base + (int)threadIdx.x;
		bool valid = row < end;
		LargeRef r;
		r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
		if (valid) r = largeRef(W, C, row);
		if (MODE == 2)
		{
			// (k_large_position, row for row)
			if (valid) valid = W.rootDone[r.root] == 0;
			float minSep = 0.0f;
			if (valid)
			{
				ContactConstraint cc;
				memset(&cc, 0, sizeof(cc));
				lcLoad(W, row, cc, LC_MASS_FIRST, LC_MASS_FIRST + 4);
				lcLoad(W, row, cc, LC_POS_FIRST, LC_WORDS);
				const float4 pa = W.b_pos[r.bodyA], pb = W.b_pos[r.bodyB];
				BodyPos pA, pB;
				pA.c = v2(pa.x, pa.y); pA.a = pa.z;
				pB.c = v2(pb.x, pb.y); pB.a = pb.z;
				b2dSolvePosition(&cc, &pA, &pB, B2D_BAUMGARTE, &minSep);
				if (r.nsA) W.b_pos[r.bodyA] = make_float4(pA.c.x, pA.c.y, pA.a, pa.w);
				if (r.nsB) W.b_pos[r.bodyB] = make_float4(pB.c.x, pB.c.y, pB.a, pb.w);
			}
			waveAtomicMaxU32Guarded(W.rootPen, r.root, floatBits(0.0f - minSep), valid);
		}
		else if (valid)
		{
			// (k_large_velocity, row for row)
			ContactConstraint cc;
			memset(&cc, 0, sizeof(cc));
			lcLoad(W, row, cc, 0, LC_VEL_WORDS);
			BodyVel vA, vB;
			vA.v = v2(0, 0); vA.w = 0; vB = vA;
			if (r.nsA) { const float4 v = W.b_vel[r.bodyA]; vA.v = v2(v.x, v.y); vA.w = v.z; }
			if (r.nsB) { const float4 v = W.b_vel[r.bodyB]; vB.v = v2(v.x, v.y); vB.w = v.z; }
			if (MODE == 0)
			{
				b2dWarmStart(&cc, &vA, &vB);
			}
			else
			{
				b2dSolveVelocity(&cc, &vA, &vB);
				lcStore(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
			}
			if (r.nsA) W.b_vel[r.bodyA] = make_float4(vA.v.x, vA.v.y, vA.w, 0.0f);
			if (r.nsB) W.b_vel[r.bodyB] = make_float4(vB.v.x, vB.v.y, vB.w, 0.0f);
		}
	}
}

// ---- the hub workgroup of a FUSED launch (k_rest_hub): what it waits for ----------------------------------------------------
// The rest rows of this sweep run in the other workgroups of the same launch. A body the hub workgroup touches (REST_SERIAL_BIT)
// that has rest rows arrives as a tagged row from the last of them: wait for that tag, put the row into the body table, go
// on as k_sweep_end would behind a k_large_rest launch - the same arithmetic in the same order on every body, the same bits.
#define REST_SETTLED 0x4000 // (on top of a tag: ranks stay below 64)
struct RestJoin
{
	int* bar;        // grid barrier words: [4] abort, [5] workgroups of fused launches that have finished their rest rows
	int tag;         // (epoch & 0x7fff) << 16 of this launch
	int arriveNeed;  // MODE 2: the verdict waits until bar[5] has reached this
};

template <int MODE>
__device__ __forceinline__ bool restSettleBody(const DW& W, const RestJoin& rj, int body)
{
	const unsigned long long m = W.bodyRest[body];
	if ((m & REST_SERIAL_BIT) == 0ull) return true; // (not ours to wait for: a static body, or a launch that is not fused)
	const int d = __popcll(m & ~REST_SERIAL_BIT);
	if (d == 0) return true; // (no rest rows: the colour launches left the row in the table)
	float4* const rows = MODE == 2 ? W.b_pos : W.b_vel;
	const float4* const x = (MODE == 2 ? W.b_posv : W.b_cutv) + body;
	const int need = rj.tag + d;
	int spins = 0;
	for (;;)
	{
		const f4v r = ldRow(x);
		if (__float_as_int(r.w) == need)
		{
			// (several lanes may settle one body at once - a body in two leftover rows, in two joints: the same value)
			rows[body] = make_float4(r.x, r.y, r.z, MODE == 2 ? rows[body].w : 0.0f);
			// ... and nobody settles it a second time: the table row is the hub workgroup's from here on (a partner's first row
			// with the hub is swept by the fixed point, its second one with the leftovers, later)
			stRow(const_cast<float4*>(x), r.x, r.y, r.z, need + REST_SETTLED);
			return true;
		}
		if (__float_as_int(r.w) == need + REST_SETTLED) return true;
		if (++spins > (rj.bar[6] != 0 ? rj.bar[6] : DATAFLOW_SPIN_MAX) || ((spins & 1023) == 0 && ldcI(&rj.bar[4]) != 0)) return false;
		__builtin_amdgcn_s_sleep(1);
	}
}

// The bodies of the hub rows [first, first + cnt) of hubList / of the joints of the large islands, settled. All lanes of the
// workgroup call; false: a wait was abandoned and every lane leaves. Called right before the rows / joints are swept: the
// hub rows are in the order in which their partners become ready (k_hub_order), so the first pass of the fixed point runs
// while the partners of the last one are still being worked on by the rest rows.
__device__ __forceinline__ bool restSettleVote(const DW& W, const RestJoin& rj, bool ok)
{
	const int bad = __syncthreads_or(ok ? 0 : 1);
	if (bad && threadIdx.x == 0)
	{
		stcI(&rj.bar[4], 1);
		atomicOr(&W.st->c.overflow, 64);
	}
	return bad == 0;
}
template <int MODE>
__device__ __forceinline__ bool restSettleHubRows(const DW& W, const ContactArrays& C, const RestJoin& rj, int first, int cnt)
{
	bool ok = true;
	for (int k = first + (int)threadIdx.x; k < first + cnt; k += SWEEP_END_LANES)
	{
		const LargeRef r = largeRef(W, C, W.hubList[k]);
		if (MODE == 2 && W.rootDone[r.root]) continue; // (a closed island's rows do not run: nothing is handed on)
		if (r.nsA) ok = restSettleBody<MODE>(W, rj, r.bodyA) && ok;
		if (r.nsB) ok = restSettleBody<MODE>(W, rj, r.bodyB) && ok;
	}
	return restSettleVote(W, rj, ok);
}
template <int MODE>
__device__ __forceinline__ bool restSettleJoints(const DW& W, const RestJoin& rj)
{
	DState* S = W.st;
	bool ok = true;
	if (W.nJoints > 0)
	{
		const int n = S->c.nLIslands;
		for (int k = (int)threadIdx.x; k < n; k += SWEEP_END_LANES)
		{
			const int root = W.li_roots[k];
			const int nj = W.rootJoints[root];
			if (nj == 0) continue;
			if (MODE == 2 && W.rootDone[root]) continue;
			const int start = W.rootJointStart[root];
			for (int t = 0; t < nj; ++t)
			{
				const JointRec* j = &W.joints[W.lj_list[start + t]];
				ok = restSettleBody<MODE>(W, rj, j->bodyA) && ok;
				ok = restSettleBody<MODE>(W, rj, j->bodyB) && ok;
				if (j->type == B2D_JOINT_GEAR)
				{
					const GearRec* g = &W.gears[j->enableLimit];
					ok = restSettleBody<MODE>(W, rj, g->bodyC) && ok;
					ok = restSettleBody<MODE>(W, rj, g->bodyD) && ok;
				}
			}
		}
	}
	return restSettleVote(W, rj, ok);
}

// ---- B: up to SWEEP_END_LANES rows of the PRIMARY hub as one fixed point ------------------------------------------------------------
// Rows [first, first + cnt) of hubList: constraints between the primary hub (the body with the most solid contacts:
// DW::hubMeta[0]) and `cnt` DIFFERENT partners, none of them a hub (k_hub_flag sorts the others out). The sequential sweep
// in list order is the fixed point of "every lane evaluates its constraint from the hub row it assumes it will meet": lane k's
// assumption is the hub row at the start plus the changes of the lanes before it, and depends on lanes < k only, so after
// k rounds it is final; a partner changes the hub's row by (its mass / the hub's), which is what an error shrinks by per
// round - two to four rounds in practice. Settled = no lane's assumption moves by more than 2^-21 of max(|row|, sum of the
// |changes|) per component: the rounding a sum over that many terms carries anyway.
// All lanes of the workgroup call. Returns the hub row behind the last row (every lane).
template <int MODE>
__device__ __forceinline__ float4 hubWidePass(const DW& W, const ContactArrays& C, int hubBody, float4 u0, int first, int cnt, int useGuess, int* roundsOut, const RestJoin* rj, bool* okOut)
{
	__shared__ float s_tot[2][SWEEP_END_WAVES][6]; // per wave: sums of the changes (x, y, z) and of their magnitudes
	__shared__ float4 s_hubOut;
	const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
	const int k = first + t;
	const bool have = t < cnt;
	float4* rows = MODE == 2 ? W.b_pos : W.b_vel;
	int row = 0;
	LargeRef r;
	r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
	ContactConstraint cc;
	memset(&cc, 0, sizeof(cc));
	bool active = have, hubIsA = true, otherDynamic = false;
	int otherBody = 0;
	float4 other = make_float4(0, 0, 0, 0);
	float4 guess = make_float4(0, 0, 0, 0);
	if (have)
	{
		row = W.hubList[k];
		r = largeRef(W, C, row);
		hubIsA = r.nsA && r.bodyA == hubBody;
		otherBody = hubIsA ? r.bodyB : r.bodyA;
		otherDynamic = hubIsA ? r.nsB : r.nsA;
		if (MODE == 2)
		{
			active = W.rootDone[r.root] == 0;
			lcLoad(W, row, cc, LC_MASS_FIRST, LC_MASS_FIRST + 4);
			lcLoad(W, row, cc, LC_POS_FIRST, LC_WORDS);
		}
		else lcLoad(W, row, cc, 0, LC_VEL_WORDS);
		if (useGuess) guess = W.hubDelta[k];
	}
	if (rj != nullptr)
	{
		// fused launch: the partner may still be on its way through the rest rows (the constraint's own words, asked for above,
		// arrive meanwhile). Settled = its row is in the body table, written by this lane: the load below sees it.
		bool ok = true;
		if (have && active && otherDynamic) ok = restSettleBody<MODE>(W, *rj, otherBody);
		if (!restSettleVote(W, *rj, ok)) { *okOut = false; return u0; }
	}
	if (have)
	{
		if (MODE == 2) other = rows[otherBody]; // static partners have a position too
		else if (otherDynamic) other = rows[otherBody];
	}
	const float imp0[4] = { cc.normalImpulse[0], cc.tangentImpulse[0], cc.normalImpulse[1], cc.tangentImpulse[1] };
	// exclusive prefix over the workgroup of (dx, dy, dz), in a fixed order: a shuffle tree inside the wave, the waves' totals
	// added up wave after wave; *totalAbs = the sums of the magnitudes over all lanes
	int buf = 0;
	auto prefix = [&](float dx, float dy, float dz, float* ex, float* ey, float* ez, float* ax, float* ay, float* az)
	{
		float sx = dx, sy = dy, sz = dz;
		float mx = fabsf(dx), my = fabsf(dy), mz = fabsf(dz);
#pragma unroll
		for (int off = 1; off < 64; off <<= 1)
		{
			const float ux = __shfl_up(sx, off), uy = __shfl_up(sy, off), uz = __shfl_up(sz, off);
			if (lane >= off) { sx += ux; sy += uy; sz += uz; }
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
		{
			mx += __shfl_xor(mx, off); my += __shfl_xor(my, off); mz += __shfl_xor(mz, off);
		}
		if (lane == 63)
		{
			s_tot[buf][wave][0] = sx; s_tot[buf][wave][1] = sy; s_tot[buf][wave][2] = sz;
			s_tot[buf][wave][3] = mx; s_tot[buf][wave][4] = my; s_tot[buf][wave][5] = mz;
		}
		__syncthreads();
		float bx = 0.0f, by = 0.0f, bz = 0.0f, tx = 0.0f, ty = 0.0f, tz = 0.0f;
		for (int q = 0; q < SWEEP_END_WAVES; ++q)
		{
			if (q < wave) { bx += s_tot[buf][q][0]; by += s_tot[buf][q][1]; bz += s_tot[buf][q][2]; }
			tx += s_tot[buf][q][3]; ty += s_tot[buf][q][4]; tz += s_tot[buf][q][5];
		}
		buf ^= 1; // (the next call writes the other buffer: nobody is still reading it - a barrier lies in between)
		*ex = bx + (sx - dx); *ey = by + (sy - dy); *ez = bz + (sz - dz);
		*ax = tx; *ay = ty; *az = tz;
	};
	float ex, ey, ez, ax, ay, az;
	float4 incoming = u0;
	if (useGuess)
	{
		prefix(guess.x, guess.y, guess.z, &ex, &ey, &ez, &ax, &ay, &az);
		incoming = make_float4(u0.x + ex, u0.y + ey, u0.z + ez, u0.w);
	}
	HubTrial tr;
	tr.hubOut = incoming; tr.otherOut = other; tr.minSep = 0.0f;
	tr.imp[0] = imp0[0]; tr.imp[1] = imp0[1]; tr.imp[2] = imp0[2]; tr.imp[3] = imp0[3];
	float dx = 0.0f, dy = 0.0f, dz = 0.0f;
	int rounds = 0;
	// (lane k is final after k rounds: cnt + 1 rounds always settle; in practice two to four)
	for (int round = 0; round <= cnt + 1; ++round)
	{
		dx = dy = dz = 0.0f;
		if (active)
		{
			tr = hubEvaluate(MODE, cc, imp0, hubIsA, otherDynamic, incoming, other);
			dx = tr.hubOut.x - incoming.x;
			dy = tr.hubOut.y - incoming.y;
			dz = tr.hubOut.z - incoming.z;
		}
		else tr.hubOut = incoming;
		prefix(dx, dy, dz, &ex, &ey, &ez, &ax, &ay, &az);
		const float4 next = make_float4(u0.x + ex, u0.y + ey, u0.z + ez, u0.w);
		const float tx = 0x1p-21f * fmaxf(fabsf(u0.x), ax), ty = 0x1p-21f * fmaxf(fabsf(u0.y), ay), tz = 0x1p-21f * fmaxf(fabsf(u0.z), az);
		const bool changed = have && (fabsf(next.x - incoming.x) > tx || fabsf(next.y - incoming.y) > ty || fabsf(next.z - incoming.z) > tz);
		incoming = next;
		++rounds;
		if (__syncthreads_or(changed ? 1 : 0) == 0) break;
	}
	// every lane met the hub row it assumed (to the tolerance): what it computed last stands
	if (t == cnt - 1) s_hubOut = make_float4(tr.hubOut.x, tr.hubOut.y, tr.hubOut.z, u0.w);
	float minSep = 0.0f;
	if (active)
	{
		if (MODE != 2)
		{
			cc.normalImpulse[0] = tr.imp[0]; cc.tangentImpulse[0] = tr.imp[1];
			cc.normalImpulse[1] = tr.imp[2]; cc.tangentImpulse[1] = tr.imp[3];
		}
		if (otherDynamic) rows[otherBody] = tr.otherOut;
		minSep = tr.minSep;
	}
	if (have) W.hubDelta[k] = make_float4(dx, dy, dz, 0.0f);
	if (MODE == 1 && have) lcStore(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
	if (MODE == 2) waveAtomicMaxU32(W.rootPen, r.root, floatBits(0.0f - minSep), have && active);
	__syncthreads();
	*roundsOut += rounds;
	return s_hubOut;
}

// ---- C: hub rows [first, n) of hubList lane after lane, by ONE wave ---------------------------------------------------------
// What B cannot take - the second constraint of a partner with the hub (a box in a corner touches two walls), constraints of a
// second hub, of two hubs with one another, constraints swept in order for lack of a home block. Chunks of 64: the lanes fetch
// their rows together, then take turns; every turn reads both body rows from memory and writes them back (the turns of a wave
// are ordered by its own s_waitcnt). A handful of rows per sweep where this kernel is meant to run: while an island keeps
// more than SE_LEFT_INLINE_MAX of them the host sends them to k_large_hub's eight prefetching waves instead (b2hip.hip).
#define SE_LEFT_INLINE_MAX 32
template <int MODE>
__device__ __forceinline__ void hubLeftover(const DW& W, const ContactArrays& C, int first, int n)
{
	const int lane = (int)(threadIdx.x & 63u);
	float4* rows = MODE == 2 ? W.b_pos : W.b_vel;
	for (int base = first; base < n; base += 64)
	{
		const int k = base + lane;
		const bool have = k < n;
		const int cnt = n - base < 64 ? n - base : 64;
		int row = 0;
		LargeRef r;
		r.ci = 0; r.bodyA = 0; r.bodyB = 0; r.root = 0; r.nsA = false; r.nsB = false;
		ContactConstraint cc;
		memset(&cc, 0, sizeof(cc));
		bool active = have;
		if (have)
		{
			row = W.hubList[k];
			r = largeRef(W, C, row);
			if (MODE == 2)
			{
				active = W.rootDone[r.root] == 0;
				lcLoad(W, row, cc, LC_MASS_FIRST, LC_MASS_FIRST + 4);
				lcLoad(W, row, cc, LC_POS_FIRST, LC_WORDS);
			}
			else lcLoad(W, row, cc, 0, LC_VEL_WORDS);
		}
		float minSep = 0.0f;
		for (int t = 0; t < cnt; ++t)
		{
			if (lane == t && active)
			{
				// (body A plays the hub's part in hubEvaluate: "hub" and "other" are just the constraint's two bodies here)
				const float4 ra = rows[r.bodyA], rb = rows[r.bodyB];
				const float imp0[4] = { cc.normalImpulse[0], cc.tangentImpulse[0], cc.normalImpulse[1], cc.tangentImpulse[1] };
				float4 inA = ra;
				if (MODE != 2 && !r.nsA) inA = make_float4(0, 0, 0, 0);
				const HubTrial tr = hubEvaluate(MODE, cc, imp0, true, r.nsB, inA, rb);
				if (MODE != 2)
				{
					cc.normalImpulse[0] = tr.imp[0]; cc.tangentImpulse[0] = tr.imp[1];
					cc.normalImpulse[1] = tr.imp[2]; cc.tangentImpulse[1] = tr.imp[3];
				}
				if (r.nsA) rows[r.bodyA] = make_float4(tr.hubOut.x, tr.hubOut.y, tr.hubOut.z, MODE == 2 ? ra.w : 0.0f);
				if (r.nsB) rows[r.bodyB] = make_float4(tr.otherOut.x, tr.otherOut.y, tr.otherOut.z, MODE == 2 ? rb.w : 0.0f);
				minSep = tr.minSep;
			}
			// this turn's stores before the next turn's loads
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		if (MODE == 1 && have) lcStore(W, row, cc, LC_IMP_FIRST, LC_IMP_FIRST + 4);
		if (MODE == 2) waveAtomicMaxU32(W.rootPen, r.root, floatBits(0.0f - minSep), have && active);
	}
}

// ---- D: joints, verdict of a position iteration ---------------------------------------------------------------------------
__device__ __forceinline__ void sweepEndJoints(const DW& W, const StepParams& sp, int mode)
{
	DState* S = W.st;
	const int n = S->c.nLIslands;
	for (int k = (int)threadIdx.x; k < n; k += SWEEP_END_LANES)
	{
		const int root = W.li_roots[k];
		const int nj = W.rootJoints[root];
		if (nj == 0) continue;
		if (mode == 2 && W.rootDone[root]) continue;
		const int start = W.rootJointStart[root];
		JointBodiesGlobal bodies(W);
		const int okay = b2dSolveIslandJoints(W, sp, mode, start, nj, bodies);
		if (mode == 2) W.rootJointOkay[root] = okay;
	}
}

// What one k_sweep_end launch does (a workgroup of SWEEP_END_LANES lanes); `rj`: as the hub workgroup of a fused launch.
template <int MODE>
__device__ __forceinline__ void sweepEndBody(const DW& W, const StepParams& sp, int tailFirst, int tailEnd, int what, int* stampBar, const RestJoin* rj)
{
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	// (fused launch: whatever this launch goes on to do, EVERY body that carries REST_SERIAL_BIT is settled below - it was
	// handed on instead of written to the table - the hub rows' bodies pass by pass, the joints' before the joint walk)
	// (B2HIP_SWEEP_STAMPS=1: where a velocity launch's time goes - 10 ns ticks since its start at the end of A, B, C, D in the
	// words the block solver's stamps use; they come home with the read-back as DState::stamps: tools/gpu_tumbler_probe.py)
	const unsigned long long t0 = (stampBar != nullptr && MODE == 1) ? wall_clock64() : 0ull;
#define SE_STAMP(k) do { if (stampBar != nullptr && MODE == 1 && threadIdx.x == 0) stampBar[8 + (k)] = (int)(wall_clock64() - t0); } while (0)
	// ---- A
	for (int col = tailFirst; col < tailEnd; ++col)
	{
		const int begin = W.colorStart[col], end = W.colorStart[col + 1];
		if (end <= begin) continue; // (uniform)
		sweepEndColour<MODE>(W, C, begin, end);
		__syncthreads(); // (rows written by this workgroup, read by this workgroup: one CU, one L1)
	}
	SE_STAMP(0);
	if (what & SE_HUB)
	{
		const int nRows = S->c.nHubRows;
		const int nWide = W.hubWide ? (S->c.nHubWide < nRows ? S->c.nHubWide : nRows) : 0;
		// ---- B
		if (nWide > 0)
		{
			const int hubBody = (int)(uint32_t)(W.hubMeta[0] & 0xffffffffull);
			float4* rows = MODE == 2 ? W.b_pos : W.b_vel;
			float4 u = rows[hubBody];
			int rounds = 0;
			for (int first = 0; first < nWide; first += SWEEP_END_LANES)
			{
				const int cnt = nWide - first < SWEEP_END_LANES ? nWide - first : SWEEP_END_LANES;
				bool okPass = true;
				u = hubWidePass<MODE>(W, C, hubBody, u, first, cnt, (what & SE_GUESS) ? 1 : 0, &rounds, rj, &okPass);
				if (!okPass) return; // (uniform: the vote of the whole workgroup)
			}
			if (threadIdx.x == 0)
			{
				rows[hubBody] = u;
				atomicAdd(&S->c.hubRounds, rounds);
			}
			__syncthreads();
		}
		SE_STAMP(1);
		// ---- C
		if (nWide < nRows)
		{
			if (rj != nullptr && !restSettleHubRows<MODE>(W, C, *rj, nWide, nRows - nWide)) return;
			if (!(what & SE_HUB_WIDE_ONLY))
			{
				if (threadIdx.x < 64) hubLeftover<MODE>(W, C, nWide, nRows);
				__syncthreads();
			}
		}
	}
	SE_STAMP(2);
	if (rj != nullptr)
	{
		// (a fused launch that sweeps no hub rows - an island with joints only, or B2HIP_HUB_WIDE... - still settles them all)
		if (!(what & SE_HUB) && !restSettleHubRows<MODE>(W, C, *rj, 0, S->c.nHubRows)) return;
		if (!restSettleJoints<MODE>(W, *rj)) return;
	}
	// ---- D
	if (what & SE_JOINTS_INIT) { sweepEndJoints(W, sp, 0); __syncthreads(); }
	if (what & SE_JOINTS_VEL) { sweepEndJoints(W, sp, 1); __syncthreads(); }
	if (what & SE_JOINTS_POS) { sweepEndJoints(W, sp, 2); __syncthreads(); }
	SE_STAMP(3);
#undef SE_STAMP
	if (what & SE_POS_END)
	{
		// (k_large_pos_end: per-island early out, b2Island.cpp:329-334)
		__shared__ int s_open;
		if (threadIdx.x == 0) s_open = 0;
		if (rj != nullptr)
		{
			// the verdict is over ALL rows of the iteration: the rest rows' workgroups have offered their penetration maxima
			// (atomics past the L2) before they arrive
			__shared__ int s_joined;
			if (threadIdx.x == 0)
			{
				int spins = 0, okj = 1;
				while (ldcI(&rj->bar[5]) < rj->arriveNeed)
				{
					if (++spins > (rj->bar[6] != 0 ? rj->bar[6] : PERSIST_SPIN_MAX) || ldcI(&rj->bar[4]) != 0) { stcI(&rj->bar[4], 1); atomicOr(&S->c.overflow, 64); okj = 0; break; }
					__builtin_amdgcn_s_sleep(1);
				}
				s_joined = okj;
			}
			__syncthreads();
			if (!s_joined) return;
		}
		__syncthreads();
		const int n = S->c.nLIslands;
		int open = 0;
		for (int k = (int)threadIdx.x; k < n; k += SWEEP_END_LANES)
		{
			const int root = W.li_roots[k];
			if (W.rootDone[root]) continue;
			const float minSeparation = -__uint_as_float(__hip_atomic_load(&W.rootPen[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
			if (minSeparation >= -3.0f * B2D_LINEAR_SLOP && W.rootJointOkay[root]) W.rootDone[root] = 1;
			else ++open;
		}
		if (open) atomicAdd(&s_open, open);
		__syncthreads();
		const bool allDone = s_open == 0;
		if (threadIdx.x == 0)
		{
			S->c.posItersLarge += 1;
			if (allDone) S->c.allLargeDone = 1;
		}
		if ((what & SE_POS_BEGIN) && !allDone)
		{
			// (k_large_pos_begin for the iteration that follows)
			for (int k = (int)threadIdx.x; k < n; k += SWEEP_END_LANES)
			{
				const int root = W.li_roots[k];
				W.rootPen[root] = 0;
				W.rootJointOkay[root] = 1;
			}
		}
	}
}

template <int MODE>
__global__ __launch_bounds__(SWEEP_END_LANES) void k_sweep_end(DW W, StepParams sp, int tailFirst, int tailEnd, int what, int* stampBar)
{
	b2dPhaseStamp(W);
	if (MODE == 2 && W.st->c.allLargeDone) return;
	sweepEndBody<MODE>(W, sp, tailFirst, tailEnd, what, stampBar, nullptr);
}

// ---- k_large_rest and k_sweep_end in ONE launch (round 6) ---------------------------------------------------------------------
// On the settled 100 000-box Tumbler a velocity sweep ended with k_large_rest (~28 us: chains of tagged rows, most CUs idle)
// and then k_sweep_end (~32 us: ONE workgroup, 255 CUs idle) - 60 of the sweep's ~120 us in two launches that use the device
// one after the other although almost nothing of the second depends on the first: the hub's ~900 partners are boxes at the
// container's walls, few of which carry rest colours at all. Here the LAST workgroup of the launch is k_sweep_end's (it is
// dispatched last: the rest rows' workgroups never wait for it), the others sweep the rest rows; what the hub workgroup needs
// of them arrives as tagged rows (RestJoin, REST_SERIAL_BIT). Same arithmetic, same order on every body: the bits of the two
// launches (tests/test_gpu_sweep_end.py). MODE 2: the verdict of the iteration waits for every rest workgroup's arrival.
template <int MODE>
__global__ __launch_bounds__(SWEEP_END_LANES) void k_rest_hub(DW W, StepParams sp, int restFirst, int nColors, int what, int* bar, int epoch, int arriveNeed, int* stampBar)
{
	b2dPhaseStamp(W);
	if (MODE == 2 && W.st->c.allLargeDone) return;
	const int nRest = (int)gridDim.x - 1;
	if ((int)blockIdx.x < nRest)
	{
		const bool ok = restSweepRows<MODE>(W, restFirst, nColors, bar, epoch, (int)blockIdx.x, nRest, true);
		if (MODE == 2)
		{
			// (this workgroup's penetration maxima are on their way past the L2: drained before it counts as arrived)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if (ok && threadIdx.x == 0) __hip_atomic_fetch_add(&bar[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		return;
	}
	RestJoin rj;
	rj.bar = bar;
	rj.tag = (epoch & 0x7fff) << 16;
	rj.arriveNeed = arriveNeed;
	sweepEndBody<MODE>(W, sp, 0, 0, what, stampBar, &rj);
}

// ---- a solve that can be run again (round 6; b2hip_host_phases.h: runLarge) ----------------------------------------------------------
// What the large-island solver CHANGES, saved before its first launch and put back if a wait between its workgroups timed
// out: the rows of the large islands' bodies (position + sleep time, sweep start, velocity, transform, flags, force), the
// impulses and flags of their contacts, their joints (accumulated impulses, per-step scratch). Everything else the solver
// touches is scratch that its launches fill before they read it (constraint rows, warm-start deltas, exchange rows - tagged
// with an epoch the next launch does not share) or is reset by k_solver_recover_reset. The small islands are somebody else's:
// their solver runs beside this one on the side stream and keeps what it wrote.
// dir 0: save, 1: restore.
__global__ __launch_bounds__(256) void k_solver_snapshot(DW W, int dir)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	const ContactArrays& C = W.ca[S->cur];
	const int nB = S->c.nLBodies, nC = S->c.nLContacts, nI = S->c.nLIslands;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nB; k += gridDim.x * blockDim.x)
	{
		const int body = W.li_bodies[k];
		float4* const s = W.solveSnapBody + (size_t)k * 6;
		if (dir == 0)
		{
			s[0] = W.b_pos[body]; s[1] = W.b_pos0[body]; s[2] = W.b_vel[body]; s[3] = W.b_xf[body]; s[4] = W.b_force[body];
			s[5] = make_float4(__uint_as_float(W.b_flags[body]), 0.0f, 0.0f, 0.0f);
		}
		else
		{
			W.b_pos[body] = s[0]; W.b_pos0[body] = s[1]; W.b_vel[body] = s[2]; W.b_xf[body] = s[3]; W.b_force[body] = s[4];
			W.b_flags[body] = __float_as_uint(s[5].x);
		}
	}
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nC; k += gridDim.x * blockDim.x)
	{
		const int ci = W.li_contacts[k];
		if (dir == 0) { W.solveSnapImp[k] = C.imp[ci]; W.solveSnapCFlags[k] = C.flags[ci]; }
		else { C.imp[ci] = W.solveSnapImp[k]; C.flags[ci] = W.solveSnapCFlags[k]; }
	}
	// the joints of the large islands (lj_list, island by island), and the gear records some of them own
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nI; k += gridDim.x * blockDim.x)
	{
		const int root = W.li_roots[k];
		const int nj = W.rootJoints[root], start = W.rootJointStart[root];
		for (int t = 0; t < nj; ++t)
		{
			const int j = W.lj_list[start + t];
			if (dir == 0) W.solveSnapJoints[j] = W.joints[j]; else W.joints[j] = W.solveSnapJoints[j];
			if (W.joints[j].type == B2D_JOINT_GEAR)
			{
				const int g = W.joints[j].enableLimit;
				if (dir == 0) W.solveSnapGears[g] = W.gears[g]; else W.gears[g] = W.solveSnapGears[g];
			}
		}
	}
}

// The overflow word of the step so far, to mapped host memory: word[0] the flags, word[1] the number of this publication,
// word[2] the constraints still without a colour (a resident solver reads the colours on the device).
__global__ void k_solver_status(DW W, int* word, int seq)
{
	b2dPhaseStamp(W);
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		const int flags = __hip_atomic_load(&W.st->c.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(&word[0], flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&word[2], W.st->c.nUncolored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&word[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// Before the solve runs a second time: the timed-out wait forgotten (overflow bit 6 and its qualifier, the barrier words),
// the per-step solver scratch as k_island_init / the solver's own first kernels would find it.
__global__ __launch_bounds__(256) void k_solver_recover_reset(DW W, int* bar)
{
	b2dPhaseStamp(W);
	DState* S = W.st;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W.nBodies; i += gridDim.x * blockDim.x)
	{
		for (int k = 0; k < ROOT_PEN_SLOTS; ++k) W.rootPen[(size_t)k * W.nBodies + i] = 0;
		W.rootDone[i] = 0;
		W.rootSleepMin[i] = 0x7f7fffffu;
		W.rootJointOkay[i] = 1;
		W.bodyRest[i] = 0;
		W.bodyActive[i] = 0;
	}
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MAX_BLOCKS + 1; i += gridDim.x * blockDim.x)
	{
		W.blkCursor[(size_t)i * BLK_SLOT] = 0;
		W.blkBodyCursor[(size_t)i * BLK_SLOT] = 0;
	}
	if (blockIdx.x == 0 && threadIdx.x <= MAX_COLORS) W.colorCursor[colorSlot(threadIdx.x)] = 0;
	if (blockIdx.x == 0 && threadIdx.x < 32) bar[threadIdx.x] = 0;
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		atomicAnd(&S->c.overflow, ~(64 | 0x2000));
		S->c.allLargeDone = 0;
		S->c.posItersLarge = 0;
	}
}

#endif
