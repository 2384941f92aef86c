#include <stdio.h>
#include <vector>
struct b2h_world;
extern "C" {
b2h_world* b2h_create(int scene, int p0, int p1, float f0, float f1, unsigned seed, int flags, int threads);
void b2h_destroy(b2h_world* h);
void b2h_step(b2h_world* h, int steps, float dt, int velIters, int posIters);
void b2h_record_events(b2h_world* h, int mode);
int b2h_get_events_ex(b2h_world* h, int cap, int* out);
void b2h_set_filter(b2h_world* h, int enable);
int b2h_body_count(b2h_world* h);
int b2h_contact_count(b2h_world* h);
int b2h_query_aabb(b2h_world* h, float lx, float ly, float ux, float uy, int cap, int* out);
int b2h_raycast_closest(b2h_world* h, float x1, float y1, float x2, float y2, float* out7);
void b2h_tree_stats(b2h_world* h, float* out4);
int b2h_joint_reactions(b2h_world* h, float inv_dt, int cap, float* out);
int b2h_probe_dynamic_tree(unsigned seed, int count, int ops, float* out3);
int b2h_retune_joints(b2h_world* h, int round);
int b2h_joint_anchors(b2h_world* h, int cap, float* out);
int b2h_body_joint_lists(b2h_world* h, int cap, int* out);
int b2h_wheel_states(b2h_world* h, int cap, float* out);
int b2h_rope_states(b2h_world* h, int cap, int* out);
}
int main()
{
	// every harness scene family (12 life cycle, 13 chain shapes), with the recording listener and the user filter on; queries,
	// ray casts, tree statistics (the host's shadow b2DynamicTree) and joint reactions every 20th step
	const int scenes[][3] = { {0,0,0}, {1,12,2}, {2,8,0}, {3,300,40}, {4,40,5}, {5,200,0}, {6,6,6}, {7,60,6}, {8,80,0}, {9,40,10}, {10,100,6}, {11,100,5}, {12,48,0}, {13,70,0}, {14,80,0} };
	std::vector<int> ev(10 << 18);
	for (auto& s : scenes)
	{
		// (flags 7: continuous physics on; modes 15 / 23: the listener's PreSolve switches contacts off / edits their material -
		// from inside TOI sub-steps as well)
		// (flags 15: ... and b2World::SetSubStepping - one TOI event per call)
		const int runs[][2] = { {6, 7}, {7, 7}, {7, 15}, {7, 23}, {15, 7}, {15, 15} };
		for (auto& run : runs)
		{
			const int flags = run[0];
			b2h_world* w = b2h_create(s[0], s[1], s[2], 40.0f, 2.0f, 5, flags, 1);
			if (!w) { printf("scene %d: create failed\n", s[0]); continue; }
			b2h_record_events(w, run[1]);
			if (s[0] == 12 || s[0] == 5) b2h_set_filter(w, 1);
			long total = 0;
			for (int k = 0; k < 200; ++k)
			{
				b2h_step(w, 1, 1.0f / 60.0f, 8, 3);
				total += b2h_get_events_ex(w, 1 << 18, ev.data());
				if (k % 20 == 19)
				{
					int hits[256];
					float out7[7], stats[4];
					std::vector<float> reac(4 * 4096);
					b2h_query_aabb(w, -10.0f + k * 0.1f, -1.0f, 10.0f, 12.0f, 256, hits);
					b2h_raycast_closest(w, -30.0f, 0.5f + 0.05f * k, 30.0f, 1.0f, out7);
					b2h_tree_stats(w, stats);
					b2h_joint_reactions(w, 60.0f, 4096, reac.data());
					// (late round 3: joint setters / getters, anchors, per-body joint lists)
					std::vector<int> lists(5 * 8192);
					b2h_joint_anchors(w, 4096, reac.data());
					b2h_body_joint_lists(w, 8192, lists.data());
					b2h_wheel_states(w, 1024, reac.data());
					b2h_rope_states(w, 1024, lists.data());
					if (k % 60 == 59) b2h_retune_joints(w, k / 60);
				}
			}
			printf("scene %d flags %d listener mode %d: %d bodies %d contacts %ld callbacks\n", s[0], flags, run[1], b2h_body_count(w), b2h_contact_count(w), total);
			fflush(stdout);
			b2h_destroy(w);
		}
	}
	for (unsigned seed = 1; seed <= 3; ++seed)
	{
		float out3[3];
		const int bad = b2h_probe_dynamic_tree(seed, 400, 4000, out3);
		printf("dynamic tree, seed %u: %d mismatches against brute force, height %g\n", seed, bad, out3[0]);
	}
	return 0;
}
