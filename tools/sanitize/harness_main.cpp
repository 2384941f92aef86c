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
}
int main()
{
	// every harness scene family, life cycle last (12), with the recording listener and the user filter on
	const int scenes[][3] = { {0,0,0}, {1,12,2}, {2,8,0}, {3,300,40}, {4,40,5}, {5,200,0}, {6,6,6}, {7,60,6}, {8,80,0}, {9,40,10}, {10,100,6}, {11,100,5}, {12,48,0} };
	std::vector<int> ev(10 << 18);
	for (auto& s : scenes)
	{
		for (int flags = 6; flags <= 7; ++flags)
		{
			b2h_world* w = b2h_create(s[0], s[1], s[2], 40.0f, 2.0f, 5, flags, 1);
			if (!w) { printf("scene %d: create failed\n", s[0]); continue; }
			b2h_record_events(w, 7);
			if (s[0] == 12 || s[0] == 5) b2h_set_filter(w, 1);
			long total = 0;
			for (int k = 0; k < 200; ++k)
			{
				b2h_step(w, 1, 1.0f / 60.0f, 8, 3);
				total += b2h_get_events_ex(w, 1 << 18, ev.data());
			}
			printf("scene %d flags %d: %d bodies %d contacts %ld callbacks\n", s[0], flags, b2h_body_count(w), b2h_contact_count(w), total);
			fflush(stdout);
			b2h_destroy(w);
		}
	}
	return 0;
}
