#include <stdio.h>
extern "C" int testbed_run(const char* name, int steps, double* out6);
int main()
{
	const char* names[] = { "SleepCollideTest", "TunnelingTest", "QueryTest", "ManyBodies6", "MultithreadDemo", "Car", "Pyramid", "Tumbler", "SleepCollidePerf" };
	const int steps[] = { 700, 600, 1, 160, 240, 240, 120, 150, 120 };
	for (int i = 0; i < 9; ++i)
	{
		double out[6];
		int r = testbed_run(names[i], steps[i], out);
		printf("%s -> %d (%g bodies, %g contacts)\n", names[i], r, out[0], out[1]);
		fflush(stdout);
	}
	return 0;
}
