#include <stdio.h>
#include <string.h>
extern "C" int testbed_run(const char* name, int steps, double* out6);
extern "C" int testbed_entry_count();
extern "C" const char* testbed_entry_name(int i);
int main()
{
	// every registered scene of the reference's Testbed for a short run; the three with a TestPassed() predicate and the
	// ones that edit the world from callbacks for longer
	for (int i = 0; i < testbed_entry_count(); ++i)
	{
		const char* name = testbed_entry_name(i);
		int steps = 60;
		if (!strcmp(name, "SleepCollideTest")) steps = 700;
		else if (!strcmp(name, "TunnelingTest")) steps = 600;
		else if (!strcmp(name, "QueryTest")) steps = 1;
		else if (!strcmp(name, "Breakable") || !strcmp(name, "ConveyorBelt") || !strcmp(name, "MultithreadDemo") || !strcmp(name, "Car")) steps = 240;
		else if (!strncmp(name, "ManyBodies", 10) && strcmp(name, "ManyBodies6")) steps = 3; // (10 000 - 50 000 bodies on the CPU oracle)
		double out[6];
		int r = testbed_run(name, steps, out);
		printf("%s -> %d (%g bodies, %g contacts)\n", name, r, out[0], out[1]);
		fflush(stdout);
	}
	return 0;
}
