"""Steady-state per-step kernel budget from a rocprofv3 --kernel-trace CSV: only the last N steps (a step starts at
k_step_begin), per kernel: launches per step, mean duration, time per step; plus the idle time between kernels."""
import csv, glob, sys, collections
root, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
begins = [i for i, r in enumerate(rows) if r[2] == "k_step_begin"]
if len(begins) <= nsteps:
    sys.exit("not enough steps in the trace")
lo, hi = begins[-nsteps - 1], begins[-1]
sel = rows[lo:hi]
acc = collections.defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in sel:
    acc[n][0] += 1
    acc[n][1] += e - s
    busy += e - s
span = rows[hi][0] - rows[lo][0]
print("steps %d: span %.1f us/step, kernels busy %.1f us/step, idle between kernels %.1f us/step, %.1f launches/step"
      % (nsteps, span / nsteps / 1e3, busy / nsteps / 1e3, (span - busy) / nsteps / 1e3, len(sel) / nsteps))
print("%-44s %10s %12s %12s" % ("kernel", "per step", "mean us", "us/step"))
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %10.2f %12.2f %12.2f" % (n[:44], c / nsteps, t / c / 1e3, t / nsteps / 1e3))
if len(sys.argv) > 3:
    # the last steps in launch order: name, start offset inside the step, duration
    for si in range(2):
        a, b = begins[-3 + si], begins[-2 + si]
        t0 = rows[a][0]
        print("---- step", si)
        for s, e, n in rows[a:b]:
            print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:60]))
