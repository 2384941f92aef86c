"""One step of a rocprofv3 kernel trace, launch by launch: start (us since the step began), duration, idle gap before it, queue, kernel.
usage: trace_one_step.py <trace dir> [which step from the end = 2] [first us] [last us]"""
import csv, glob, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e12
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_step_begin')]
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    t = (s - t0) / 1e3
    if lo <= t <= hi:
        print("%8.1f %7.1f gap %6.1f  q%s %s" % (t, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
    prev_end = max(prev_end, e)
print("step: %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
