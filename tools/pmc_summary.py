"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.
pmc_summary.py <dir> [last N]   - with `last N` only the last N dispatches of every kernel count (the steady-state steps
at the end of a run whose first steps build and settle the scene)."""
import csv, glob, sys, collections
root = sys.argv[1]
last = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[2] == "last" else 0
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "").split("(")[0]
        rows[name][row["Counter_Name"]].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
wr = csv.writer(sys.stdout)
wr.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "total"])
out = []
for k, counters in rows.items():
    for c, vals in counters.items():
        vals.sort()
        if last: vals = vals[-last:]
        tot = sum(v for _, v in vals)
        out.append((k, c, len(vals), tot / max(len(vals), 1), tot))
for k, c, n, mean, tot in sorted(out, key=lambda r: -r[4]):
    wr.writerow([k, c, n, "%.3f" % mean, "%.1f" % tot])
