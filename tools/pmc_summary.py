"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "").split("(")[0]
        a = acc[name][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
print("kernel,counter,dispatches,mean_per_dispatch,total")
for k in sorted(acc, key=lambda k: -max(v[0] for v in acc[k].values())):
    for c, (tot, n) in acc[k].items():
        print("%s,%s,%d,%.3f,%.1f" % (k, c, n, tot / max(n, 1), tot))
