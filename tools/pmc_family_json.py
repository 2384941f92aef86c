"""profiles/*_pmc_traffic.json for a kernel FAMILY (bench.py's headline roofline: the large-island solver family of the
launch-per-colour path), per STEP: from two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE; separate, as
MI355X_MICROARCH.md prescribes) over the same command, the counters of every dispatch of the family's kernels inside the last
N steps (a step begins with k_step_begin) are added up and divided by N.
usage: pmc_family_json.py <fetch dir> <write dir> <workload key> <steps>"""
import csv, glob, json, sys, collections
fetch_dir, write_dir, workload, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
FAMILY = ("k_large_integrate", "k_large_init", "k_large_velocity", "k_large_rest", "k_rest_hub", "k_large_warm", "k_sweep_end", "k_large_position", "k_large_store_impulses", "k_large_after_velocity",
          "k_large_integrate_positions", "k_large_pos_begin", "k_large_finalize", "k_large_sleep", "k_large_hub", "k_large_joints", "k_large_pos_end", "k_joints_sort")


def per_step(root, counter):
    rows = []
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
            rows.append((int(row.get("Dispatch_Id", 0) or 0), name.split("<")[0], float(row["Counter_Value"])))
    rows.sort()
    begins = [d for d, n, _ in rows if n == "k_step_begin"]
    if len(begins) <= steps:
        raise SystemExit("not enough steps in %s" % root)
    lo, hi = begins[-steps - 1], begins[-1]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for d, n, v in rows:
        if lo <= d < hi and n in FAMILY:
            acc[n][0] += 1
            acc[n][1] += v
    return {n: (c / steps, v / steps) for n, (c, v) in acc.items()}


f = per_step(fetch_dir, "FETCH_SIZE")
w = per_step(write_dir, "WRITE_SIZE")
# bytes = FETCH_SIZE * 1024 * 2 (gfx950 counts 128-B read requests at 64 B: MI355X_MICROARCH.md "HBM") + WRITE_SIZE * 1024
kernels = {}
total = 0.0
for n in sorted(set(f) | set(w)):
    fb = f.get(n, (0, 0.0))[1] * 1024 * 2
    wb = w.get(n, (0, 0.0))[1] * 1024
    kernels[n] = {"launches_per_step": round(f.get(n, w.get(n))[0], 2), "fetch_bytes_per_step": round(fb, 1), "write_bytes_per_step": round(wb, 1)}
    total += fb + wb
print(json.dumps({
    "kernel": "large-island solver family", "workload": workload, "state": "steady", "steps": steps,
    "hbm_bytes_per_step": round(total, 1), "launches_per_step": round(sum(k["launches_per_step"] for k in kernels.values()), 2), "kernels": kernels,
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over the settled workload; the counters of every "
              "dispatch of the family's kernels inside the last %d steps added up, per step; bytes = FETCH_SIZE*1024*2 (gfx950 counts 128-B read "
              "requests at 64 B: MI355X_MICROARCH.md 'HBM') + WRITE_SIZE*1024. Infinity-Cache hits are counted, and access widths other than "
              "16 B per lane are uncalibrated: an estimate of the memory-side traffic, not a byte count." % steps}, indent=1))
