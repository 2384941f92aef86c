"""profiles/*_pmc_traffic.json from two tools/pmc_summary.py tables (FETCH_SIZE pass, WRITE_SIZE pass):
pmc_traffic_json.py <dir> <kernel> <workload key> <fetch.csv> <write.csv>   (bench.py's committed_pmc_traffic reads the result)"""
import csv, json, sys
_, _, kernel, workload, fetch_csv, write_csv = sys.argv
def mean_of(path, counter):
    for line in open(path).read().splitlines()[1:]:
        # (kernel names may hold commas - template arguments - and older summaries did not quote them)
        name, ctr, n, mean, _tot = [x.strip('"') for x in line.rsplit(",", 4)]
        if kernel in name and ctr == counter:
            return float(mean), int(n)
    raise SystemExit("no %s row for %s in %s" % (counter, kernel, path))
f, n = mean_of(fetch_csv, "FETCH_SIZE")
w, _ = mean_of(write_csv, "WRITE_SIZE")
print(json.dumps({
    "kernel": kernel, "workload": workload, "state": "steady", "dispatches": n,
    "FETCH_SIZE_kb_per_dispatch": round(f, 2), "WRITE_SIZE_kb_per_dispatch": round(w, 2),
    "hbm_bytes_per_dispatch": round(f * 1024 * 2 + w * 1024, 1),
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over steps of the settled workload "
              "(tools/gpu_profile_round.sh, tools/pmc_summary.py); bytes = FETCH_SIZE*1024*2 (gfx950 counts 128-B read requests at 64 B: "
              "MI355X_MICROARCH.md 'HBM') + WRITE_SIZE*1024. Infinity-Cache hits are counted, and access widths other than 16 B per lane "
              "are uncalibrated: an estimate of the memory-side traffic, not a byte count."}, indent=1))
