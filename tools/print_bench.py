"""Prints the headline figures of a bench.py JSON line: print_bench.py <file>"""
import json, sys
j = json.load(open(sys.argv[1]))
print("value %.1f %s  ms/step %.4f  solver %.1f us  frac %.4f" % (j["value"], j["unit"], j["ms_per_step"], j["roofline"]["mean_launch_us"], j["roofline"]["frac"]))
print("device profile", j.get("device_profile_ms"))
for e in j.get("extra_configs", []):
    print("  %-60s %.3f ms/step" % (e["workload"][:60], e["ms_per_step"]))
if j.get("roofline_small_islands"): print("small islands", j["roofline_small_islands"]["mean_launch_us"], j["roofline_small_islands"]["frac"])
print("cpu", j.get("cpu_baseline", {}).get("value"))
