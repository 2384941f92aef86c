"""A/B of environment switches on one harness scene, same process, same box: wall-clock ms per step over a timed window behind
a settle. usage: gpu_ab.py <scene id> <p0> <p1> <settle> <timed> <ccd 0|1> [ENV=VAL[,ENV=VAL] ...]   ("-" = no switch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
import numpy as np
amd = H.Harness(H.AMD_LIB)
scene, p0, p1, settle, timed, ccd = (int(a) for a in sys.argv[1:7])
variants = sys.argv[7:] or ["-"]
fl = (H.F_CONTINUOUS if ccd else 0) | H.F_SLEEP | H.F_WARM
keys = set()
for v in variants:
    if v != "-":
        keys.update(kv.split("=")[0] for kv in v.split(","))
for rep in range(2):
    for v in variants:
        for k in keys: os.environ.pop(k, None)
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("="); os.environ[k] = val
        w = amd.world(scene, p0, p1, flags=fl)
        w.step(settle)
        ts = []
        for _ in range(timed):
            t = time.perf_counter(); w.step(1); ts.append(time.perf_counter() - t)
        ts = np.array(ts) * 1e3
        print("%-40s mean %.3f ms  p50 %.3f  p99 %.3f   hash %s" % (v, ts.mean(), np.percentile(ts, 50), np.percentile(ts, 99), H.fnv1a64(w.bodies())), flush=True)
        w.close()
