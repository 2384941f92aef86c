"""Wall-clock ms per step of one harness scene over a long run, as means over windows (what a bench window is worth: round 6).
usage: gpu_step_series.py <scene> <p0> <p1> <steps> <window> [ccd]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
scene, p0, p1, steps, win = (int(a) for a in sys.argv[1:6])
fl = (H.F_CONTINUOUS if len(sys.argv) > 6 else 0) | H.F_SLEEP | H.F_WARM
w = H.Harness(H.AMD_LIB).world(scene, p0, p1, flags=fl)
ts, cc = [], []
for s in range(steps):
    t = time.perf_counter(); w.step(1); ts.append(1e3 * (time.perf_counter() - t))
    if (s + 1) % win == 0:
        a = np.array(ts[-win:])
        print("steps %4d..%4d: mean %.3f ms p50 %.3f p99 %.3f  contacts %d" % (s + 1 - win, s, a.mean(), np.median(a), np.percentile(a, 99), w.contact_count), flush=True)
w.close()
