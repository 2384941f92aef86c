"""The large-island solver family of the settled Tumbler (timing mode 5: one HIP event pair per step around the family) for
several values of B2HIP_REST_ROWS - how many of the top colours go through k_large_rest's data flow instead of a launch each.
usage: python tools/gpu_rest_rows_sweep.py [n = 316] [settle = 400]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh, b2hip
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 316
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
amd = bh.Harness(bh.AMD_LIB); L = b2hip.lib()
for rows in ("0", "4096", "16384", "30000", "50000", "80000", "120000", "180000"):
    os.environ.pop("B2HIP_NO_REST", None)
    if rows == "0": os.environ["B2HIP_NO_REST"] = "1"
    os.environ["B2HIP_REST_ROWS"] = rows
    w = amd.world(bh.TUMBLER, n, 0, flags=bh.F_SLEEP | bh.F_WARM)
    dev = C.c_void_p(w.device_world())
    w.step(settle)
    per = bench.time_steps(lambda: w.step(1), 30)
    roof = bench.kernel_roofline(L, dev, lambda: w.step(1), 5, 10)
    c = b2hip.Counters(); L.b2hip_get_counters(dev, C.byref(c))
    print("REST_ROWS %7s: %.3f ms per step (p50 %.3f), solver family %.3f ms, %.0f launches per step, colours %d, frac %.4f" % (rows, per.mean(), np.median(per), roof["family_ms_per_step"], roof["launches_per_step"], c.colors, roof["frac"]), flush=True)
    w.close()
