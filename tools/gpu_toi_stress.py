"""Repeatability stress of the device TOI path: the same dense CCD scenes many times, bitwise vs the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB); orc = H.Harness(H.ORACLE_LIB)
FL = (0 if os.environ.get("NO_CCD") else H.F_CONTINUOUS) | H.F_SLEEP | H.F_WARM
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cases = [("field400", H.FIELD, 40, dict(p0=400, p1=60, f0=40.0, f1=3.0, seed=5)),
         ("field800", H.FIELD, 40, dict(p0=800, p1=200, f0=50.0, f1=3.0, seed=29)),
         ("bullets150", H.BULLETS, 60, dict(p0=150, p1=8, seed=29))]
for name, scene, steps, kw in cases:
    o = (amd if os.environ.get("SELF") else orc).world(scene, flags=FL, **kw)
    want = []
    for s in range(steps):
        o.step(); want.append((o.bodies().tobytes(), o.contact_count))
    o.close()
    fails = []
    t0 = time.time()
    for rep in range(reps):
        a = amd.world(scene, flags=FL, **kw)
        for s in range(steps):
            a.step()
            if (a.bodies().tobytes(), a.contact_count) != want[s]:
                fails.append((rep, s + 1)); break
        a.close()
    print(name, "mode", os.environ.get("B2HIP_FORCE_LARGE", "default"), "reps", reps, "fails", fails, "%.1fs" % (time.time() - t0), flush=True)
