"""The four timed scenes side by side, quickly (no bench bookkeeping): 1 M field from step 30, Tumbler 316 from 400, Pyramid 316 from
340, Pyramid 141 from 240 - wall clock per step over 20 steps and the device profile's main phases. For before / after looks
at a kernel change on one box; the figures of record are bench.py's."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
fl = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
for name, scene, p0, p1, settle, flags in (("field1m", 3, 1000000, 10000, 30, fl), ("tumbler316", 2, 316, 0, 400, H.F_SLEEP | H.F_WARM), ("pyramid316", 1, 316, 1, 340, fl), ("pyramid141", 1, 141, 1, 240, fl)):
    w = amd.world(scene, p0, p1, flags=flags)
    w.step(settle)
    t = time.perf_counter(); w.step(20); dt = (time.perf_counter() - t) / 20
    pr = w.profile()
    print(name, "ms/step %.3f" % (dt * 1e3), {k: round(pr[k], 3) for k in ("collide", "solve", "broadphaseSyncFixtures", "broadphaseFindContacts", "solveTOI")})
    w.close()
