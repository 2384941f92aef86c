"""Randomised parity campaign for the continuous-collision paths: fields of free bodies with bullets at several densities
and seeds, bitwise against the C oracle every step (exact-order mode, so that dense fields with large islands compare too).
Prints which path ran (whole-phase fallbacks) per case."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
os.environ["B2HIP_FORCE_LARGE"] = "2"
amd, orc = H.Harness(H.AMD_LIB), H.Harness(H.ORACLE_LIB)
hip = b2hip.lib()
FL = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for k in range(cases):
    n = int(rng.integers(600, 2600)); bullets = int(rng.integers(50, 500)); arena = float(rng.choice([35.0, 60.0, 120.0, 0.0]))
    rmax = float(rng.choice([0.0, 2.0, 3.0])); seed = int(rng.integers(1, 10000)); steps = 30
    # (default radii in the smallest arena: every body overlaps a hundred others - 200 000 contacts among 2 000 bodies, half a
    # minute per step in the oracle itself and minutes in the device's serial TOI replay; keep that combination small)
    # (at 700 bodies it still runs into the documented scratch limit of a TOI event - 256 candidate contacts on the two seed
    # bodies, B2HIP_ERR_CAPACITY - on some seeds)
    if rmax == 0.0 and arena == 35.0: n = min(n, 400)
    # (2 185 bodies of radius up to 3 in the same arena: 70 000 contacts, 32 per body - one TOI event displaces bodies with more
    # than the 512 contacts whose impacts can be recomputed per event: B2HIP_ERR_CAPACITY, the documented limit, on step 0)
    if rmax >= 3.0 and arena == 35.0: n = min(n, 1500)
    kw = dict(p0=n, p1=bullets, f0=arena, f1=rmax, seed=seed, flags=FL)
    a, o = amd.world(H.FIELD, **kw), orc.world(H.FIELD, **kw)
    dev = C.c_void_p(a.device_world())
    first = None; events = 0
    for s in range(steps):
        a.step(1); o.step(1)
        ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr)); events += ctr.toi_events
        if a.contact_count != o.contact_count or not np.array_equal(a.bodies(), o.bodies()):
            first = s; break
    print("case %2d: %4d bodies %3d bullets arena %5.1f rmax %.1f seed %4d: %4d events, %d whole-phase fallbacks, first mismatch %s" % (
        k, n, bullets, arena, rmax, seed, events, ctr.toi_serial_fallbacks, first), flush=True)
    bad += first is not None
    a.close(); o.close()
print("mismatching cases:", bad)
