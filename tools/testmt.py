"""Headless batch runner in the manner of the reference's Testbed/Framework/TestMT.cpp (:50-231): for every scene of the
harness, (1) a profile run (mean b2Profile fields over the steps), (2) the A/B consistency rule - two worlds built alike
and stepped side by side must agree on position, angle and awake flag of every body at every step -, and for scenes small
enough for the C oracle (3) a parity run against it in exact-order mode. Writes a CSV with the reference's columns.

    python tools/testmt.py [--csv out.csv] [--iterations 2] [scene ...]
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H

CCD = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
# name, scene, kwargs, steps, oracle parity?
ENTRIES = [
    ("HelloWorld", H.HELLO, dict(flags=CCD), 60, True),
    ("Pyramid 20", H.PYRAMID, dict(p0=20, p1=1, flags=CCD), 200, True),
    ("Pyramids 5x3", H.PYRAMID, dict(p0=5, p1=3, flags=CCD), 200, True),
    ("Pyramid 141 (config 2)", H.PYRAMID, dict(p0=141, p1=1, flags=CCD), 300, False),
    ("Piles 400x5", H.PILES, dict(p0=400, p1=5, seed=7, flags=CCD), 200, True),
    ("Rain 600", H.RAIN, dict(p0=600, seed=7, flags=CCD), 200, True),
    ("Circle stacks", H.CIRCLE_STACK, dict(p0=8, p1=6, flags=CCD), 200, True),
    ("Field 2000 + 200 bullets", H.FIELD, dict(p0=2000, p1=200, seed=7, flags=CCD), 100, True),
    ("Field 100k + 5k bullets", H.FIELD, dict(p0=100000, p1=5000, seed=3, flags=CCD), 60, False),
    ("Bullets room", H.BULLETS, dict(p0=80, p1=6, seed=5, flags=CCD), 200, True),
    ("Sensors", H.SENSORS, dict(p0=40, seed=5, flags=CCD), 240, True),
    ("Ropes (distance joints)", H.ROPES, dict(p0=200, p1=14, seed=9, flags=CCD), 240, True),
    ("Machines (prismatic, weld, gear, pulley)", H.MACHINES, dict(p0=200, p1=6, seed=3, flags=CCD), 300, True),
    ("Vehicles (wheel, rope, friction, motor, mouse)", H.VEHICLES, dict(p0=200, p1=6, seed=3, flags=CCD), 300, True),
    ("Tumbler 20", H.TUMBLER, dict(p0=20, p1=0, flags=H.F_SLEEP | H.F_WARM), 200, True),
    ("Tumbler 100", H.TUMBLER, dict(p0=100, p1=0, flags=H.F_SLEEP | H.F_WARM), 120, False),
    ("Chains (b2ChainShape terrain)", H.CHAINS, dict(p0=70, p1=0, seed=21, flags=CCD), 240, True),
    ("Life cycle (edits between steps)", H.LIFECYCLE, dict(p0=48, p1=0, seed=5, flags=CCD), 200, True),
    ("Pyramid 316 (config 4's share)", H.PYRAMID, dict(p0=316, p1=1, flags=CCD), 340, False),
    ("Tumbler 316 (config 3)", H.TUMBLER, dict(p0=316, p1=0, flags=H.F_SLEEP | H.F_WARM), 160, False),
]
COLS = ["step", "broadphase", "broadphaseFindContacts", "broadphaseSyncFixtures", "collide", "solve", "solveTraversal",
        "solveInit", "solvePosition", "solveVelocity", "solveTOI", "solveTOIFindMinContact", "locking"]


def consistent(amd, scene, kw, steps):
    a, b = amd.world(scene, **kw), amd.world(scene, **kw)
    bad = -1
    for s in range(steps):
        a.step(1); b.step(1)
        x, y = a.bodies(), b.bodies()
        if not (np.array_equal(x[:, :3], y[:, :3]) and np.array_equal(x[:, -1], y[:, -1])):  # position, angle, awake
            bad = s
            break
    a.close(); b.close()
    return bad


def parity(amd, orc, scene, kw, steps):
    os.environ["B2HIP_FORCE_LARGE"] = "2"  # the reference's constraint order for every island
    try:
        a, o = amd.world(scene, **kw), orc.world(scene, **kw)
        bad = -1
        for s in range(steps):
            a.step(1); o.step(1)
            if a.contact_count != o.contact_count or not np.array_equal(a.bodies(), o.bodies()):
                bad = s
                break
        a.close(); o.close()
    finally:
        os.environ.pop("B2HIP_FORCE_LARGE", None)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", default="mt_test_%s.csv" % time.strftime("%Y%m%d%H%M%S"))
    ap.add_argument("--iterations", type=int, default=1, help="consistency iterations per scene (0 = skip)")
    ap.add_argument("scenes", nargs="*")
    args = ap.parse_args()
    amd, orc = H.Harness(H.AMD_LIB), H.Harness(H.ORACLE_LIB)
    fails = incons = 0
    with open(args.csv, "w") as csv:
        csv.write("Name, Test Result, Inconsistent Index, Oracle Mismatch Index, Bodies, Contacts, ms/step (wall), " + ", ".join(COLS) + "\n")
        for name, scene, kw, steps, small in ENTRIES:
            if args.scenes and not any(s.lower() in name.lower() for s in args.scenes):
                continue
            w = amd.world(scene, **kw)
            w.step(min(steps // 4, 30))
            w.reset_profile()
            t0 = time.perf_counter()
            w.step(steps)
            wall = 1e3 * (time.perf_counter() - t0) / steps
            prof = w.profile()
            b = w.bodies()
            ok = bool(np.isfinite(b).all())
            nb, nc = w.body_count, w.contact_count
            w.close()
            bad = -1
            for _ in range(args.iterations):
                bad = consistent(amd, scene, kw, steps)
                if bad != -1:
                    break
            mis = parity(amd, orc, scene, kw, steps) if small else -1
            result = "PASS" if ok and mis == -1 else "FAIL"
            fails += result == "FAIL"
            incons += bad != -1
            print("%-28s %s  consistency %s  oracle %s  %6d bodies %7d contacts  %.3f ms/step" % (
                name, result, "PASS" if bad == -1 else "*** FAILURE on step %d ***" % bad,
                ("PASS" if mis == -1 else "*** MISMATCH on step %d ***" % mis) if small else "n/a", nb, nc, wall), flush=True)
            csv.write("%s, %s, %d, %d, %d, %d, %.3f, " % (name, result, bad, mis, nb, nc, wall) + ", ".join("%.3f" % prof.get(c, 0.0) for c in COLS) + "\n")
    print("-" * 64)
    print("Tests finished. See %s for details" % args.csv)
    print("Test result: %s" % ("Success - all tests passed" if fails == 0 else "*** FAILURE *** - %d tests failed" % fails))
    print("Consistency result: %s" % ("Success - no inconsistencies found" if incons == 0 else "*** FAILURE *** - inconsistencies found in %d tests" % incons))
    print("-" * 64)
    return 1 if fails or incons else 0


if __name__ == "__main__":
    sys.exit(main())
