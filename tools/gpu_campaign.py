"""Randomised parity campaign over every scene family of the harness: random sizes / seeds / world flags, bitwise against
the C oracle every step in exact-order mode (any island size compares). usage: gpu_campaign.py [seed] [cases] [steps]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H
DEFAULT = os.environ.get("MODE") == "default"  # coloured order for large islands: compare loosely, look for errors
# MODE=tier: default mode with the reference-order tier at its 512-row limit; a case whose islands all stayed inside the
# tier (no large island in any step) must be bitwise equal to the oracle - the in-LDS solver with joints, free bodies ...
TIER = os.environ.get("MODE") == "tier"
if TIER: os.environ["B2HIP_SMALL_MAX_W"] = "512"
elif not DEFAULT: os.environ["B2HIP_FORCE_LARGE"] = "2"
import b2hip
hipL = b2hip.lib() if TIER else None
amd, orc = H.Harness(H.AMD_LIB), H.Harness(H.ORACLE_LIB)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
bad = 0
for k in range(cases):
    fam = int(rng.integers(0, 10)) if os.environ.get("FAMILIES") != "joints" else int(rng.integers(7, 10))
    seed = int(rng.integers(1, 100000))
    if fam == 0: name, scene, kw = "rain", H.RAIN, dict(p0=int(rng.integers(50, 1200)), seed=seed)
    elif fam == 1: name, scene, kw = "piles", H.PILES, dict(p0=int(rng.integers(5, 300)), p1=int(rng.integers(2, 9)), seed=seed)
    elif fam == 2: name, scene, kw = "pyramid", H.PYRAMID, dict(p0=int(rng.integers(3, 32)), p1=int(rng.integers(1, 4)))
    elif fam == 3: name, scene, kw = "circles", H.CIRCLE_STACK, dict(p0=int(rng.integers(2, 20)), p1=int(rng.integers(2, 10)))
    elif fam == 4: name, scene, kw = "tumbler", H.TUMBLER, dict(p0=int(rng.integers(4, 22)), p1=0)
    elif fam == 5: name, scene, kw = "sensors", H.SENSORS, dict(p0=int(rng.integers(10, 250)), seed=seed)
    elif fam == 7: name, scene, kw = "ropes", H.ROPES, dict(p0=int(rng.integers(0, 300)), p1=int(rng.integers(2, 18)), seed=seed)
    elif fam == 8: name, scene, kw = "machines", H.MACHINES, dict(p0=int(rng.integers(0, 300)), p1=int(rng.integers(1, 9)), seed=seed)
    elif fam == 9: name, scene, kw = "vehicles", H.VEHICLES, dict(p0=int(rng.integers(0, 300)), p1=int(rng.integers(1, 9)), seed=seed)
    else: name, scene, kw = "bullets", H.BULLETS, dict(p0=int(rng.integers(10, 200)), p1=int(rng.integers(2, 10)), seed=seed)
    flags = (H.F_CONTINUOUS if rng.random() < 0.6 else 0) | (H.F_SLEEP if rng.random() < 0.8 else 0) | (H.F_WARM if rng.random() < 0.85 else 0)
    a, o = amd.world(scene, flags=flags, **kw), orc.world(scene, flags=flags, **kw)
    first = None
    large_seen = 0
    for s in range(steps):
        a.step(1); o.step(1)
        if TIER:
            ctr = b2hip.Counters(); hipL.b2hip_get_counters(C.c_void_p(a.device_world()), C.byref(ctr))
            large_seen = max(large_seen, ctr.large_islands)
            if large_seen: break
        if DEFAULT:
            x, y = a.bodies(), o.bodies()
            if not np.isfinite(x).all() or a.contact_count == 0 and o.contact_count > 10: first = "error at %d" % s; break
            continue
        if a.contact_count != o.contact_count or not np.array_equal(a.bodies(), o.bodies()):
            first = s; break
    if DEFAULT and first is None:
        x, y = a.bodies(), o.bodies()
        dev = float(np.abs(x[:, :2] - y[:, :2]).max())
        print("         default mode: max position deviation from the oracle after %d steps %.4f, contacts %d vs %d" % (steps, dev, a.contact_count, o.contact_count))
        a.close(); o.close(); continue
    if TIER and large_seen:
        print("case %2d %-8s %s: left the tier (large island), not compared" % (k, name, kw), flush=True)
        a.close(); o.close(); continue
    if first is None:
        ia, fa, ma = a.contacts(); io, fo, mo = o.contacts()
        if not (np.array_equal(ia, io) and np.array_equal(fa, fo) and np.array_equal(ma.view(np.uint32), mo.view(np.uint32))): first = "contacts"
    print("case %2d %-8s %s flags %d: %5d bodies %6d contacts, first mismatch %s" % (k, name, kw, flags, a.body_count, a.contact_count, first), flush=True)
    bad += first is not None
    a.close(); o.close()
print("mismatching cases:", bad)
