"""GPU check of the continuous-collision path: HIP world vs the CPU oracle, CCD on, bitwise."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H

amd = H.Harness(H.AMD_LIB)
orc = H.Harness(H.ORACLE_LIB)
FL = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM


def cmp(name, scene, steps, **kw):
    a = orc.world(scene, flags=FL, **kw)
    b = amd.world(scene, flags=FL, **kw)
    t0 = time.time()
    bad = None
    for s in range(steps):
        a.step(); b.step()
        A = a.bodies(); B = b.bodies()
        if A.tobytes() != B.tobytes() or a.contact_count != b.contact_count:
            bad = s + 1
            d = np.abs(A - B).max()
            idx = np.argwhere(A != B)
            print(name, "MISMATCH at step", s + 1, "maxdiff", d, "contacts", a.contact_count, b.contact_count, "first", idx[:6].tolist(), flush=True)
            break
    if bad is None:
        ca = a.contacts(); cb = b.contacts()
        okc = ca[0].shape == cb[0].shape and (ca[0] == cb[0]).all() and (ca[1] == cb[1]).all() and ca[2].tobytes() == cb[2].tobytes()
        print(name, "bit-exact", steps, "steps; contacts", a.contact_count, "contacts equal", okc, "time %.1fs" % (time.time() - t0), flush=True)
    a.close(); b.close()


cmp("hello", H.HELLO, 90)
cmp("pyramid12", H.PYRAMID, 120, p0=12)
cmp("bullets40", H.BULLETS, 200, p0=40, p1=6, seed=2)
cmp("bullets120", H.BULLETS, 200, p0=120, p1=8, seed=13)
cmp("field400_b60", H.FIELD, 150, p0=400, p1=60, f0=40.0, f1=3.0, seed=5)
cmp("rain150", H.RAIN, 150, p0=150, seed=7)
cmp("piles", H.PILES, 120, p0=25, p1=6, seed=13)
cmp("tumbler6", H.TUMBLER, 200, p0=6)
