"""GPU bring-up 4: both harness backends (reference vs drop-in API on HIP) on every scene (dev script)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1] if len(sys.argv) > 1 else "0"
if mode != "0": os.environ["B2HIP_FORCE_LARGE"] = mode
import b2harness as bh
ref = bh.Harness(bh.REF_LIB); amd = bh.Harness(bh.AMD_LIB)
print(ref.backend, amd.backend)
scenes = [("hello", bh.HELLO, 0, 0, 60), ("pyr12", bh.PYRAMID, 12, 1, 120), ("pyr5x3", bh.PYRAMID, 5, 3, 120),
          ("piles", bh.PILES, 40, 5, 200), ("rain", bh.RAIN, 200, 0, 200), ("cstack", bh.CIRCLE_STACK, 8, 4, 200),
          ("field", bh.FIELD, 3000, 0, 100), ("rain2k", bh.RAIN, 2000, 0, 120)]
for name, sc, p0, p1, steps in scenes:
    a = amd.world(sc, p0, p1, seed=7); r = ref.world(sc, p0, p1, seed=7)
    m0 = np.array_equal(a.mass().view(np.uint32), r.mass().view(np.uint32))
    first = None; t0 = time.time()
    for s in range(steps):
        a.step(1); r.step(1)
        A = a.bodies(); R = r.bodies()
        if first is None and not np.array_equal(A.view(np.uint32), R.view(np.uint32)):
            first = s
            d = np.abs(A - R); i = int(np.argmax(d.max(axis=1)))
            print(name, "FIRST DIFF step", s, "body", i, A[i], R[i], "nbad", int((d.max(axis=1) > 0).sum()), "contacts", a.contact_count, r.contact_count)
        if a.contact_count != r.contact_count and first is None:
            print(name, "contact count diff at", s, a.contact_count, r.contact_count)
    ia, fa, ma = a.contacts(); ir, fr, mr = r.contacts()
    same_ids = ia.shape == ir.shape and np.array_equal(ia, ir)
    same_man = same_ids and np.array_equal(ma.view(np.uint32), mr.view(np.uint32)) and np.array_equal(fa, fr)
    dmax = np.abs(a.bodies()[:, :6] - r.bodies()[:, :6]).max()
    print("%-8s mode %s bodies %d mass_exact %s first_diff %s final_maxdiff %.3g contacts %d/%d ids_equal %s manifolds_equal %s awake %d/%d  %.2fs" % (
        name, mode, a.body_count, m0, first, dmax, a.contact_count, r.contact_count, same_ids, same_man,
        int(a.bodies()[:, 6].sum()), int(r.bodies()[:, 6].sum()), time.time() - t0))
    a.close(); r.close()
