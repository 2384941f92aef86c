"""GPU bring-up 3: locate the first divergence from the reference (dev script)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "0"
if mode != "0": os.environ["B2HIP_FORCE_LARGE"] = mode
src = open(os.path.join(ROOT, "tools", "gpu_smoke1.py")).read().split("w = hello(); rw")[0]
exec(src)

def trace(name, w, rw, steps, vi=8, pi=3):
    first = None
    for s in range(steps):
        w.step(1.0/60.0, vi, pi); rw.step(1, 1.0/60.0, vi, pi)
        a = w.bodies8(); b = rw.bodies()
        if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            d = np.abs(a - b)
            i = int(np.argmax(d.max(axis=1)))
            print(name, "FIRST DIFF at step", s, "body", i, "mine", a[i], "ref", b[i], "nbad", int((d.max(axis=1) > 0).sum()),
                  "contacts", w.contact_count, rw.contact_count, w.counters())
            # which contacts differ
            mc = w.contacts(); ids, fl, man = rw.contacts()
            mine = set((int(c["body_a"]), int(c["body_b"])) for c in mc); refs = set((int(x[0]), int(x[2])) for x in ids)
            print(name, "contact set diff mine-ref", sorted(mine - refs)[:10], "ref-mine", sorted(refs - mine)[:10])
            first = s
            break
    print(name, "steps", steps, "first diff", first, "contacts", w.contact_count, rw.contact_count)

for rows in (10, 20, 30, 40):
    w = pyramid(rows); rw = ref.world(bh.PYRAMID, rows, 1); trace("pyr%d-mode%s" % (rows, mode), w, rw, 200); w.close()
