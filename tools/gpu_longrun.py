"""Long run of the bench scene (crosses the 16 384-step wrap of the mailbox tag epoch): evolution of the pile, finite,
and - with two runs - run-to-run determinism."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 17000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
res = []
for rep in range(reps):
    w = amd.world(H.PYRAMID, 141, 1, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM)
    t0 = time.time()
    for s in range(0, steps, every):
        w.step(min(every, steps - s))
        b = w.bodies()
        if rep == 0:
            print("  step %6d: contacts %d, max |x| %.2f, min y %.3f, max speed %.4f, awake %d" % (
                s + every, w.contact_count, np.abs(b[:, 0]).max(), b[1:, 1].min(), np.hypot(b[:, 3], b[:, 4]).max(), int((b[:, 9] != 0).sum()) if b.shape[1] > 9 else -1), flush=True)
    dt = time.time() - t0
    res.append((H.fnv1a64(b), w.contact_count))
    print("run %d: %d steps in %.1fs (%.3f ms/step), contacts %d, finite %s" % (rep, steps, dt, 1e3 * dt / steps, w.contact_count, bool(np.isfinite(b).all())), flush=True)
    w.close()
if reps > 1: print("deterministic:", all(r == res[0] for r in res))
