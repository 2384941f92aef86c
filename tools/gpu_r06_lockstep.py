"""Round 6: two variants of one scene, one after the other (the switches are read from the environment whenever buffers grow, so
each variant keeps its environment for its whole run); prints per step the number of bodies whose state differs.
usage: gpu_r06_lockstep.py <scene> <p0> <p1> <steps> "ENV=V,ENV=V" "ENV=V,..." """
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as bh
scene, p0, p1, steps = (int(a) for a in sys.argv[1:5])
amd = bh.Harness(bh.AMD_LIB)
runs = []
for spec in sys.argv[5:7]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv and kv != "-")
    os.environ.update(env)
    w = amd.world(scene, p0, p1, flags=bh.F_SLEEP | bh.F_WARM)
    out = []
    for s in range(steps):
        w.step(1); out.append((w.bodies().copy(), w.contact_count))
    w.close()
    for k in env: os.environ.pop(k, None)
    runs.append(out)
shown = 0
for s in range(steps):
    (a, ca), (b, cb) = runs[0][s], runs[1][s]
    d = np.abs(a - b).max(axis=1)
    nd = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=1).sum())
    if nd and shown < 12:
        shown += 1
        i = int(d.argmax())
        print("step %d: %d bodies differ, max |d| %.3g at body %d: %s vs %s; contacts %d / %d" % (s, nd, d.max(), i, a[i][:6], b[i][:6], ca, cb), flush=True)
print("done: last step %d bodies differ" % nd)
