"""Round 6: the hub rows in the order of their partners' highest colour (k_hub_order). Prints, for the Tumbler after N steps,
the largest position difference between variants: the fixed point over the workgroup against the lane-after-lane sweep in the
SAME order, fused launch against two launches, the new order against contact order.
usage: gpu_r06_hub_order_check.py [n = 100] [steps = 40]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as bh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
amd = bh.Harness(bh.AMD_LIB)
KEYS = ("B2HIP_HUB_SERIAL", "B2HIP_HUB_ORDER", "B2HIP_NO_HUB_ORDER", "B2HIP_REST_HUB", "B2HIP_SOLVER_LAUNCHES", "B2HIP_NO_SWEEP_END", "B2HIP_NO_SWEEP_BLOCKS")
def run(env):
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    w = amd.world(bh.TUMBLER, n, 0, flags=bh.F_SLEEP | bh.F_WARM)
    w.step(steps)
    b = w.bodies().copy(); w.close()
    for k in KEYS: os.environ.pop(k, None)
    return b
L = {"B2HIP_SOLVER_LAUNCHES": "1"}
V = {"wide, ordered, fused": dict(L), "wide, ordered, two launches": dict(L, B2HIP_REST_HUB="0"),
     "serial, ordered, fused": dict(L, B2HIP_HUB_SERIAL="1", B2HIP_HUB_ORDER="1"), "serial, ordered, two launches": dict(L, B2HIP_HUB_SERIAL="1", B2HIP_HUB_ORDER="1", B2HIP_REST_HUB="0"),
     "serial, ordered, round-4 launches": dict(L, B2HIP_HUB_SERIAL="1", B2HIP_HUB_ORDER="1", B2HIP_NO_SWEEP_END="1"),
     "wide, contact order": dict(L, B2HIP_NO_HUB_ORDER="1"), "serial, contact order": dict(L, B2HIP_HUB_SERIAL="1")}
R = {k: run(v) for k, v in V.items()}
base = R["serial, ordered, two launches"]
for k, b in R.items():
    print("%-36s max |dp| vs 'serial, ordered, two launches': %.3g   finite %s" % (k, np.abs(b[:, :2] - base[:, :2]).max(), bool(np.isfinite(b).all())))
names = list(R)
print("matrix of max |dp| (rows / columns in the order above):")
for a in names:
    print("  " + " ".join("%9.2e" % np.abs(R[a][:, :2] - R[b][:, :2]).max() for b in names))
d = np.abs(R["wide, ordered, fused"][:, :2] - R["serial, ordered, fused"][:, :2]).max(axis=1)
print("wide ordered vs serial ordered: bodies off by more than 1e-4: %d of %d; the worst: body %d" % (int((d > 1e-4).sum()), len(d), int(d.argmax())))
