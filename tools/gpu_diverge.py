"""Two worlds of the bench scene stepped side by side: first step at which their states differ (run-to-run determinism)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 141
fl = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
a = amd.world(H.PYRAMID, rows, 1, flags=fl)
b = amd.world(H.PYRAMID, rows, 1, flags=fl)
first = None
for s in range(steps):
    a.step(1); b.step(1)
    if a.contact_count != b.contact_count or not np.array_equal(a.bodies(), b.bodies()):
        first = s
        x, y = a.bodies(), b.bodies()
        d = np.nonzero((x != y).any(axis=1))[0]
        print("first difference at step", s, "contacts", a.contact_count, b.contact_count, "bodies differing", len(d), d[:10].tolist())
        break
print("env", {k: v for k, v in os.environ.items() if k.startswith("B2HIP")}, "steps", steps, "first divergence", first)
