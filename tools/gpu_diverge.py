"""Run-to-run determinism: two worlds of the same scene stepped side by side, first step at which their states differ.
usage: gpu_diverge.py [steps] [case ...]   (no case = all)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
CCD = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
CASES = {
    "pyramid141": (H.PYRAMID, dict(p0=141, p1=1, flags=CCD)),
    "pyramid60x40": (H.PYRAMID, dict(p0=60, p1=40, flags=CCD)),
    "piles20000x5": (H.PILES, dict(p0=20000, p1=5, seed=3, flags=CCD)),
    "bulletfield100k": (H.FIELD, dict(p0=100000, p1=5000, seed=3, flags=CCD)),
    "densefield": (H.FIELD, dict(p0=3000, p1=400, f0=60.0, f1=3.0, seed=7, flags=CCD)),
    "tumbler100": (H.TUMBLER, dict(p0=100, p1=0, flags=H.F_SLEEP | H.F_WARM)),
    "rain2000": (H.RAIN, dict(p0=2000, seed=5, flags=CCD)),
    "bullets150": (H.BULLETS, dict(p0=150, p1=8, seed=29, flags=CCD)),
}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
names = sys.argv[2:] or list(CASES)
for name in names:
    scene, kw = CASES[name]
    a = amd.world(scene, **kw)
    b = amd.world(scene, **kw)
    first = None
    for s in range(steps):
        a.step(1); b.step(1)
        if a.contact_count != b.contact_count or not np.array_equal(a.bodies(), b.bodies()):
            first = s
            break
    print("%-16s %d bodies, %d steps: first divergence %s" % (name, a.body_count, steps, first), flush=True)
    a.close(); b.close()
