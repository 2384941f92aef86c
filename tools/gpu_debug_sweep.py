"""Debug: k_blocks_sweep vs launch-per-colour side by side; prints the first step / bodies where they part."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as bh, b2hip
scene, p0, p1, seed, steps = [int(x) for x in sys.argv[1:6]]
amd = bh.Harness(bh.AMD_LIB)
os.environ.pop("B2HIP_SOLVER_LAUNCHES", None)
a = amd.world(scene, p0, p1, seed=seed)
os.environ["B2HIP_SOLVER_LAUNCHES"] = "1"
b = amd.world(scene, p0, p1, seed=seed)
os.environ.pop("B2HIP_SOLVER_LAUNCHES", None)
L = b2hip.lib()
def ctr(w):
    c = b2hip.Counters(); L.b2hip_get_counters(C.c_void_p(w.device_world()), C.byref(c))
    return {k: getattr(c, k) for k in ("islands", "large_islands", "large_island_bodies", "large_island_contacts", "colors", "blocks", "cut_constraints", "block_max_rows", "partitions", "sweep_solver_steps", "pos_iterations_large", "hub_constraints")}
for s in range(steps):
    a.step(1); b.step(1)
    A, B = a.bodies(), b.bodies()
    bad = np.nonzero((A.view(np.uint32) != B.view(np.uint32)).any(axis=1))[0]
    if len(bad) or a.contact_count != b.contact_count:
        print("step", s, "differing bodies", len(bad), bad[:12], "contacts", a.contact_count, b.contact_count)
        print(" sweep   ", ctr(a)); print(" launches", ctr(b))
        for i in bad[:4]:
            print("  body", i, "\n   sweep   ", A[i], "\n   launches", B[i])
        break
    if s % 20 == 0: print("step", s, "same;", ctr(a), flush=True)
else:
    print("identical for", steps, "steps", ctr(a))
