import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 316
t0 = time.time(); w = amd.world(H.TUMBLER, n, 0, flags=H.F_SLEEP | H.F_WARM); print("built", w.body_count, "%.1fs" % (time.time() - t0), flush=True)
dev = C.c_void_p(w.device_world())
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for s in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    t0 = time.time(); w.step(every); dt = (time.time() - t0) / every
    ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
    print("step", s, "%.1f ms" % (dt * 1e3), "contacts", w.contact_count, "touching", ctr.touching_contacts, "islands", ctr.islands, "L", ctr.large_islands, ctr.large_island_bodies, ctr.large_island_contacts, "colors", ctr.colors, "hub rows", ctr.hub_constraints, "rounds", ctr.hub_fixpoint_rounds, "serial chunks", ctr.hub_serial_chunks, flush=True)
b = w.bodies(); print("finite", bool(np.isfinite(b).all()))
print({k: round(v, 2) for k, v in w.profile().items() if v and k != "steps"})
