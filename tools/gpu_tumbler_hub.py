import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
n = int(sys.argv[1]); chunks = int(sys.argv[2]); every = int(sys.argv[3])
w = amd.world(H.TUMBLER, n, 0, flags=H.F_SLEEP | H.F_WARM)
dev = C.c_void_p(w.device_world())
for s in range(chunks):
    t0 = time.time(); w.step(every); dt = (time.time() - t0) / every
    ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
    b = w.bodies()
    print("after %d steps: %.2f ms/step contacts %d touching %d islands %d large %d/%d colors %d finite %s max|pos| %.1f" % (
        (s + 1) * every, dt * 1e3, w.contact_count, ctr.touching_contacts, ctr.islands, ctr.large_island_bodies, ctr.large_island_contacts, ctr.colors,
        bool(np.isfinite(b).all()), float(np.abs(b[:, :2]).max())), flush=True)
    if dt > 0.5: print("too slow, stopping"); break
