import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 316
t0 = time.time()
w = amd.world(H.PYRAMID, rows, 1, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM)
print("built", w.body_count, "bodies in %.1fs" % (time.time() - t0), flush=True)
dev = C.c_void_p(w.device_world())
t0 = time.time(); w.step(60); print("60 warm-up steps %.2fs" % (time.time() - t0), flush=True)
w.reset_profile()
t0 = time.time(); w.step(100); dt = (time.time() - t0) / 100
ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
print("ms/step %.3f  contacts %d  large bodies/contacts %d/%d colors %d toi events %d fallbacks %d" % (dt * 1e3, w.contact_count, ctr.large_island_bodies, ctr.large_island_contacts, ctr.colors, ctr.toi_events, ctr.toi_serial_fallbacks))
print({k: round(v, 3) for k, v in w.profile().items() if v and k != "steps"})
b = w.bodies()
print("finite", bool(np.isfinite(b).all()), "min y %.3f max |x| %.1f" % (b[1:, 1].min(), np.abs(b[:, 0]).max()))
