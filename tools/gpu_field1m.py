import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
n = int(sys.argv[1]); bullets = int(sys.argv[2]); steps = int(sys.argv[3])
t0 = time.time(); w = amd.world(H.FIELD, n, bullets, seed=3, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM); print("built", w.body_count, "%.1fs" % (time.time() - t0), flush=True)
dev = C.c_void_p(w.device_world())
for s in range(steps):
    t0 = time.time(); w.step(1); dt = time.time() - t0
    ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
    print("step %d: %.1f ms contacts %d touching %d islands %d toi events %d calls %d pending %d fallbacks %d" % (s, dt * 1e3, w.contact_count, ctr.touching_contacts, ctr.islands, ctr.toi_events, ctr.toi_calls, ctr.toi_pending_first_pass, ctr.toi_serial_fallbacks), flush=True)
    if dt > 2.0: print("too slow, stopping"); break
print({k: round(v, 2) for k, v in w.profile().items() if v and k != "steps"})
b = w.bodies(); print("finite", bool(np.isfinite(b).all()))
