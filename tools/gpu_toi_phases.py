import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
hip.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(H.FIELD, 100000, 5000, seed=3, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(5)
acc = np.zeros(12); ev = 0
for _ in range(10):
    w.step(1)
    t = np.zeros(12, np.int32); hip.b2hip_debug_read(dev, 15, 0, 12, t.ctypes.data_as(C.c_void_p))
    ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
    acc += t; ev += ctr.toi_events
names = ["arg-min", "advance+update min contact", "gather candidates", "sort+evaluate candidates", "walk+commit", "solve island", "sync fixtures", "pair search+create", "invalidate", "gather+sync+recompute TOI"]
print("events", ev)
for n, v in zip(names, acc):
    print("%-30s %7.2f us/event" % (n, v / 100.0 / max(ev, 1)))
print("total %.2f us/event" % (acc.sum() / 100.0 / max(ev, 1)))
