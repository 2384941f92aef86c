import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
hip.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(H.PYRAMID, 141, 1, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(150)
acc = np.zeros(7)
for _ in range(50):
    w.step(1)
    b = np.zeros(16, np.int32)
    hip.b2hip_debug_read(dev, 11, 0, 16, b.ctypes.data_as(C.c_void_p))
    acc += b[8:15]
acc /= 50 * 100.0  # us
names = ["integrate+barrier", "init+barrier", "velocity sweeps", "store+barrier", "integrate pos+barrier", "position iterations", "write-back"]
prev = 0.0
for n, t in zip(names, acc):
    print("%-24s %8.1f us (cumulative %8.1f)" % (n, t - prev, t)); prev = t
