import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import b2harness as H
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
hip.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(H.PYRAMID, 141, 1, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(250)
cc = np.zeros(65, np.int32); hip.b2hip_debug_read(dev, 12, 0, 65, cc.ctypes.data_as(C.c_void_p))
print("constraints per colour:", cc[:16].tolist())
m = np.zeros(w.body_count, np.uint64); hip.b2hip_debug_read(dev, 13, 0, w.body_count, m.ctypes.data_as(C.c_void_p))
pop = np.array([bin(int(x)).count("1") for x in m])
print("reserved colours per body: max", pop.max(), "hist", np.bincount(pop).tolist())
hi = np.array([int(x).bit_length() for x in m])
print("highest colour per body hist", np.bincount(hi).tolist())
