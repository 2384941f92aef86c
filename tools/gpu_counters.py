import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
hip.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(H.PYRAMID, 141, 1, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(200)
names = ["nContacts","nDestroy","nTouching","nMoves","nPairs","nNewContacts","nRoots","nSIslands","nSBodies","nSContacts","nSW","nChunks","nLIslands","nLBodies","nLContacts","nColors","nUncolored","colorRounds","nLargeProxies","posItersLarge","allLargeDone","needRecolor","maxSmallW","chunkW","overflow","nIslands","nToiList","nToiEvents","nToiCalls","toiBase","toiOverflow","nToiOrder","nToiDestroy","toiUnsafe","nToiGroups","nToiMoved","nUncolList"]
for s in range(5):
    w.step(1)
    c = np.zeros(40, np.int32)
    hip.b2hip_debug_read(dev, 7, 0, 40, c.ctypes.data_as(C.c_void_p))
    print({n: int(v) for n, v in zip(names, c) if v})
    ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr)); print("   moved", ctr.moved_proxies, "new", ctr.new_contacts, "destroyed", ctr.destroyed_contacts)
