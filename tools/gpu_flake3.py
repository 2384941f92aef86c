import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
hip.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
FL = H.F_SLEEP | H.F_WARM
kw = dict(p0=800, p1=200, f0=50.0, f1=3.0, seed=29)
NB = 801


def rd(dev, which, n, dtype, width):
    a = np.zeros((n, width), dtype)
    assert hip.b2hip_debug_read(dev, which, 0, n, a.ctypes.data_as(C.c_void_p)) == 0
    return a


found = 0
for attempt in range(80):
    a = amd.world(H.FIELD, flags=FL, **kw); b = amd.world(H.FIELD, flags=FL, **kw)
    da, db = C.c_void_p(a.device_world()), C.c_void_p(b.device_world())
    hit = False
    for s in range(40):
        a.step(); b.step()
        va, vb = rd(da, 9, NB, np.float32, 4), rd(db, 9, NB, np.float32, 4)
        if va.tobytes() != vb.tobytes():
            pa, pb = rd(da, 8, NB, np.float32, 4), rd(db, 8, NB, np.float32, 4)
            la, lb = rd(da, 10, NB + 40, np.int32, 1).reshape(-1), rd(db, 10, NB + 40, np.int32, 1).reshape(-1)
            ca, cb = la[NB:], lb[NB:]
            d = np.argwhere((va != vb).any(axis=1)).reshape(-1)
            print("attempt", attempt, "step", s + 1, "bodies with different vel after integrate:", d.tolist()[:10], "pre equal", pa.tobytes() == pb.tobytes())
            print("  counters A", ca[:34].tolist(), "\n  counters B", cb[:34].tolist())
            nLa, nLb = ca[13], cb[13]
            print("  nLBodies", nLa, nLb, "unique", len(set(la[:nLa].tolist())), len(set(lb[:nLb].tolist())), "same set", set(la[:nLa].tolist()) == set(lb[:nLb].tolist()))
            for k in d[:4]:
                print("   body", k, "preA", pa[k].tolist(), "preB", pb[k].tolist(), "A", va[k].tolist(), "B", vb[k].tolist(), "in li A", int((la[:nLa] == k).sum()), "B", int((lb[:nLb] == k).sum()))
            hit = True
            break
    a.close(); b.close()
    if hit:
        found += 1
        if found >= 3: break
print("done", found)
