import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB); orc = H.Harness(H.ORACLE_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
FL = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
kw = dict(p0=800, p1=200, f0=50.0, f1=3.0, seed=29)
a = amd.world(H.FIELD, flags=FL, **kw); o = orc.world(H.FIELD, flags=FL, **kw)
dev = C.c_void_p(a.device_world())
for s in range(40):
    a.step(); o.step()
    ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
    A = a.bodies(); O = o.bodies()
    ia, fa, ma = a.contacts(); io, fo, mo = o.contacts()
    sa = set(map(tuple, ia.tolist())); so = set(map(tuple, io.tolist()))
    print("step", s + 1, "contacts", a.contact_count, o.contact_count, "bodies equal", A.tobytes() == O.tobytes(),
          "toi events", ctr.toi_events, "calls", ctr.toi_calls, "pending", ctr.toi_pending_first_pass, "ovf", ctr.overflow_flags)
    if sa == so:
        bad = np.argwhere((ma.view(np.uint32) != mo.view(np.uint32)).any(axis=1) | (fa != fo)).reshape(-1)
        print("  contact records differing:", len(bad))
        for k in bad[:4]:
            print("   ", ia[k].tolist(), "flags", fa[k], fo[k], "\n     amd", ma[k].tolist(), "\n     orc", mo[k].tolist())
    if sa != so:
        print("  only oracle:", sorted(so - sa)[:10], " only amd:", sorted(sa - so)[:10])
        for (fA, fB, bA, bB) in sorted(so - sa)[:3]:
            print("   bodies", bA, O[bA], "|", bB, O[bB])
            print("   amd   ", bA, A[bA], "|", bB, A[bB])
            fat = np.zeros(4, np.float32)
            for f in (fA, fB):
                hip.b2hip_get_fat_aabb(dev, f, fat.ctypes.data_as(C.POINTER(C.c_float)))
                print("   amd fat", f, fat.tolist())
        d = np.argwhere(A != O)
        print("  body diffs", d[:8].tolist())
        break
