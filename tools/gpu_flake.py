"""Hunt run-to-run nondeterminism: two HIP worlds of the same dense scene in lockstep."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
FL = H.F_SLEEP | H.F_WARM
kw = dict(p0=800, p1=200, f0=50.0, f1=3.0, seed=29)
found = 0
for attempt in range(80):
    a = amd.world(H.FIELD, flags=FL, **kw); b = amd.world(H.FIELD, flags=FL, **kw)
    for s in range(40):
        a.step(); b.step()
        A = a.bodies(); B = b.bodies()
        ia, fa, ma = a.contacts(); ib, fb, mb = b.contacts()
        same_set = ia.shape == ib.shape and (ia == ib).all()
        same_man = same_set and ma.tobytes() == mb.tobytes() and (fa == fb).all()
        if A.tobytes() != B.tobytes() or not same_man:
            d = np.argwhere((A != B).any(axis=1)).reshape(-1)
            ctr = b2hip.Counters(); hip.b2hip_get_counters(C.c_void_p(a.device_world()), C.byref(ctr))
            print("attempt", attempt, "step", s + 1, "bodies differing", len(d), d[:10].tolist(), "maxabs", float(np.abs(A - B).max()),
                  "contacts", a.contact_count, b.contact_count, "same set", same_set, "same manifolds", same_man,
                  "islands", ctr.islands, "large", ctr.large_islands, ctr.large_island_bodies, ctr.large_island_contacts, "colors", ctr.colors, flush=True)
            if same_set and not same_man:
                bad = np.argwhere((ma.view(np.uint32) != mb.view(np.uint32)).any(axis=1) | (fa != fb)).reshape(-1)
                print("   manifold diffs", len(bad), ia[bad[:5]].tolist())
                for k in bad[:2]:
                    print("    ", ma[k].tolist(), "\n    ", mb[k].tolist())
            for k in d[:3]:
                print("   body", k, A[k].tolist(), "\n        ", B[k].tolist())
            found += 1
            break
    a.close(); b.close()
    if found >= 3: break
print("done, found", found)
