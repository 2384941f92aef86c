"""Phase-level bisection of run-to-run nondeterminism: two HIP worlds stepped phase by phase through the C ABI."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
hip.b2hip_step_begin.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
for n in ("b2hip_collide", "b2hip_solve", "b2hip_sync_fixtures", "b2hip_find_new_contacts", "b2hip_solve_toi", "b2hip_step_end"):
    getattr(hip, n).argtypes = [C.c_void_p]
hip.b2hip_debug_hash.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
FL = (H.F_CONTINUOUS if os.environ.get("CCD") else 0) | H.F_SLEEP | H.F_WARM
kw = dict(p0=800, p1=200, f0=50.0, f1=3.0, seed=29)


def hashes(dev):
    out = []
    for which in (0, 1, 2):
        v = C.c_uint64()
        assert hip.b2hip_debug_hash(dev, which, C.byref(v)) == 0
        out.append(v.value)
    return out


found = 0
for attempt in range(60):
    a = amd.world(H.FIELD, flags=FL, **kw); b = amd.world(H.FIELD, flags=FL, **kw)
    da, db = C.c_void_p(a.device_world()), C.c_void_p(b.device_world())
    bad = None
    for s in range(40):
        for dev in (da, db):
            assert hip.b2hip_step_begin(dev, 1.0 / 60.0, 8, 3) == 0
        for phase in ("begin", "b2hip_collide", "b2hip_solve", "b2hip_sync_fixtures", "b2hip_find_new_contacts", "b2hip_solve_toi", "b2hip_step_end"):
            if phase != "begin":
                for dev in (da, db):
                    assert getattr(hip, phase)(dev) == 0, phase
            if phase == "b2hip_step_end":
                break
            ha, hb = hashes(da), hashes(db)
            if ha != hb and bad is None:
                bad = (s + 1, phase, [x == y for x, y in zip(ha, hb)])
        if bad:
            break
    if bad:
        hip.b2hip_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]
        def trace(dev):
            out = []
            i = 0
            while True:
                lab = C.create_string_buffer(64); hv = C.c_uint64()
                if hip.b2hip_debug_trace(dev, i, lab, 64, C.byref(hv)) != 0: break
                out.append((lab.value.decode(), hv.value)); i += 1
            return out
        ta, tb = trace(da), trace(db)
        first = next((i for i, (x, y) in enumerate(zip(ta, tb)) if x != y), None)
        print("   trace len", len(ta), len(tb), "first differing stage", first, ta[first] if first is not None else None, tb[first] if first is not None else None,
              "prev", ta[first - 1][0] if first else None)
        print("attempt", attempt, "first divergence: step %d after %s; equal(bodies, contacts, proxies) = %s" % bad, flush=True)
        found += 1
    a.close(); b.close()
    if found >= 4:
        break
print("done", found)
