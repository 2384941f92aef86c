"""Debug run for the sub-stepping x material-editing PreSolve divergence: the campaign's worlds (seed, cases as arguments),
mode 23 only, comparing body states, callbacks AND the contacts' mixed materials after every call."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
os.environ["B2HIP_FORCE_LARGE"] = "2"
amd, orc = H.Harness(H.AMD_LIB), H.Harness(H.ORACLE_LIB)
FL = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM | H.F_SUBSTEP
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for k in range(cases):
    n = int(rng.integers(200, 1200)); bullets = int(rng.integers(20, 200)); arena = float(rng.choice([60.0, 120.0, 0.0]))
    rmax = float(rng.choice([0.0, 2.0])); seed = int(rng.integers(1, 10000)); steps = 240; mode = int(rng.choice([0, 7, 15, 23]))
    n = min(n, 500)
    if mode != 23: continue
    kw = dict(p0=n, p1=bullets, f0=arena, f1=rmax, seed=seed, flags=FL)
    a, o = amd.world(H.FIELD, **kw), orc.world(H.FIELD, **kw)
    a.record_events(mode=mode); o.record_events(mode=mode)
    hist = []
    first = None
    for s in range(steps):
        a.step(1); o.step(1)
        ea, eo = a.events_ex(), o.events_ex()
        ia, ma = a.contact_materials(); io, mo = o.contact_materials()
        ba, bo = a.bodies(), o.bodies()
        hist.append((ea, eo))
        okb = np.array_equal(ba.view(np.uint32), bo.view(np.uint32))
        oke = sorted(map(tuple, ea.tolist())) == sorted(map(tuple, eo.tolist()))
        okm = np.array_equal(ia, io) and np.array_equal(ma.view(np.uint32), mo.view(np.uint32))
        if not (okb and oke and okm):
            first = s
            print("case %d (n %d bullets %d arena %.0f rmax %.1f seed %d): call %d bodies %s events %s materials %s" % (k, n, bullets, arena, rmax, seed, s, okb, oke, okm))
            if not okb:
                bad = np.nonzero((ba.view(np.uint32) != bo.view(np.uint32)).any(axis=1))[0]
                print("  bodies differing:", bad.tolist())
                for b in bad[:6]: print("   dev", ba[b].tolist(), "\n   orc", bo[b].tolist())
            if np.array_equal(ia, io):
                badm = np.nonzero((ma.view(np.uint32) != mo.view(np.uint32)).any(axis=1))[0]
                for j in badm[:10]: print("  material differs on contact", ia[j].tolist(), "dev", ma[j].tolist(), "orc", mo[j].tolist(),
                                          "rule (lo+hi)%%5=%d (lo+2hi)%%9=%d" % ((min(ia[j][0], ia[j][2]) + max(ia[j][0], ia[j][2])) % 5, (min(ia[j][0], ia[j][2]) + 2 * max(ia[j][0], ia[j][2])) % 9))
            else: print("  contact id sets differ", len(ia), len(io))
            sa, so = set(map(tuple, ea.tolist())), set(map(tuple, eo.tolist()))
            for t in sorted(sa - so): print("  only dev:", t)
            for t in sorted(so - sa): print("  only orc:", t)
            print("  this call's callbacks in order (oracle):")
            for t in eo.tolist()[:60]: print("    ", t)
            print("  this call's callbacks in order (device):")
            for t in ea.tolist()[:60]: print("    ", t)
            # the callbacks of the three calls before, for the bodies concerned
            for back in range(1, 4):
                if s - back < 0: break
                pa, po = hist[s - back]
                print("  call %d: %d callbacks (device) %d (oracle), equal as lists: %s" % (s - back, len(pa), len(po), pa.tolist() == po.tolist()))
            dev = C.c_void_p(a.device_world()); ctr = b2hip.Counters(); b2hip.lib().b2hip_get_counters(dev, C.byref(ctr))
            print("  reruns so far:", ctr.toi_pre_solve_reruns)
            break
    if first is None: print("case %d: ok" % k)
    a.close(); o.close()
