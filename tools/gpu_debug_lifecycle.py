"""Debug: first mismatch of the life-cycle scene, device vs oracle (prints the differing rows).
usage: gpu_debug_lifecycle.py <force_large> <flags> <events 0/1> [count seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import b2harness as bh
np.set_printoptions(linewidth=200, precision=8)
os.environ["B2HIP_FORCE_LARGE"] = sys.argv[1] if len(sys.argv) > 1 else "2"
amd = bh.Harness(bh.AMD_LIB); ora = bh.Harness(bh.ORACLE_LIB)
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 6
events = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 48
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a = amd.world(bh.LIFECYCLE, count, 0, seed=seed, flags=flags); b = ora.world(bh.LIFECYCLE, count, 0, seed=seed, flags=flags)
if events:
    a.record_events(mode=7); b.record_events(mode=7)
for s in range(240):
    a.step(1); b.step(1)
    A, B = a.bodies(), b.bodies()
    MA, MB = a.mass(), b.mass()
    bad = np.nonzero((A.view(np.uint32) != B.view(np.uint32)).any(axis=1))[0]
    badm = np.nonzero((MA.view(np.uint32) != MB.view(np.uint32)).any(axis=1))[0]
    evbad = False
    if events:
        ea, eb = sorted(map(tuple, a.events_ex().tolist())), sorted(map(tuple, b.events_ex().tolist()))
        evbad = ea != eb
    ia, fa, ma = a.contacts(); ib, fb, mb = b.contacts()
    cbad = ia.shape != ib.shape or not np.array_equal(ia, ib) or not np.array_equal(fa, fb) or not np.array_equal(ma.view(np.uint32), mb.view(np.uint32))
    if cbad or len(bad) or len(badm) or evbad:
        print("step", s, "contacts", a.contact_count, b.contact_count, "bad bodies", bad[:10], "bad mass", badm[:10], "events differ", evbad)
        for i in list(bad[:4]) + list(badm[:2]):
            print(" body", i, "\n  dev", A[i], MA[i], "\n  ora", B[i], MB[i])
        sa, sb = set(map(tuple, ia.tolist())), set(map(tuple, ib.tolist()))
        print(" contacts only on device", sorted(sa - sb)[:10], "\n contacts only in oracle", sorted(sb - sa)[:10])
        if ia.shape == ib.shape and np.array_equal(ia, ib):
            d = np.nonzero((ma.view(np.uint32) != mb.view(np.uint32)).any(axis=1) | (fa != fb))[0]
            print(" contact rows differing", d[:10])
            for k in d[:3]:
                print("  ids", ia[k], "flags %x %x" % (fa[k], fb[k]), "\n   dev", ma[k], "\n   ora", mb[k])
        if evbad:
            print(" events only on device", sorted(set(ea) - set(eb))[:8], "\n events only in oracle", sorted(set(eb) - set(ea))[:8])
        break
else:
    print("240 steps identical")
