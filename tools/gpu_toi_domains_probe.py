"""Config 5 on one GPU: what the component-wise TOI event loops (k_toi_domains) spend their time on.
Per component the kernel leaves its phase ticks (10 ns), events, TOI calls and pending contacts in DW::hubList
(b2d_kernels_toi.h: toiLoopRun); this prints the distribution over the components of a few steps and the phase budget of
the longest ones. usage: python tools/gpu_toi_domains_probe.py [bodies] [bullets] [settle] [steps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh  # noqa: E402
import b2hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
bullets = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 40
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
amd = bh.Harness(bh.AMD_LIB)
L = b2hip.lib()
L.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(bh.FIELD, n, bullets, seed=3, flags=bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(settle)
PH = ["recompute+min", "seed update", "gather cand", "cand update", "walk+commit", "island solve", "sync fixtures", "find new", "invalidate", "collect recomp"]
cap = 4096
for s in range(steps):
    # wipe the diagnostics area, step, read it back
    w.step(1)
    buf = np.zeros(16 * (cap + 1), dtype=np.int32)
    rc = L.b2hip_debug_read(dev, 15, 0, len(buf), buf.ctypes.data_as(C.c_void_p))
    if rc:
        print("debug_read failed", rc); break
    dom = buf[16:].reshape(cap, 16)
    ev = dom[:, 12]
    # (stale rows of earlier steps stay behind the live ones: a step with fewer components shows some of the last step's - the
    # counters tell how many are live)
    nd = int(buf[12])
    nd = min(nd, cap)
    d = dom[:nd]
    tot = d[:, :10].sum(axis=1) * 0.01
    print("step %d: %d components, events %d (max %d per component), pending contacts max %d; component time us: mean %.1f p50 %.1f p99 %.1f max %.1f" % (
        settle + s, nd, int(d[:, 12].sum()), int(d[:, 12].max(initial=0)), int(d[:, 14].max(initial=0)), tot.mean() if nd else 0,
        np.percentile(tot, 50) if nd else 0, np.percentile(tot, 99) if nd else 0, tot.max(initial=0)))
    hist = np.bincount(d[:, 12], minlength=1)
    print("   events per component:", {i: int(c) for i, c in enumerate(hist) if c})
    for k in np.argsort(-tot)[:3]:
        print("   component %5d: %2d events, %3d calls, %3d pending, %.1f us = " % (k, d[k, 12], d[k, 13], d[k, 14], tot[k]) +
              ", ".join("%s %.1f" % (PH[i], d[k, i] * 0.01) for i in range(10)))
    if nd:
        print("   all components, us per phase summed / events: " + ", ".join("%s %.2f" % (PH[i], d[:, i].sum() * 0.01 / max(1, d[:, 12].sum())) for i in range(10)))
print({k: round(v, 3) for k, v in w.profile().items() if v and k != "steps"})
w.close()
