"""Which TOI path ran (parallel chains vs serial loop) per scene, and parity vs the oracle."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
amd = H.Harness(H.AMD_LIB); orc = H.Harness(H.ORACLE_LIB)
hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
FL = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
cases = [("hello", H.HELLO, 90, {}), ("pyramid12", H.PYRAMID, 150, dict(p0=12)), ("pyramid40", H.PYRAMID, 100, dict(p0=40)),
         ("piles", H.PILES, 150, dict(p0=40, p1=6, seed=3)), ("rain300", H.RAIN, 200, dict(p0=300, seed=9)),
         ("bullets80", H.BULLETS, 150, dict(p0=80, p1=6, seed=4)), ("field300_nobullets", H.FIELD, 150, dict(p0=300, p1=0, f0=30.0, f1=2.0, seed=6)),
         ("tumbler8", H.TUMBLER, 200, dict(p0=8))]
for name, scene, steps, kw in cases:
    a = amd.world(scene, flags=FL, **kw); o = orc.world(scene, flags=FL, **kw)
    dev = C.c_void_p(a.device_world())
    events = 0; bad = None; steps_with_events = 0
    for s in range(steps):
        a.step(); o.step()
        ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
        events += ctr.toi_events; steps_with_events += ctr.toi_events > 0
        if bad is None and (a.bodies().tobytes() != o.bodies().tobytes() or a.contact_count != o.contact_count):
            bad = s + 1
    print("%-20s mode %s steps %d: toi events %d in %d steps, serial fallbacks %d, first mismatch vs oracle: %s" %
          (name, os.environ.get("B2HIP_FORCE_LARGE", "default"), steps, events, steps_with_events, ctr.toi_serial_fallbacks, bad), flush=True)
    a.close(); o.close()
