"""Large multi-island scenes: per-phase device times and solver roofline (dev script)."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh, b2hip
amd = bh.Harness(bh.AMD_LIB); L = b2hip.lib()
amd.lib.b2h_device_world.restype = C.c_void_p; amd.lib.b2h_device_world.argtypes = [C.c_void_p]
L.b2hip_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
L.b2hip_get_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_char), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)]
FL = (bh.F_CONTINUOUS if os.environ.get("CCD") else 0) | bh.F_SLEEP | bh.F_WARM
cases = [("piles 20000x5", bh.PILES, 20000, 5, 60, 40), ("piles 100000x5", bh.PILES, 100000, 5, 40, 20),
         ("field 200k", bh.FIELD, 200000, 0, 30, 20), ("pyramid 60 x 40", bh.PYRAMID, 60, 40, 60, 40),
         ("bulletfield 100k", bh.FIELD, 100000, 5000, 10, 10), ("tumbler 100", bh.TUMBLER, 100, 0, 30, 20), ("tumbler 316", bh.TUMBLER, 316, 0, 20, 10)]
if len(sys.argv) > 1: cases = [c for c in cases if sys.argv[1] in c[0]]
for name, sc, p0, p1, warm, steps in cases:
    t0 = time.time(); w = amd.world(sc, p0, p1, seed=3, flags=FL); tb = time.time() - t0
    dev = amd.lib.b2h_device_world(w.ptr)
    t0 = time.time(); w.step(warm); tw = time.time() - t0
    w.reset_profile()
    t0 = time.time(); w.step(steps); dt = (time.time() - t0) / steps
    prof = w.profile()
    L.b2hip_set_kernel_timing(dev, 1); w.step(1)
    buf = C.create_string_buffer(64); ms = C.c_float(); n = C.c_int(); by = C.c_double()
    L.b2hip_get_kernel_timing(dev, buf, 64, C.byref(ms), C.byref(n), C.byref(by))
    ctr = b2hip.Counters(); L.b2hip_get_counters(dev, C.byref(ctr))
    print("%-16s bodies %d contacts %d build %.1fs warm %.1fs  ms/step %.3f  islands S/L %d/%d  Sb/Sc %d/%d Lb/Lc %d/%d colors %d" % (
        name, w.body_count, w.contact_count, tb, tw, dt * 1e3, ctr.small_islands, ctr.large_islands, ctr.small_island_bodies,
        ctr.small_island_contacts, ctr.large_island_bodies, ctr.large_island_contacts, ctr.colors))
    print("   toi: events %d calls %d pending %d serial fallbacks %d" % (ctr.toi_events, ctr.toi_calls, ctr.toi_pending_first_pass, ctr.toi_serial_fallbacks))
    print("   phases ms:", {k: round(v, 3) for k, v in prof.items() if v and k != "steps"})
    if ms.value > 0:
        print("   kernel %s: %.1f us total in %d launches, %.1f MB algorithmic -> %.1f GB/s (%.2f%% of 8 TB/s)" % (
            buf.value.decode(), ms.value * 1e3, n.value, by.value / 1e6, by.value / ms.value / 1e6, by.value / ms.value / 1e6 / 80.0))
    w.close()
