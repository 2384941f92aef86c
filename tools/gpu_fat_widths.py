"""Distribution of the fat-AABB extents the broad-phase grid is sized from, and the moved proxies per step.
usage: python tools/gpu_fat_widths.py pyramid|tumbler rows settle_steps"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh  # noqa: E402
import b2hip  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "pyramid"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 316
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 320
amd = bh.Harness(bh.AMD_LIB)
L = b2hip.lib()
if kind == "pyramid":
    w = amd.world(bh.PYRAMID, rows, 1, seed=3, flags=bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM)
else:
    w = amd.world(bh.TUMBLER, rows, 0, seed=3, flags=bh.F_SLEEP | bh.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(settle)
for s in range(5):
    w.step(1)
    ctr = b2hip.Counters()
    L.b2hip_get_counters(dev, C.byref(ctr))
    n = L.b2hip_fixture_count(dev)
    fat = np.zeros((n, 4), np.float32)
    L.b2hip_get_fat_aabbs(dev, 0, n, fat.ctypes.data_as(C.c_void_p))
    ext = np.maximum(fat[:, 2] - fat[:, 0], fat[:, 3] - fat[:, 1])
    med = float(np.median(ext))
    prof = w.profile()
    print("step %d: proxies %d, moved %d, new contacts %d; extent median %.3f p99 %.3f p99.9 %.3f max %.3f; wider than 1.5/2/3/4 x median: %s; broadphase %.3f ms findContacts %.3f ms"
          % (settle + s, n, ctr.moved_proxies, ctr.new_contacts, med, np.percentile(ext, 99), np.percentile(ext, 99.9), ext.max(),
             [int((ext > k * med).sum()) for k in (1.5, 2, 3, 4)], prof.get("broadphase", -1), prof.get("broadphaseFindContacts", -1)))
