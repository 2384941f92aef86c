"""1 M-body field (config 5 on one GPU): ms per step with the state read-back at every step and on demand.
usage: python tools/gpu_lazy_field.py [bodies] [bullets] [steps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh  # noqa: E402
import b2hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
bullets = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
amd = bh.Harness(bh.AMD_LIB)
L = b2hip.lib()
w = amd.world(bh.FIELD, n, bullets, seed=3, flags=bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(30)
for lazy in (0, 1, 0, 1):
    L.b2hip_set_lazy_readback(dev, lazy)
    t = time.perf_counter()
    w.step(steps)
    t1 = time.perf_counter()
    w.bodies()
    t2 = time.perf_counter()
    print("%d bodies, lazy=%d: %.3f ms/step, then the states in %.3f ms" % (w.body_count, lazy, 1000 * (t1 - t) / steps, 1000 * (t2 - t1)))
