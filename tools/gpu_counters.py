"""Counters of one harness scene at a step: gpu_counters.py <scene> <p0> <p1> <steps> [ccd] [more single steps = 3]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
scene, p0, p1, steps = (int(a) for a in sys.argv[1:5])
fl = (H.F_CONTINUOUS if len(sys.argv) > 5 and sys.argv[5] == "ccd" else 0) | H.F_SLEEP | H.F_WARM
w = H.Harness(H.AMD_LIB).world(scene, p0, p1, flags=fl)
L = b2hip.lib(); dev = C.c_void_p(w.device_world())
w.step(steps)
for _ in range(3):
    w.step(1)
    c = b2hip.Counters(); L.b2hip_get_counters(dev, C.byref(c))
    print({n: getattr(c, n) for n, _ in c._fields_ if getattr(c, n)})
w.close()
