"""The fixed cost of a step: HelloWorld (one box on a ground body) stepped N times through the C ABI, wall clock per step.
usage: gpu_floor.py [steps] [ccd 0/1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2hip
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ccd = len(sys.argv) > 2 and sys.argv[2] == "1"
w = b2hip.World(continuous=ccd)
g = w.create_body(b2hip.STATIC, position=(0.0, -10.0))
w.create_fixture(g, b2hip.box_shape(50.0, 10.0))
b = w.create_body(b2hip.DYNAMIC, position=(0.0, 4.0), allow_sleep=False)
w.create_fixture(b, b2hip.box_shape(1.0, 1.0), density=1.0, friction=0.3)
for _ in range(200):
    w.step(1.0 / 60.0, 6, 2)
t0 = time.perf_counter()
for _ in range(steps):
    w.step(1.0 / 60.0, 6, 2)
dt = (time.perf_counter() - t0) / steps
print("HelloWorld floor: %.1f us/step wall (ccd %s), device profile %s" % (dt * 1e6, ccd, [round(v, 4) for v in w.profile()]))
