"""First GPU bring-up: HelloWorld + pyramids through the C ABI vs oracle/_ref (dev script)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh
import b2hip

f32 = np.float32
ref = bh.Harness(bh.REF_LIB)

def hello():
    w = b2hip.World(continuous=False)
    g = w.create_body(b2hip.STATIC, (0.0, -10.0))
    w.create_fixture(g, b2hip.box_shape(50.0, 10.0), 0.0)
    b = w.create_body(b2hip.DYNAMIC, (0.0, 4.0))
    w.create_fixture(b, b2hip.box_shape(1.0, 1.0), 1.0, friction=0.3)
    return w

def pyramid(rows):
    w = b2hip.World(continuous=False)
    width = f32(1.125) * f32(rows) + f32(10.0)
    L = f32(0.5) * width * f32(1) + f32(40.0)
    g = w.create_body(b2hip.STATIC)
    w.create_fixture(g, b2hip.edge_shape((-L, 0.0), (L, 0.0)), 0.0)
    x0 = f32(-0.5) * width * f32(1) + width * f32(0) + f32(5.0)
    x = np.array([x0, 0.75], f32); dX = np.array([0.5625, 1.25], f32); dY = np.array([1.125, 0.0], f32)
    box = b2hip.box_shape(0.5, 0.5)
    for i in range(rows):
        y = x.copy()
        for j in range(i, rows):
            b = w.create_body(b2hip.DYNAMIC, (float(y[0]), float(y[1])))
            w.create_fixture(b, box, 5.0)
            y = (y + dY).astype(f32)
        x = (x + dX).astype(f32)
    return w

def compare(name, w, rw, steps, vi, pi):
    worst = 0.0
    t0 = time.time()
    for s in range(steps):
        w.step(1.0/60.0, vi, pi); rw.step(1, 1.0/60.0, vi, pi)
        a = w.bodies8(); b = rw.bodies()
        exact = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        d = np.abs(a[:, :6] - b[:, :6]).max()
        worst = max(worst, d)
        if s < 3 or s % 20 == 19 or not exact and s < 8:
            print(name, "step", s, "exact" if exact else "DIFF %.3g" % d, "contacts", w.contact_count, rw.contact_count,
                  "awake", int(a[:, 6].sum()), int(b[:, 6].sum()))
    print(name, "done worst", worst, "time", time.time() - t0, w.counters())
    print(name, "profile", [round(x, 3) for x in w.profile()])

w = hello(); rw = ref.world(bh.HELLO); compare("hello", w, rw, 60, 6, 2); w.close()
for rows in (3, 10, 20, 40):
    w = pyramid(rows); rw = ref.world(bh.PYRAMID, rows, 1); compare("pyr%d" % rows, w, rw, 60, 8, 3); w.close()
os.environ["B2HIP_FORCE_LARGE"] = "1"
w = pyramid(10); rw = ref.world(bh.PYRAMID, 10, 1); compare("pyr10-large", w, rw, 60, 8, 3); w.close()

w = pyramid(20); rw = ref.world(bh.PYRAMID, 20, 1); compare("pyr20-large", w, rw, 120, 8, 3); w.close()
del os.environ["B2HIP_FORCE_LARGE"]
w = pyramid(141); rw = ref.world(bh.PYRAMID, 141, 1); compare("pyr141", w, rw, 60, 8, 3)
t0 = time.time()
for i in range(100): w.step(1.0/60.0, 8, 3)
print("pyr141 ms/step wall", (time.time()-t0)*10, w.counters(), [round(x,3) for x in w.profile()], w.solver_timing())
w.close()
