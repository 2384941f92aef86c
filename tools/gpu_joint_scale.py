"""Cost of jointed islands at scale: N cars (chassis + two wheels on wheel joints, one driven) each on its own strip of
ground, i.e. N small islands that all carry joints. Prints ms/step (default mode) and the per-kernel budget.
usage: gpu_joint_scale.py [cars] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2hip

cars = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = b2hip.World(gravity=(0.0, -10.0))
side = int(cars ** 0.5) + 1
for r in range(side):  # one static strip per row of cars
    g = w.create_body(b2hip.STATIC, position=(6.0 * side, 10.0 * r - 0.25))
    w.create_fixture(g, b2hip.box_shape(6.0 * side + 4.0, 0.25))
n = 0
for r in range(side):
    for c in range(side):
        if n == cars: break
        x, y = 12.0 * c + 2.0, 10.0 * r + 1.0
        ch = w.create_body(b2hip.DYNAMIC, position=(x, y))
        w.create_fixture(ch, b2hip.box_shape(1.5, 0.4), density=1.0)
        for k, dx in enumerate((-1.0, 1.0)):
            wh = w.create_body(b2hip.DYNAMIC, position=(x + dx, y - 0.6))
            w.create_fixture(wh, b2hip.circle_shape(0.4), density=1.0, friction=0.9)
            w.create_wheel_joint(ch, wh, anchor_a=(dx, -0.6), axis=(0.0, 1.0), frequency_hz=4.0, damping_ratio=0.7, enable_motor=(k == 0),
                                 motor_speed=-2.0, max_motor_torque=20.0)
        n += 1
for _ in range(30): w.step()
t0 = time.perf_counter()
for _ in range(steps): w.step()
dt = (time.perf_counter() - t0) / steps
c = w.counters()
print("cars %d bodies %d joints %d: %.3f ms/step; islands %d (small %d, large %d), contacts %d, colours %d" % (
    cars, w.body_count, 2 * cars, 1e3 * dt, c["islands"], c["small_islands"], c["large_islands"], c["contacts"], c["colors"]))
names = ["step", "collide", "solve", "solveTraversal", "solveInit", "solveVelocity", "solvePosition", "solveTOI", "solveTOIFindMinContact",
         "broadphase", "broadphaseSyncFixtures", "broadphaseFindContacts", "locking"]
print("  last step, ms: " + ", ".join("%s %.3f" % (n, v) for n, v in zip(names, w.profile()) if v > 0.0005))
# host wall-clock per phase call (the calls enqueue asynchronously; a phase that waits for the device shows it here)
import ctypes as C
L, p = w.L, w.p
acc = {}
for _ in range(40):
    for name, call in (("step_begin", lambda: L.b2hip_step_begin(p, C.c_float(1.0 / 60.0), 8, 3)), ("collide", lambda: L.b2hip_collide(p)),
                       ("solve", lambda: L.b2hip_solve(p)), ("sync_fixtures", lambda: L.b2hip_sync_fixtures(p)),
                       ("find_new_contacts", lambda: L.b2hip_find_new_contacts(p)), ("step_end", lambda: L.b2hip_step_end(p))):
        t = time.perf_counter()
        rc = call()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        assert rc == 0, (name, rc)
print("  host wall-clock per phase call, ms: " + ", ".join("%s %.3f" % (k, 1e3 * v / 40) for k, v in acc.items()))
