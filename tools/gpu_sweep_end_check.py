"""k_sweep_end against round 4's launch sequence (tests/test_gpu_sweep_end.py is the pytest form of the same comparisons):
  1. folding only (tail colours, leftover hub rows lane after lane, joints, position verdict) - with the hub rows swept lane
     after lane on both sides (B2HIP_HUB_SERIAL=1) the states must be the same bits as with B2HIP_NO_SWEEP_END=1;
  2. the hub rows as one fixed point against the lane-after-lane sweep: a tolerance (the scene is chaotic);
  3. run-to-run determinism of the default;  4. ms per step of the settled 100 000-box Tumbler, default against round 4.
usage: python tools/gpu_sweep_end_check.py [quick]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh
amd = bh.Harness(bh.AMD_LIB)
KEYS = ("B2HIP_HUB_SERIAL", "B2HIP_NO_SWEEP_END", "B2HIP_NO_TAIL", "B2HIP_HUB_WIDE", "B2HIP_TAIL_ROWS", "B2HIP_SOLVER_LAUNCHES", "B2HIP_NO_BLOCKS", "B2HIP_NO_REST", "B2HIP_REST_ROWS", "B2HIP_NO_BODY_WARM")


def run(scene, p0, p1, steps, env, flags=bh.F_SLEEP | bh.F_WARM, seed=3, every=1):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    w = amd.world(scene, p0, p1, seed=seed, flags=flags)
    out = []
    for s in range(steps):
        w.step(1)
        if s % every == every - 1:
            out.append((bh.fnv1a64(w.bodies()), w.contact_count))
    b = w.bodies()
    w.close()
    for k in KEYS:
        os.environ.pop(k, None)
    return out, b


def first_diff(a, b):
    return next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), None)


ok = True
ccd = bh.F_SLEEP | bh.F_WARM | bh.F_CONTINUOUS
L = {"B2HIP_SOLVER_LAUNCHES": "1"}
cases = [("tumbler60", bh.TUMBLER, 60, 0, 120, bh.DEFAULT_FLAGS, {}), ("tumbler60 launch per colour", bh.TUMBLER, 60, 0, 120, bh.DEFAULT_FLAGS, L),
         ("tumbler100", bh.TUMBLER, 100, 0, 160, bh.DEFAULT_FLAGS, {}),
         ("pyramid90 launch per colour", bh.PYRAMID, 90, 1, 150, ccd, L),
         ("vehicles buried", bh.VEHICLES, 700, 5, 200, bh.DEFAULT_FLAGS, {}), ("vehicles buried, launch per colour", bh.VEHICLES, 700, 5, 200, bh.DEFAULT_FLAGS, L),
         ("machines buried", bh.MACHINES, 600, 6, 200, bh.DEFAULT_FLAGS, {}), ("machines buried, launch per colour", bh.MACHINES, 600, 6, 200, bh.DEFAULT_FLAGS, L)]
for name, scene, p0, p1, steps, fl, extra in cases:
    a, _ = run(scene, p0, p1, steps, dict(extra, B2HIP_HUB_SERIAL="1", B2HIP_NO_SWEEP_END="1"), fl)
    b, _ = run(scene, p0, p1, steps, dict(extra, B2HIP_HUB_SERIAL="1"), fl)
    c, _ = run(scene, p0, p1, steps, dict(extra, B2HIP_HUB_SERIAL="1", B2HIP_NO_REST="1", B2HIP_TAIL_ROWS="100"), fl)
    d, _ = run(scene, p0, p1, steps, dict(extra, B2HIP_HUB_SERIAL="1", B2HIP_REST_ROWS="100000"), fl)
    e, _ = run(scene, p0, p1, steps, dict(extra, B2HIP_HUB_SERIAL="1", B2HIP_NO_REST="1", B2HIP_TAIL_ROWS="100000"), fl)
    ds = [first_diff(a, x) for x in (b, c, d, e)]
    print("[1] %-36s == round-4 launches: %s (first difference: default %s / no rest, tail of 100 rows %s / rest of 28 000 rows %s / no rest, whole tail %s)" % (name, all(x is None for x in ds), *ds), flush=True)
    ok = ok and all(x is None for x in ds)
for n, steps in ((60, 40), (100, 40)):
    _, bs = run(bh.TUMBLER, n, 0, steps, {"B2HIP_HUB_SERIAL": "1"})
    _, bw = run(bh.TUMBLER, n, 0, steps, {})
    _, bo = run(bh.TUMBLER, n, 0, steps, {"B2HIP_NO_SWEEP_END": "1"})
    dw, do = np.abs(bs[:, :2] - bw[:, :2]).max(), np.abs(bs[:, :2] - bo[:, :2]).max()
    print("[2] tumbler%d after %d steps: wide fixed point vs lane after lane %.2e; round 4's chunks of 64 vs lane after lane %.2e" % (n, steps, dw, do), flush=True)
    ok = ok and np.isfinite(bw).all() and dw < 1e-3
a, _ = run(bh.TUMBLER, 100, 0, 200, {})
b, _ = run(bh.TUMBLER, 100, 0, 200, {})
print("[3] tumbler100 default, 200 steps twice: identical %s" % (a == b), flush=True)
ok = ok and a == b
if len(sys.argv) < 2:
    import ctypes as C, b2hip
    L = b2hip.lib()
    for env in ({}, {"B2HIP_REST_ROWS": "16384"},
                {"B2HIP_NO_BODY_WARM": "1"}, {"B2HIP_NO_REST": "1"}, {"B2HIP_NO_SWEEP_END": "1"}):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        w = amd.world(bh.TUMBLER, 316, 0, flags=bh.F_SLEEP | bh.F_WARM)
        w.step(400)
        per = []
        for _ in range(40):
            t0 = time.perf_counter(); w.step(1); per.append(1e3 * (time.perf_counter() - t0))
        c = b2hip.Counters(); L.b2hip_get_counters(C.c_void_p(w.device_world()), C.byref(c))
        print("[4] tumbler316 steps 400..439 %-28s mean %.3f ms p50 %.3f  colours %d hub rows %d fixed-point rounds per step %d profile %s" % (env or "default", np.mean(per), np.median(per), c.colors, c.hub_constraints, c.hub_fixpoint_rounds, {k: round(v, 3) for k, v in w.profile().items() if k in ("solve", "solveVelocity", "solveTraversal")}), flush=True)
        w.close()
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
