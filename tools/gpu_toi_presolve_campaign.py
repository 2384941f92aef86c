"""Randomised parity campaign for PreSolve answers that act inside TOI sub-steps (b2hip.hip: toiPreSolveRounds): fields of
free bodies with bullets, the harness's listener switching contacts off (mode 15) or editing materials (mode 23) with
continuous physics on; states, contact counts and the step's callbacks (as a set) against the C oracle every step, in
exact-order mode. Prints the number of phases that were run again per case. Third argument `substep`: the same worlds with
b2World::SetSubStepping on (one TOI event per call), listener modes 0 (none), 7 (recording), 15, 23."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
os.environ["B2HIP_FORCE_LARGE"] = "2"
amd, orc = H.Harness(H.AMD_LIB), H.Harness(H.ORACLE_LIB)
hip = b2hip.lib()
FL = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
SUB = len(sys.argv) > 3 and sys.argv[3] == "substep"
if SUB: FL |= H.F_SUBSTEP
bad = 0
for k in range(cases):
    n = int(rng.integers(200, 1200)); bullets = int(rng.integers(20, 200)); arena = float(rng.choice([60.0, 120.0, 0.0]))
    rmax = float(rng.choice([0.0, 2.0])); seed = int(rng.integers(1, 10000)); steps = 240 if SUB else 40; mode = int(rng.choice([0, 7, 15, 23] if SUB else [15, 23]))
    if SUB: n = min(n, 500)
    kw = dict(p0=n, p1=bullets, f0=arena, f1=rmax, seed=seed, flags=FL)
    a, o = amd.world(H.FIELD, **kw), orc.world(H.FIELD, **kw)
    if mode: a.record_events(mode=mode); o.record_events(mode=mode)
    dev = C.c_void_p(a.device_world())
    first = None; calls = 0
    for s in range(steps):
        a.step(1); o.step(1)
        ea, eo = (a.events_ex(), o.events_ex()) if mode else (np.zeros((0, 1)), np.zeros((0, 1)))
        calls += len(eo)
        if (a.contact_count != o.contact_count or not np.array_equal(a.bodies().view(np.uint32), o.bodies().view(np.uint32))
                or sorted(map(tuple, ea.tolist())) != sorted(map(tuple, eo.tolist()))):
            first = s; break
    ctr = b2hip.Counters(); hip.b2hip_get_counters(dev, C.byref(ctr))
    print("case %2d: %4d bodies %3d bullets arena %5.1f rmax %.1f seed %4d mode %d: %6d callbacks, %3d phases run again, first mismatch %s" % (
        k, n, bullets, arena, rmax, seed, mode, calls, ctr.toi_pre_solve_reruns, first), flush=True)
    bad += first is not None
    a.close(); o.close()
print("mismatching cases:", bad)
