"""Prints the one-step deviations of the default solver from the oracle (tests/test_gpu_onestep.py) per scene and step.
Usage (GPU box): python tools/gpu_onestep.py [scene ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh  # noqa: E402
import b2hip  # noqa: E402
import test_gpu_onestep as t  # noqa: E402

libs = (b2hip.lib(), b2hip.load(bh.ORACLE_LIB, optional_ok=True))
# STEPS=245,300 overrides the steps at which the one-step comparisons are made
if os.environ.get("STEPS"):
    for _n in (sys.argv[1:] or list(t.SCENES)):
        sc = list(t.SCENES[_n])
        sc[2] = tuple(int(x) for x in os.environ["STEPS"].split(","))
        t.SCENES[_n] = tuple(sc)
for name in (sys.argv[1:] or list(t.SCENES)):
    rep = []
    t0 = time.time()
    try:
        t.run_scene(libs, name, rep)
    except AssertionError as e:
        print("%s: ASSERT %s" % (name, e))
    for n, step, dev in rep:
        print("%s step %d: pos %.3g angle %.3g vel %.3g spin %.3g (scale %.1f, max speed %.2f, %d large-island constraints); contacts %d, set diff %d, touching diff %d, flags differ %d" % (
            n, step, dev["pos"], dev["angle"], dev["vel"], dev["spin"], dev["scale"], dev["speed_max"], dev["large_island_contacts"],
            dev["contacts"], dev["contact_set_diff"], dev["touching_diff"], dev["flags_differ"]), flush=True)
        print("    absolute: |dp| max %.3g m (p99 %.3g, p50 %.3g), |dv| max %.3g m/s (p99 %.3g), max |dv| / max(|v|, 1 m/s) %.3g" % (
            dev["pos_m"], dev["pos_m_p99"], dev["pos_m_p50"], dev["vel_mps"], dev["vel_mps_p99"], dev["vel_over_speed"]), flush=True)
    print("%s: %.1f s" % (name, time.time() - t0), flush=True)
