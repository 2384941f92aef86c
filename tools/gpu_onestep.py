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
    print("%s: %.1f s" % (name, time.time() - t0), flush=True)
