"""The reference's OWN Testbed scene headers, unmodified, on the drop-in API ("unmodified Testbed").

oracle/Makefile compiles tests/testbed/scenes_main.cpp - a headless stand-in for Testbed/Framework/Test.h plus
`#include "Testbed/Tests/<Scene>.h"` straight from /root/reference - three times into oracle/_ref/ (git-ignored, travels
with the snapshot like the compiled reference): against the reference's Box2D (libtestbed_ref.so), against
box2d-mt_amd/host over the C oracle (libtestbed_oracle.so) and against box2d-mt_amd/host over libb2hip.so
(libtestbed_amd.so, the product). Scenes: EVERY entry of the reference's Testbed (Testbed/Tests/TestEntries.cpp, 64 entries
from 61 scene headers; Rope.h needs Box2D/Rope, which is outside the hot path's scope and is the one header that does not
compile: tools/testbed_compile_check.sh says 61 of 62).

  CPU : drop-in host layer over the oracle  ==  reference build, per-step summaries (body / contact counts, position sum,
        top speed, awake count) over the whole run
  GPU : the three TestPassed predicates pass on the product; the others run finite and, in exact-order mode, reproduce
        the oracle-backed run step for step
The listener calls the reference makes from inside its TOI sub-steps (b2World.cpp:866,946, b2Island.cpp:398-530) are
delivered too: begin / end / PostSolve after the step in call order (b2hip_get_toi_callbacks), PreSolve by the step itself
where the sub-step calls it, because its answer changes that sub-step (include/b2hip.h, b2hip_toi_callback). TunnelingTest
(edits the world from them), Breakable (reads the landing impulse from a sub-step's PostSolve) and ConveyorBelt (sets the
belt's tangent speed from the PreSolve of the landing sub-step, whose own solver already uses it) follow the reference step
for step. ManyBodies 1-5 (10 000 - 50 000 bodies) are compared on the GPU
against the reference-order run only through their smaller sibling ManyBodies6 (the C oracle's broad-phase is brute force).
"""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NONE, PASS, FAIL = 0, 1, 2


def load(kind):
    path = os.path.join(ROOT, "oracle", "_ref", "libtestbed_%s.so" % kind)
    if not os.path.exists(path):
        pytest.skip("%s not built (needs /root/reference at build time: make -C oracle testbed)" % os.path.basename(path))
    L = C.CDLL(path, mode=C.RTLD_LOCAL)
    L.testbed_run.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    L.testbed_trace.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    return L


def trace(L, name, steps):
    out = np.zeros((steps, 6))
    res = L.testbed_trace(name.encode(), steps, out.ctypes.data)
    return res, out


def trace_hash(L, name, steps):
    """FNV-1a of every body's full state (position, angle, velocities, awake, type: bit patterns) after every step"""
    out = np.zeros(steps, np.uint64)
    L.testbed_trace_hash.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    res = L.testbed_trace_hash(name.encode(), steps, out.ctypes.data)
    return res, out


def all_entries():
    import re
    return re.findall(r'\{ "(\w+)", ', open(os.path.join(ROOT, "tests", "testbed", "scenes_main.cpp")).read())


BIG_SCENES = ("ManyBodies1", "ManyBodies2", "ManyBodies3", "ManyBodies4", "ManyBodies5")
LONG = {"SleepCollideTest": 700, "Tumbler": 300, "QueryTest": 1, "SleepCollidePerf": 120, "TunnelingTest": 900}
CPU_SCENES = [(n, LONG.get(n, 200)) for n in all_entries() if n not in BIG_SCENES]


@pytest.mark.parametrize("name,steps", CPU_SCENES)
def test_reference_scenes_on_the_drop_in_layer_match_the_reference(name, steps):
    ra, ta = trace(load("ref"), name, steps)
    rb, tb = trace(load("oracle"), name, steps)
    assert ra == rb, "TestPassed differs"
    bad = np.nonzero((ta != tb).any(axis=1))[0]
    assert bad.size == 0, "%s: summaries differ first at step %d: %s vs %s" % (name, bad[0], ta[bad[0]], tb[bad[0]])
    assert ta[-1, 4] == 1.0 and ta[-1, 0] > 0
    # ... and the full body state, bit for bit, after every step (the summaries are sums)
    _, ha = trace_hash(load("ref"), name, steps)
    _, hb = trace_hash(load("oracle"), name, steps)
    bad = np.nonzero(ha != hb)[0]
    assert bad.size == 0, "%s: body-state bits differ first at step %d" % (name, bad[0])


def test_tunneling_test_predicate_on_the_drop_in_layer():
    ra, ta = trace(load("ref"), "TunnelingTest", 900)
    rb, tb = trace(load("oracle"), "TunnelingTest", 900)
    assert ra == PASS and rb == PASS
    assert tb[-1, 4] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps", [("SleepCollideTest", 1800), ("TunnelingTest", 1800), ("QueryTest", 1)])
def test_reference_test_passed_predicates_on_the_gpu(name, steps):
    """TestMT.cpp:36,113-114: the three scenes that define TestPassed() must report PASS on the product (default mode)."""
    out = np.zeros(6)
    res = load("amd").testbed_run(name.encode(), steps, out.ctypes.data)
    assert res == PASS, "%s: TestPassed = %d" % (name, res)
    assert out[4] == 1.0


def test_every_testbed_entry_is_covered():
    names = all_entries()
    assert len(names) == 64 and len(CPU_SCENES) == 64 - len(BIG_SCENES)


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps", [(n, min(k, 200)) for n, k in CPU_SCENES])
def test_reference_scenes_on_the_gpu_match_the_oracle_backed_run(name, steps, monkeypatch):
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")  # every island in the reference's constraint order
    ra, ta = trace(load("amd"), name, steps)
    rb, tb = trace(load("oracle"), name, steps)
    assert ra == rb
    bad = np.nonzero((ta != tb).any(axis=1))[0]
    assert bad.size == 0, "%s: summaries differ first at step %d: %s vs %s" % (name, bad[0], ta[bad[0]], tb[bad[0]])
    _, ha = trace_hash(load("amd"), name, steps)
    _, hb = trace_hash(load("oracle"), name, steps)
    bad = np.nonzero(ha != hb)[0]
    assert bad.size == 0, "%s: body-state bits differ first at step %d" % (name, bad[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps", [("ManyBodies6", 240), ("MultithreadDemo", 300), ("Car", 240), ("SleepCollidePerf", 200)])
def test_reference_scenes_run_in_default_mode(name, steps):
    res, t = trace(load("amd"), name, steps)
    assert res in (NONE, PASS)
    assert (t[:, 4] == 1.0).all(), "non-finite body state"
    assert t[-1, 0] > 0


# ---- ManyBodies1 .. 5 (10 000 - 50 000 bodies) directly against traces of the reference build -----------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", BIG_SCENES)
def test_many_bodies_1_to_5_on_the_gpu_match_the_reference_trace(name, monkeypatch):
    """The reference's own large Testbed scenes, unmodified, on the product in its DEFAULT mode against traces generated
    from the reference build in the container (tests/golden/testbed_big.npz, tests/golden/make_golden_testbed.py: the C
    oracle's brute-force broad-phase cannot follow at these sizes): the six summaries and the hash of every body's full
    state, step for step. (These fields are sparse: their islands stay in the reference-order tier.)"""
    monkeypatch.delenv("B2HIP_FORCE_LARGE", raising=False)
    g = np.load(os.path.join(ROOT, "tests", "golden", "testbed_big.npz"))
    want_s, want_h = g[name + "/summaries"], g[name + "/hashes"]
    steps = len(want_h)
    _, got_s = trace(load("amd"), name, steps)
    _, got_h = trace_hash(load("amd"), name, steps)
    bad = np.nonzero((got_s != want_s).any(axis=1))[0]
    assert bad.size == 0, "%s: summaries differ from the reference build first at step %d: %s vs %s" % (name, bad[0], got_s[bad[0]], want_s[bad[0]])
    bad = np.nonzero(got_h != want_h)[0]
    assert bad.size == 0, "%s: body-state bits differ from the reference build first at step %d" % (name, bad[0])
