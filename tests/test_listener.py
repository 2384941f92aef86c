"""b2ContactListener::PreSolve / PostSolve and a user b2ContactFilter.

The harness installs the same recording listener / filter on every backend (box2d-mt_amd/harness/harness.cpp): the listener logs
every PreSolve (old and new manifold words, enabled flag after the call) and PostSolve (solver point count, impulse bits)
callback; in mode 8 its PreSolve disables the contacts a fixed rule of the body indices picks (b2Contact::SetEnabled(false),
b2Contact.h:117-123), the filter refuses the pairs another fixed rule picks (b2ContactFilter::ShouldCollide override,
b2WorldCallbacks.h:52-63). Both change the physics, so the body states are compared as well.

  CPU : the drop-in host layer over the C oracle  vs  the real reference build   (the oracle is pinned)
  GPU : the product (HIP)                          vs  the oracle                 (exact-order mode: bit for bit)

The reference calls the listener in island / contact-array order when it runs on one thread and in proxy-id order when its
callbacks are deferred (b2ContactManager.cpp:431-434, 466-469); per step the SET of callbacks and their payloads is what is
pinned, so each step's rows are sorted before they are compared. Reference sites: b2Contact.cpp:283-297 (PreSolve),
b2Island.cpp:532-570 (Report -> PostSolve), b2ContactManager.cpp:283-287 (AddPair -> ShouldCollide).
"""
import os

import numpy as np
import pytest

import b2harness as bh

MODE_ALL = 1 | 2 | 4          # begin / end, PreSolve, PostSolve: recorded
MODE_DISABLE = 1 | 2 | 4 | 8  # ... and PreSolve switches contacts off by the harness's rule
MODE_MATERIAL = 1 | 2 | 4 | 16  # ... and PreSolve edits the contacts' material: SetTangentSpeed on every ground contact (a conveyor
#                                 belt, Testbed/Tests/ConveyorBelt.h:70-83), SetFriction / SetRestitution / Reset* on others (b2Contact.h:129-160)
MODES = {"record": MODE_ALL, "filter": MODE_ALL, "disable": MODE_DISABLE, "material": MODE_MATERIAL}
CASES = [("rain", bh.RAIN, 120, 0, 4, 150), ("piles", bh.PILES, 25, 5, 6, 140), ("pyramid", bh.PYRAMID, 9, 1, 1, 120),
         ("circlestack", bh.CIRCLE_STACK, 6, 5, 1, 120)]


def rows(ev):
    """a step's callbacks as a sorted list of tuples (call order inside a step differs between the backends)"""
    return sorted(tuple(r) for r in ev.tolist())


def run_pair(a, b, steps, mode, use_filter, what):
    for w in (a, b):
        w.record_events(mode=mode)
        if use_filter:
            w.set_filter(True)
    seen = {0: 0, 1: 0, 2: 0, 3: 0}
    disabled = 0
    for s in range(steps):
        a.step(1)
        b.step(1)
        ea, eb = a.events_ex(), b.events_ex()
        assert rows(ea) == rows(eb), "%s: listener callbacks differ at step %d" % (what, s)
        for k in seen:
            seen[k] += int((eb[:, 0] == k).sum())
        disabled += int(((eb[:, 0] == 2) & (eb[:, 9] == 0)).sum())
        assert a.contact_count == b.contact_count, "%s: contact count at step %d" % (what, s)
        assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32)), "%s: body states differ at step %d" % (what, s)
    return seen, disabled


@pytest.mark.parametrize("name,scene,p0,p1,seed,steps", CASES)
@pytest.mark.parametrize("variant", ["record", "disable", "filter", "material"])
def test_oracle_listener_and_filter_match_the_reference(ref, oracle, name, scene, p0, p1, seed, steps, variant):
    a = ref.world(scene, p0, p1, seed=seed)
    b = oracle.world(scene, p0, p1, seed=seed)
    seen, disabled = run_pair(a, b, steps, MODES[variant], variant == "filter", name + "/" + variant)
    assert seen[2] > 0 and seen[3] > 0, "no PreSolve / PostSolve callback ever fired: test is vacuous"
    if variant == "disable" and name != "circlestack":
        assert disabled > 0, "the PreSolve rule never disabled a contact: test is vacuous"
    a.close()
    b.close()


CCD_CASES = [("bullets", bh.BULLETS, 20, 4, 1, 120), ("field", bh.FIELD, 300, 40, 5, 80), ("rain", bh.RAIN, 120, 0, 4, 150),
             ("piles", bh.PILES, 25, 5, 6, 140)]


@pytest.mark.parametrize("name,scene,p0,p1,seed,steps", CCD_CASES)
@pytest.mark.parametrize("variant", ["disable", "material"])
def test_oracle_presolve_acts_inside_the_toi_substep_that_called_it(ref, oracle, name, scene, p0, p1, seed, steps, variant):
    """Continuous physics on: b2Contact::Update calls PreSolve from INSIDE b2World::StepSolveTOI (b2World.cpp:866,946), and a
    contact the callback switches off there keeps the sweeps of its bodies and stays out of the sub-step's island
    (b2World.cpp:873-881, 948-954); a material it edits is what the sub-step's solver reads. The same listener on the real
    reference build and on the drop-in layer over the oracle: every callback of every step and the body states, bit for bit.
    (With the calls replayed after the step - rounds 2 and 3 until this test - every one of these eight runs diverged from
    the reference within 75 steps.)"""
    flags = bh.DEFAULT_FLAGS | bh.F_CONTINUOUS
    a = ref.world(scene, p0, p1, seed=seed, flags=flags)
    b = oracle.world(scene, p0, p1, seed=seed, flags=flags)
    seen, disabled = run_pair(a, b, steps, MODES[variant], False, name + "/ccd/" + variant)
    assert seen[2] > 0 and seen[3] > 0
    if variant == "disable":
        assert disabled > 0
    a.close()
    b.close()


def test_material_edits_change_the_motion(oracle):
    """the conveyor-belt rule really moves things (else the material variant above proves nothing)"""
    a = oracle.world(bh.PILES, 25, 5, seed=6)
    b = oracle.world(bh.PILES, 25, 5, seed=6)
    b.record_events(mode=16)
    a.step(100)
    b.step(100)
    assert np.abs(a.bodies()[:, 0] - b.bodies()[:, 0]).max() > 0.5  # carried sideways by the belt
    a.close()
    b.close()


def test_filter_changes_the_contact_set(oracle):
    """the user filter really refuses pairs (else the filter variant above proves nothing)"""
    a = oracle.world(bh.RAIN, 120, 0, seed=4)
    b = oracle.world(bh.RAIN, 120, 0, seed=4)
    b.set_filter(True)
    a.step(100)
    b.step(100)
    assert a.contact_count != b.contact_count or not np.array_equal(a.bodies(), b.bodies())
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,p0,p1,seed,steps", CASES + [("field", bh.FIELD, 1500, 0, 8, 60)])
@pytest.mark.parametrize("variant", ["record", "disable", "filter", "material"])
def test_device_listener_and_filter_match_the_oracle(amd, oracle, monkeypatch, name, scene, p0, p1, seed, steps, variant):
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")  # every island in the reference's order: states stay bit-equal
    a = amd.world(scene, p0, p1, seed=seed)
    b = oracle.world(scene, p0, p1, seed=seed)
    seen, disabled = run_pair(a, b, steps, MODES[variant], variant == "filter", name + "/" + variant)
    assert seen[2] > 0 and seen[3] > 0
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,p0,p1,seed,steps", CCD_CASES)
@pytest.mark.parametrize("variant", ["disable", "material"])
def test_device_presolve_acts_inside_the_toi_substep_that_called_it(amd, oracle, monkeypatch, name, scene, p0, p1, seed, steps, variant):
    """The device's event loop is one kernel: the step calls PreSolve for the logged Updates in order, and the first answer
    that changes its contact sends the phase back to its snapshot for another run with the answers so far (b2hip.hip:
    toiPreSolveRounds). Callbacks (each exactly once) and states equal the oracle's, which calls PreSolve inline as the
    reference does; the counter proves that phases were in fact run again."""
    import ctypes as C
    import b2hip
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    flags = bh.DEFAULT_FLAGS | bh.F_CONTINUOUS
    a = amd.world(scene, p0, p1, seed=seed, flags=flags)
    b = oracle.world(scene, p0, p1, seed=seed, flags=flags)
    seen, disabled = run_pair(a, b, steps, MODES[variant], False, name + "/ccd/" + variant)
    assert seen[2] > 0 and seen[3] > 0
    ctr = b2hip.Counters()
    b2hip.lib().b2hip_get_counters(C.c_void_p(a.device_world()), C.byref(ctr))
    assert ctr.toi_pre_solve_reruns > 0, "no PreSolve answer ever changed a TOI sub-step: the test is vacuous"
    a.close()
    b.close()


@pytest.mark.gpu
def test_device_post_solve_in_default_mode_reports_every_solved_constraint(amd):
    """Default (coloured) mode on a pile that takes the block solver: one PostSolve per touching contact of the solved
    islands, with the impulses the contact list shows afterwards."""
    w = amd.world(bh.PYRAMID, 30, 1)
    w.record_events(mode=4)
    for s in range(80):
        w.step(1)
        ev = w.events_ex()
        ids, flags, man = w.contacts()
        touching = int(((flags & 1) != 0).sum())
        assert (ev[:, 0] == 3).all()
        if s > 40:
            assert abs(len(ev) - touching) <= 0.02 * touching + 2, "step %d: %d PostSolve callbacks for %d touching contacts" % (s, len(ev), touching)
    w.close()


def test_immediate_callbacks_run_on_the_executors_threads(oracle, ref):
    """b2TaskExecutor use (b2TaskExecutor.h:27-79, b2WorldCallbacks.h:135-173): with a 4-thread executor the listener's
    *Immediate callbacks are called from the worker threads, each with its own threadId < 4, and the deferred callbacks
    afterwards on the stepping thread in the reference's order - the recorded callbacks and the physics (mode 8 disables
    contacts from PreSolve) equal the reference build's run with 4 threads and the drop-in layer's own run with 1."""
    import ctypes as C
    worlds = {"drop-in x4": oracle.world(bh.RAIN, 200, 0, seed=4, threads=4), "drop-in x1": oracle.world(bh.RAIN, 200, 0, seed=4, threads=1),
              "reference x4": ref.world(bh.RAIN, 200, 0, seed=4, threads=4)}
    for w in worlds.values():
        w.record_events(mode=MODE_DISABLE)
    for s in range(120):
        rows_ = {}
        for name, w in worlds.items():
            w.step(1)
            rows_[name] = rows(w.events_ex())
        assert rows_["drop-in x4"] == rows_["reference x4"] == rows_["drop-in x1"], "callbacks differ at step %d" % s
    a = worlds["drop-in x4"].bodies()
    for name in ("drop-in x1", "reference x4"):
        assert np.array_equal(a.view(np.uint32), worlds[name].bodies().view(np.uint32)), name
    hits = np.zeros(9, np.int32)
    w = worlds["drop-in x4"]
    w.L.b2h_immediate_calls_by_thread.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    w.L.b2h_immediate_calls_by_thread(w.ptr, hits.ctypes.data_as(C.POINTER(C.c_int)))
    assert hits[8] == 0 and hits[4:8].sum() == 0, "a callback saw a thread id outside the executor's 4 threads: %s" % hits
    assert np.count_nonzero(hits[:4]) >= 2, "every *Immediate callback ran on one thread: %s" % hits
    one = np.zeros(9, np.int32)
    w1 = worlds["drop-in x1"]
    w1.L.b2h_immediate_calls_by_thread(w1.ptr, one.ctypes.data_as(C.POINTER(C.c_int)))
    assert one[0] == hits[:4].sum() and one[1:].sum() == 0, "the same callbacks, all on thread 0: %s vs %s" % (one, hits)
    for w in worlds.values():
        w.close()


@pytest.mark.gpu
def test_device_callbacks_on_a_four_thread_executor_match_the_oracle(amd, oracle, monkeypatch):
    """The product with a 4-thread b2ThreadPoolTaskExecutor: the filter's ShouldCollide and the *Immediate callbacks are
    called in batches from the worker threads (b2hip_set_contact_filter_batch / b2hip_set_pre_solve_batch), the physics and
    the deferred callbacks equal the oracle's single-threaded run, and more than one thread took part."""
    import ctypes as C
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a = amd.world(bh.RAIN, 300, 0, seed=9, threads=4)
    b = oracle.world(bh.RAIN, 300, 0, seed=9, threads=1)
    run_pair(a, b, 120, MODE_DISABLE, True, "rain / 4 threads")
    hits = np.zeros(9, np.int32)
    a.L.b2h_immediate_calls_by_thread.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    a.L.b2h_immediate_calls_by_thread(a.ptr, hits.ctypes.data_as(C.POINTER(C.c_int)))
    assert hits[4:].sum() == 0 and np.count_nonzero(hits[:4]) >= 2, hits
    a.close()
    b.close()
