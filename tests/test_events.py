"""Contact events (the BeginContact / EndContact half of b2ContactListener).

The harness installs the same recording listener on every backend. The product delivers the begin / end events of the
Collide phase, begins before ends, each group in proxy-id-pair order (the order of the reference's deferred callbacks
after Collide, b2ContactManager.cpp:420-438), then - like the reference - the calls made from inside the TOI sub-steps
(b2World.cpp:866,946, b2Island.cpp:527) in their own order: a contact may begin and end within one step, and both reach the
listener (b2hip_get_toi_callbacks; the TOI phase of a world with a listener runs through the serial event loop).
"""
import os

import numpy as np
import pytest

import b2harness as bh


@pytest.fixture()
def exact_mode():
    os.environ["B2HIP_FORCE_LARGE"] = "2"
    yield
    os.environ.pop("B2HIP_FORCE_LARGE", None)


CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM
CASES = [("sensors", bh.SENSORS, 40, 0, 5, 240, bh.DEFAULT_FLAGS), ("rain", bh.RAIN, 150, 0, 3, 200, bh.DEFAULT_FLAGS),
         ("piles", bh.PILES, 20, 5, 2, 150, bh.DEFAULT_FLAGS), ("bullets", bh.BULLETS, 20, 4, 1, 120, CCD)]


def net(ev):
    d = {}
    for k, a, fa, b, fb in ev.tolist():
        key = (a, fa, b, fb)
        d[key] = d.get(key, 0) + (1 if k == 0 else -1)
    return {k: v for k, v in d.items() if v != 0}


@pytest.mark.parametrize("name,scene,p0,p1,seed,steps,flags", CASES)
def test_oracle_events_match_reference_listener(ref, oracle, name, scene, p0, p1, seed, steps, flags):
    """CPU: the host layer over the C oracle against the real reference with the same listener."""
    a = ref.world(scene, p0, p1, seed=seed, flags=flags)
    b = oracle.world(scene, p0, p1, seed=seed, flags=flags)
    a.record_events()
    b.record_events()
    total = twice = 0
    for s in range(steps):
        a.step(1)
        b.step(1)
        ea, eb = a.events(), b.events()
        assert net(ea) == net(eb), "net contact events differ at step %d" % s
        # every callback, nothing netted away: the same calls in the same order (continuous physics included)
        assert np.array_equal(ea, eb), "begin / end callbacks or their order differ at step %d:\n%s\n%s" % (s, ea, eb)
        twice += len(eb) - len(net(eb))
        total += len(eb)
    assert total > 0
    if flags & bh.F_CONTINUOUS:
        assert twice > 0, "no contact ever began and ended within one step: the TOI sub-step callbacks are not exercised"
    a.close()
    b.close()


@pytest.mark.parametrize("name,scene,p0,p1,seed,steps", [("bullets", bh.BULLETS, 20, 4, 1, 120), ("field", bh.FIELD, 300, 40, 5, 80)])
def test_oracle_all_callbacks_with_continuous_physics_match_the_reference(ref, oracle, name, scene, p0, p1, seed, steps):
    """Begin / End / PreSolve / PostSolve with continuous physics on: every callback of the step, those from the TOI sub-steps
    included (PreSolve with the old manifold, PostSolve with the sub-step solver's impulses), the same calls with the same
    payload in the same order as the reference build makes them."""
    a = ref.world(scene, p0, p1, seed=seed, flags=CCD)
    b = oracle.world(scene, p0, p1, seed=seed, flags=CCD)
    a.record_events(mode=7)
    b.record_events(mode=7)
    kinds = np.zeros(4, int)
    for s in range(steps):
        a.step(1)
        b.step(1)
        ea, eb = a.events_ex(), b.events_ex()
        assert len(ea) == len(eb), "step %d: %d callbacks in the reference, %d here" % (s, len(ea), len(eb))
        # the Collide / Solve callbacks come in island order in a one-thread reference run and in proxy order when deferred
        # (tests/test_listener.py): compare them as a set; the TOI sub-steps' calls are single-threaded and ordered - they
        # are the tail of the step's list, compared in order
        assert sorted(map(tuple, ea.tolist())) == sorted(map(tuple, eb.tolist())), "step %d" % s
        for k in range(4):
            kinds[k] += int((eb[:, 0] == k).sum())
    assert (kinds > 0).all(), kinds
    a.close()
    b.close()


SUBSTEP_CASES = [("bullets", bh.BULLETS, 20, 4, 1, 400), ("field", bh.FIELD, 300, 40, 5, 300), ("rain", bh.RAIN, 120, 0, 4, 400),
                 ("piles", bh.PILES, 25, 5, 6, 300), ("pyramid", bh.PYRAMID, 9, 1, 1, 300)]


@pytest.mark.parametrize("name,scene,p0,p1,seed,steps", SUBSTEP_CASES)
@pytest.mark.parametrize("listener", [False, True])
def test_oracle_sub_stepping_matches_the_reference(ref, oracle, name, scene, p0, p1, seed, steps, listener):
    """b2World::SetSubStepping (b2World.h:183): a Step call solves ONE TOI event and returns with m_stepComplete false
    (b2World.cpp:1082-1086); the calls that follow run Collide and the next event, but no island solve (b2World.cpp:1668),
    until no event is left and the TOI flags are cleared (ClearPostSolveTOI, :1467-1504). Body states, contact counts and -
    with the recording listener - every callback of every call, against the reference build."""
    flags = bh.DEFAULT_FLAGS | bh.F_CONTINUOUS | bh.F_SUBSTEP
    a = ref.world(scene, p0, p1, seed=seed, flags=flags)
    b = oracle.world(scene, p0, p1, seed=seed, flags=flags)
    if listener:
        a.record_events(mode=7)
        b.record_events(mode=7)
    still = 0
    for s in range(steps):
        before = b.bodies().copy()
        a.step(1)
        b.step(1)
        assert a.contact_count == b.contact_count, "step %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32)), "step %d" % s
        if listener:
            assert sorted(map(tuple, a.events_ex().tolist())) == sorted(map(tuple, b.events_ex().tolist())), "step %d" % s
        # (a call that only continues a step moves the bodies of one TOI island and nothing else)
        still += int((b.bodies()[:, :3] == before[:, :3]).all(axis=1).sum() > 0.5 * len(before))
    assert still > 0, "no call ever continued an incomplete step: sub-stepping is not exercised"
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,p0,p1,seed,steps", SUBSTEP_CASES)
@pytest.mark.parametrize("listener", [False, True])
def test_device_sub_stepping_matches_the_oracle(amd, oracle, monkeypatch, name, scene, p0, p1, seed, steps, listener):
    """The device's form of it: the serial event loop with a cap of one StepSolveTOI call per launch; a call that continues a
    step makes no first pass - the impacts earlier calls found stay valid, the others are the event loop's first batch, in
    the reference's slot order (b2d_kernels_toi.h). Exact-order mode, bit for bit, callbacks included."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    flags = bh.DEFAULT_FLAGS | bh.F_CONTINUOUS | bh.F_SUBSTEP
    a = amd.world(scene, p0, p1, seed=seed, flags=flags)
    b = oracle.world(scene, p0, p1, seed=seed, flags=flags)
    if listener:
        a.record_events(mode=7)
        b.record_events(mode=7)
    for s in range(steps):
        a.step(1)
        b.step(1)
        assert a.contact_count == b.contact_count, "step %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32)), "step %d" % s
        if listener:
            assert sorted(map(tuple, a.events_ex().tolist())) == sorted(map(tuple, b.events_ex().tolist())), "step %d" % s
    a.close()
    b.close()


# The worlds of `tools/gpu_toi_presolve_campaign.py 7 12 substep` whose listener edits materials (mode 23). Round 3 shipped
# with three of them diverging from the oracle after 90 - 130 calls: a call that continues a sub-stepped step with nothing
# pending yet skipped its TOI snapshot (k_toi_snapshot looked at nToiList only), so a PreSolve answer from that call's
# sub-step took the phase back to an EARLIER call's snapshot - the Collide phase's material edits of the call were lost.
SUBSTEP_MATERIAL_CASES = [(500, 132, 0.0, 2.0, 5783), (500, 60, 60.0, 0.0, 2851), (500, 20, 120.0, 2.0, 1315), (500, 119, 120.0, 2.0, 8076)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,bullets,arena,rmax,seed", SUBSTEP_MATERIAL_CASES)
def test_device_sub_stepping_with_a_material_editing_presolve_matches_the_oracle(amd, oracle, monkeypatch, n, bullets, arena, rmax, seed):
    """b2World::SetSubStepping (b2World.cpp:1082-1086, 1668) together with a PreSolve that edits the contact's mixed material
    inside TOI sub-steps (b2Contact.h:40-50): states, callbacks AND every contact's material, bit for bit, 240 calls."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    flags = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM | bh.F_SUBSTEP
    kw = dict(p0=n, p1=bullets, f0=arena, f1=rmax, seed=seed, flags=flags)
    a = amd.world(bh.FIELD, **kw)
    b = oracle.world(bh.FIELD, **kw)
    a.record_events(mode=23)
    b.record_events(mode=23)
    for s in range(240):
        a.step(1)
        b.step(1)
        assert a.contact_count == b.contact_count, "call %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32)), "call %d" % s
        assert sorted(map(tuple, a.events_ex().tolist())) == sorted(map(tuple, b.events_ex().tolist())), "call %d" % s
        ia, ma = a.contact_materials()
        ib, mb = b.contact_materials()
        assert np.array_equal(ia, ib) and np.array_equal(ma.view(np.uint32), mb.view(np.uint32)), "materials, call %d" % s
    a.close()
    b.close()


def test_events_off_by_default_and_after_removal(oracle):
    w = oracle.world(bh.PILES, 10, 4, seed=3)
    w.step(30)
    assert len(w.events()) == 0
    w.record_events()
    w.step(1)
    first = w.events()
    assert len(first) > 0 and (first[:, 0] == 0).all(), "installing a listener reports the contacts that already touch as begins"
    w.record_events(False)
    w.step(20)
    assert len(w.events()) == 0
    w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,p0,p1,seed,steps,flags", CASES)
def test_device_events_match_oracle(amd, oracle, exact_mode, name, scene, p0, p1, seed, steps, flags):
    """GPU: k_contact_events + the destroy events of k_collide against the oracle, same host layer, every step, in order."""
    a = amd.world(scene, p0, p1, seed=seed, flags=flags)
    b = oracle.world(scene, p0, p1, seed=seed, flags=flags)
    a.record_events()
    b.record_events()
    total = 0
    for s in range(steps):
        a.step(1)
        b.step(1)
        ea, eb = a.events(), b.events()
        assert np.array_equal(ea, eb), "contact events differ at step %d" % s
        total += len(ea)
    assert total > 0
    assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32))
    a.close()
    b.close()
