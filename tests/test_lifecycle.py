"""Body / fixture life cycle and the mutators between steps (the drop-in boundary beyond build-once scenes).

One scripted scene (box2d-mt_amd/harness/scenes.h: BuildLifecycle + LifecycleEdits, written against the public Box2D API only)
destroys bodies and fixtures in the middle of a heap (b2World::DestroyBody b2World.cpp:585-670, b2Body::DestroyFixture
b2Body.cpp:238-308), creates new ones afterwards (the freed broad-phase proxy ids are reused in the dynamic tree's LIFO
order, b2DynamicTree.cpp:53-99, and the island seed order follows m_nonStaticBodies' swap-remove, b2World.cpp:662-667),
teleports (SetTransform b2Body.cpp:451-473), puts to sleep and wakes (SetAwake b2Body.h:690-718), turns bodies into bullets
and back (SetBullet + RecalculateToiCandidacy b2ContactManager.cpp:566-640), applies linear / angular impulses
(b2Body.h:885-950), switches sensors, thick shapes and filter data (b2Fixture.cpp:180-257), retunes a wheel joint's spring,
switches bodies off and on again (SetActive b2Body.cpp:496-544, a fixture created meanwhile) and changes body types
(SetType b2Body.cpp:118-188: static, kinematic and back), destroys a jointed body (its joint goes with it) and drags a mouse joint more slowly than the sleep tolerance for longer
than b2_timeToSleep (b2Body::SetAwake(true) restarts the sleep timer whether the body sleeps or not: ADVICE r1).

  CPU : host layer over the C oracle  vs  the real reference build: body states, contact sets, manifolds and the listener's
        callbacks (begin / end / PreSolve / PostSolve), every step, bit for bit
  GPU : the product                    vs  the oracle, the same way (exact-order mode and default mode)
"""
import numpy as np
import pytest

import b2harness as bh

CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM
CASES = [(48, 5, bh.DEFAULT_FLAGS), (48, 5, CCD), (60, 11, CCD), (36, 2, bh.DEFAULT_FLAGS)]
STEPS = 240


def rows(ev):
    return sorted(tuple(r) for r in ev.tolist())


def compare(a, b, steps, what, events=True):
    if events:
        a.record_events(mode=7)
        b.record_events(mode=7)
    destroyed = 0
    for s in range(steps):
        a.step(1)
        b.step(1)
        A, B = a.bodies(), b.bodies()
        assert a.body_count == b.body_count, "%s: body count at step %d" % (what, s)
        assert a.contact_count == b.contact_count, "%s: contact count at step %d (%d vs %d)" % (what, s, a.contact_count, b.contact_count)
        assert np.array_equal(A.view(np.uint32), B.view(np.uint32)), "%s: body states differ at step %d (bodies %s)" % (
            what, s, np.nonzero((A.view(np.uint32) != B.view(np.uint32)).any(axis=1))[0][:8])
        if events:
            assert rows(a.events_ex()) == rows(b.events_ex()), "%s: listener callbacks differ at step %d" % (what, s)
        if s % 20 == 19 or s in (25, 26, 32, 33, 40, 45):
            ia, fa, ma = a.contacts()
            ib, fb, mb = b.contacts()
            assert np.array_equal(ia, ib) and np.array_equal(fa, fb), "%s: contact set at step %d" % (what, s)
            assert np.array_equal(ma.view(np.uint32), mb.view(np.uint32)), "%s: manifolds at step %d" % (what, s)
        destroyed = int((B[:, 7] < 0).sum())
    assert destroyed >= 7, "the script destroyed only %d bodies: test is vacuous" % destroyed
    return a.bodies()


@pytest.mark.parametrize("count,seed,flags", CASES)
def test_oracle_life_cycle_matches_the_reference(ref, oracle, count, seed, flags):
    a = ref.world(bh.LIFECYCLE, count, 0, seed=seed, flags=flags)
    b = oracle.world(bh.LIFECYCLE, count, 0, seed=seed, flags=flags)
    # (with continuous physics on the reference also calls PreSolve / PostSolve from its TOI sub-steps, which the bridge does
    # not report - include/b2hip.h; the callbacks are compared where the two are defined alike, the states always)
    compare(a, b, STEPS, "reference vs oracle", events=not (flags & bh.F_CONTINUOUS))
    a.close()
    b.close()


def test_slowly_dragged_body_stays_awake(oracle):
    """the mouse-joint target creeps at 0.006 m/s (below b2_linearSleepTolerance) for 66 steps = 1.1 s
    (b2_timeToSleep is 0.5 s): every SetTarget restarts the crate's sleep timer, so it must never fall asleep."""
    w = oracle.world(bh.LIFECYCLE, 36, 0, seed=2)
    crate = None
    for s in range(200):
        w.step(1)
        B = w.bodies()
        if crate is None:
            crate = B.shape[0] - 1  # the last body built by the scene (bodies created later are appended after it)
        if 112 <= s < 176:
            assert B[crate, 6] == 1.0, "the dragged crate fell asleep at step %d" % s
    w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("count,seed,flags", CASES)
@pytest.mark.parametrize("mode", ["exact", "default", "exact, rows sent early"])
def test_device_life_cycle_matches_the_oracle(amd, oracle, monkeypatch, count, seed, flags, mode):
    monkeypatch.delenv("B2HIP_EARLY_ROWS_MIN", raising=False)
    if mode.startswith("exact"):
        monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
        # (what worlds of 65 536 bodies and more do: the rows leave behind SynchronizeFixtures by DMA on a second stream, the end
        # of the step sends what changed since - here under bodies, fixtures and joints created and destroyed between steps,
        # SetTransform / SetAwake / SetActive / SetType edits and TOI events, which all have to reach the host's rows)
        if mode != "exact":
            monkeypatch.setenv("B2HIP_EARLY_ROWS_MIN", "1")
    else:
        monkeypatch.setenv("B2HIP_SMALL_MAX_W", "512")
        # the heap is one island of a few hundred contacts: the exact-order in-LDS solver at its 512-row limit takes it, and the
        # jointed islands (cart + wheels, the dragged crate) with it
    a = amd.world(bh.LIFECYCLE, count, 0, seed=seed, flags=flags)
    b = oracle.world(bh.LIFECYCLE, count, 0, seed=seed, flags=flags)
    compare(a, b, STEPS, "device (%s) vs oracle" % mode)  # (device and oracle share the bridge's definition: callbacks compared with CCD on as well)
    a.close()
    b.close()


# ---- the property setters (b2Body.h:620-688, b2Body.cpp:310-424, 546-565; b2Fixture.h:306-334) ---------------------------------
# box2d-mt_amd/harness/scenes.h: PropsEdits - on the rain scene, between steps: SetLinearDamping / SetAngularDamping,
# SetGravityScale, SetFixedRotation on and off, b2Fixture::SetDensity + ResetMassData, SetMassData, SetSleepingAllowed off and
# on, b2Fixture::SetFriction / SetRestitution (bodies and ground: contacts made afterwards), GetLinearVelocityFromLocalPoint
# feeding an impulse.
PROPS_CASES = [(120, 4, bh.DEFAULT_FLAGS), (60, 9, bh.DEFAULT_FLAGS | bh.F_CONTINUOUS)]


def compare_props(a, b, steps, what):
    for s in range(steps):
        a.step(1)
        b.step(1)
        A, B = a.bodies(), b.bodies()
        assert a.contact_count == b.contact_count, "%s: contact count at step %d" % (what, s)
        assert np.array_equal(A.view(np.uint32), B.view(np.uint32)), "%s: body states differ at step %d (bodies %s)" % (
            what, s, np.nonzero((A.view(np.uint32) != B.view(np.uint32)).any(axis=1))[0][:8])
        if s % 20 == 19:
            ia, fa, ma = a.contacts()
            ib, fb, mb = b.contacts()
            assert np.array_equal(ia, ib) and np.array_equal(fa, fb), "%s: contact set at step %d" % (what, s)
            assert np.array_equal(ma.view(np.uint32), mb.view(np.uint32)), "%s: manifolds at step %d" % (what, s)
    return a.bodies()


@pytest.mark.parametrize("count,seed,flags", PROPS_CASES)
def test_oracle_property_setters_match_the_reference(ref, oracle, count, seed, flags):
    a = ref.world(bh.PROPS, count, 0, seed=seed, flags=flags)
    b = oracle.world(bh.PROPS, count, 0, seed=seed, flags=flags)
    edited = compare_props(a, b, 200, "reference vs oracle")
    plain = ref.world(bh.RAIN, count, 0, seed=seed, flags=flags)
    plain.step(200)
    assert not np.array_equal(edited, plain.bodies()), "the edits changed nothing: test is vacuous"
    for w in (a, b, plain):
        w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("count,seed,flags", PROPS_CASES)
def test_device_property_setters_match_the_oracle(amd, oracle, monkeypatch, count, seed, flags):
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")  # (the heap on the ground may exceed the in-LDS solver: reference order everywhere)
    a = amd.world(bh.PROPS, count, 0, seed=seed, flags=flags)
    b = oracle.world(bh.PROPS, count, 0, seed=seed, flags=flags)
    compare_props(a, b, 200, "device vs oracle")
    a.close()
    b.close()
