"""Contact events (the BeginContact / EndContact half of b2ContactListener).

The harness installs the same recording listener on every backend. The product delivers one NET event per contact and
step at the end of Step(), begins before ends, each group in proxy-id-pair order (the order of the reference's deferred
callbacks after Collide, b2ContactManager.cpp:420-438). The reference additionally delivers events from inside its TOI
sub-steps, interleaved, and may deliver a begin and an end for one contact in one step; netted per step the two agree.
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
    total = 0
    for s in range(steps):
        a.step(1)
        b.step(1)
        ea, eb = a.events(), b.events()
        assert net(ea) == net(eb), "net contact events differ at step %d" % s
        assert len(eb) == len(net(eb)), "more than one event for one contact in a step"
        if len(ea) == len(eb) and not (flags & bh.F_CONTINUOUS):
            assert np.array_equal(ea, eb), "delivery order differs at step %d" % s
        total += len(eb)
    assert total > 0
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
