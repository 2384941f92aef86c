"""GPU tests of the recoveries of the large-island solver (round 6). b2World::Step has no failure path
(Box2D/Dynamics/b2World.cpp:1613-1710); the device's fast paths have places where a step could fail - a wait between the
workgroups of a resident / data-flow solver kernel that times out (a co-tenant on the device, a workgroup that was not
resident), a constraint that finds no colour free on its two bodies, incremental colouring rounds that do not converge. Each
used to end the step with an error and leave a world that refuses to step. Now (box2d-mt_amd/csrc/b2hip_host_phases.h: runLarge):

  * a timed-out wait: the state the solver had changed is put back (k_solver_snapshot) and the solve runs once more on the
    plain path - rows by colour, a launch per colour, hub rows and tail colours in ONE workgroup: nothing waits for another
    workgroup. Same colouring, same arithmetic, same order on every body: THE SAME BITS as a run that took the plain path
    from the start, step by step. Forced here with B2HIP_TEST_SPIN_MAX=1 (every wait that is not satisfied at its first look
    gives up), on every solver that waits: k_solve_blocks (a pyramid), k_blocks_sweep (the Tumbler: hub + joint),
    k_rest_hub / k_large_rest (launch per colour with rest rows).
  * no colour free (B2HIP_TEST_MAX_COLORS=N leaves N colours per range): the constraint is swept in order with the hub rows.
  * rounds that do not converge (B2HIP_TEST_COLOR_ROUNDS=1): the grid-wide rounds finish the colouring.
  Both: a valid, finite, run-to-run deterministic step; the pile stays a pile.
  * B2HIP_NO_RECOVER=1: the old behaviour - the step fails with an error.
"""
import ctypes as C
import os

import numpy as np
import pytest

import b2harness as bh
import b2hip

pytestmark = pytest.mark.gpu

KEYS = ("B2HIP_TEST_SPIN_MAX", "B2HIP_TEST_MAX_COLORS", "B2HIP_TEST_COLOR_ROUNDS", "B2HIP_NO_RECOVER", "B2HIP_SOLVER_LAUNCHES", "B2HIP_NO_REST",
        "B2HIP_HUB_SERIAL", "B2HIP_NO_SWEEP_BLOCKS", "B2HIP_REST_HUB", "B2HIP_NO_BLOCKS")
CCD = bh.F_SLEEP | bh.F_WARM | bh.F_CONTINUOUS


def run(amd, monkeypatch, scene, steps, env, **kw):
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = amd.world(scene, **kw)
    L = b2hip.lib()
    out = []
    for _ in range(steps):
        w.step(1)
        out.append((bh.fnv1a64(w.bodies()), w.contact_count))
    ctr = b2hip.Counters()
    L.b2hip_get_counters(C.c_void_p(w.device_world()), C.byref(ctr))
    b = w.bodies()
    w.close()
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    return out, b, ctr


def first_diff(a, b):
    return next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), None)


# name, scene, steps, the fast path's environment, the plain path's environment, world arguments
# (where there is a hub its fixed point is the same single-workgroup kernel in the same order on both paths)
CASES = [
    ("pyramid 90 (k_solve_blocks)", bh.PYRAMID, 150, {}, {"B2HIP_SOLVER_LAUNCHES": "1", "B2HIP_NO_REST": "1"}, dict(p0=90, p1=1, flags=CCD)),
    ("tumbler 60 (k_blocks_sweep, hub, joint)", bh.TUMBLER, 120, {}, {"B2HIP_SOLVER_LAUNCHES": "1", "B2HIP_NO_REST": "1"}, dict(p0=60)),
    ("tumbler 100, launch per colour with rest rows (k_rest_hub)", bh.TUMBLER, 120, {"B2HIP_SOLVER_LAUNCHES": "1"}, {"B2HIP_SOLVER_LAUNCHES": "1", "B2HIP_NO_REST": "1"}, dict(p0=100)),
    ("tumbler 100 (k_blocks_sweep over several hundred blocks, hub, joint)", bh.TUMBLER, 90, {}, {"B2HIP_SOLVER_LAUNCHES": "1", "B2HIP_NO_REST": "1"}, dict(p0=100)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_a_timed_out_wait_is_recovered_to_the_bits_of_the_plain_path(amd, monkeypatch, case):
    name, scene, steps, fast, plain, kw = case
    base, _, c0 = run(amd, monkeypatch, scene, steps, plain, **kw)
    assert c0.solver_recoveries == 0
    forced, state, c1 = run(amd, monkeypatch, scene, steps, dict(fast, B2HIP_TEST_SPIN_MAX="1"), **kw)
    assert np.isfinite(state).all()
    assert c1.solver_recoveries > 0, "%s: no wait timed out - the test is vacuous" % name
    first = first_diff(base, forced)
    assert first is None, "%s: the recovered run differs from the plain path at step %d (%d recoveries)" % (name, first, c1.solver_recoveries)
    again, _, c2 = run(amd, monkeypatch, scene, steps, dict(fast, B2HIP_TEST_SPIN_MAX="1"), **kw)
    assert again == forced and c2.solver_recoveries == c1.solver_recoveries, "%s: not run-to-run deterministic" % name


def test_without_recovery_a_timed_out_wait_fails_the_step(monkeypatch):
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("B2HIP_TEST_SPIN_MAX", "1")
    monkeypatch.setenv("B2HIP_NO_RECOVER", "1")
    w = b2hip.World()
    g = w.create_body(b2hip.STATIC, (0.0, 0.0))
    w.create_fixture(g, b2hip.edge_shape((-100.0, 0.0), (100.0, 0.0)))
    for i in range(40):
        for j in range(40 - i):
            b = w.create_body(b2hip.DYNAMIC, (-20.0 + 0.5 * i + 1.0 * j, 0.5 + 1.0 * i))
            w.create_fixture(b, b2hip.box_shape(0.5, 0.5), density=1.0, friction=0.4)
    failed = False
    try:
        for _ in range(60):
            w.step()
    except Exception as e:  # noqa: BLE001
        failed = "timed out" in str(e) or "failed state" in str(e)
    w.close()
    assert failed, "B2HIP_NO_RECOVER=1: a wait that gives up must fail the step as before"


@pytest.mark.parametrize("env,what", [({"B2HIP_TEST_MAX_COLORS": "2"}, "no free colour"), ({"B2HIP_TEST_COLOR_ROUNDS": "1"}, "rounds that do not converge")])
@pytest.mark.parametrize("launches", [False, True])
def test_colouring_that_runs_out_goes_on(amd, monkeypatch, env, what, launches):
    extra = {"B2HIP_SOLVER_LAUNCHES": "1"} if launches else {}
    a, sa, ca = run(amd, monkeypatch, bh.PYRAMID, 120, dict(extra, **env), p0=40, p1=1, flags=CCD)
    b, sb, cb = run(amd, monkeypatch, bh.PYRAMID, 120, dict(extra, **env), p0=40, p1=1, flags=CCD)
    assert ca.solver_recoveries > 0 and cb.solver_recoveries > 0, "%s: never happened - the test is vacuous" % what
    assert a == b, "%s: not run-to-run deterministic" % what
    assert np.isfinite(sa).all()
    # the pile stays a pile: against the unforced run the boxes are where a pyramid's boxes are
    ref, sr, cr = run(amd, monkeypatch, bh.PYRAMID, 120, dict(extra), p0=40, p1=1, flags=CCD)
    assert cr.solver_recoveries == 0
    boxes = sa[sa[:, 7] == 2]
    assert boxes[:, 1].min() > 0.3 and boxes[:, 1].max() < 1.05 * sr[:, 1].max() + 0.5, (boxes[:, 1].min(), boxes[:, 1].max(), sr[:, 1].max())
    assert abs(ca.contacts - cr.contacts) <= 0.02 * cr.contacts


def test_a_host_that_polls_late_still_gets_the_census(amd, monkeypatch):
    """A world without a partition publishes twice per step: the island census (k_block_census) and, a few microseconds to
    ~80 us later, the state behind k_color_small. Into ONE buffer with ONE count (as first built in round 6) a host that
    reached its poll late found the second number where it waited for the first - "island census was not published", one
    step in a few hundred on a busy box. Two buffers, two counts now; here the host is MADE late (the delay is read once per
    process, so this test only proves something in a process that has not polled before - it still must pass in any)."""
    monkeypatch.setenv("B2HIP_TEST_POLL_DELAY_US", "300")
    late, s1, _ = run(amd, monkeypatch, bh.TUMBLER, 80, {"B2HIP_NO_BLOCKS": "1"}, p0=60)
    monkeypatch.delenv("B2HIP_TEST_POLL_DELAY_US", raising=False)
    again, s2, _ = run(amd, monkeypatch, bh.TUMBLER, 80, {"B2HIP_NO_BLOCKS": "1"}, p0=60)
    assert np.isfinite(s1).all() and late == again
