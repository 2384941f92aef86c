"""GPU tests of the end of a sweep over islands that run launch per colour (box2d-mt_amd/csrc/b2d_kernels_sweep_end.h):
k_large_rest (the small colours of a sweep as data flow per body, one launch), k_sweep_end (tail colours, the hub rows as ONE
fixed point, leftover hub rows lane after lane, the joint walk, the verdict of a position iteration - one single-workgroup
launch), k_large_warm (the warm start of a launch-per-colour solve body by body, one launch). Reference: b2Island::Solve, Box2D/Dynamics/b2Island.cpp:256-336; b2ContactSolver.cpp:293-603, 676-752.

Bars:
  * everything but the hub's fixed point does the arithmetic of the launches it replaces, in the same order on every body:
    with the hub rows swept lane after lane on both sides (B2HIP_HUB_SERIAL=1) the states are THE SAME BITS as round 4's launch
    sequence (B2HIP_NO_SWEEP_END=1: k_large_velocity / k_large_position per colour, k_large_hub, k_large_joints,
    k_large_pos_end), step by step, whatever the split into launched colours / rest colours / tail colours;
  * the hub rows as one fixed point over up to 1024 lanes settle to 2^-21 of max(|hub row|, sum of |changes|): against the
    lane-after-lane sweep IN THE SAME ORDER 30 steps of the Tumbler - ten steps after the boxes reach the container - agree
    to 1e-3 (measured 2e-6 .. 1e-4; a falling pile multiplies a difference by ~1.6 per step) - a tolerance, stated;
  * run-to-run deterministic.
"""
import os

import numpy as np
import pytest

import b2harness as bh

pytestmark = pytest.mark.gpu

KEYS = ("B2HIP_HUB_SERIAL", "B2HIP_NO_SWEEP_END", "B2HIP_NO_TAIL", "B2HIP_HUB_WIDE", "B2HIP_TAIL_ROWS", "B2HIP_SOLVER_LAUNCHES",
        "B2HIP_NO_REST", "B2HIP_REST_ROWS", "B2HIP_FORCE_LARGE", "B2HIP_HUB_WAVES", "B2HIP_NO_BODY_WARM", "B2HIP_REST_HUB", "B2HIP_HUB_ORDER", "B2HIP_NO_HUB_ORDER", "B2HIP_NO_HUB_BUILD")
CCD = bh.F_SLEEP | bh.F_WARM | bh.F_CONTINUOUS
LAUNCHES = {"B2HIP_SOLVER_LAUNCHES": "1"}  # no resident block solver, no k_blocks_sweep: a launch per colour


def run(amd, monkeypatch, scene, steps, env, **kw):
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = amd.world(scene, **kw)
    out = []
    for _ in range(steps):
        w.step(1)
        out.append((bh.fnv1a64(w.bodies()), w.contact_count))
    b = w.bodies()
    w.close()
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    return out, b


def first_diff(a, b):
    return next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), None)


CASES = [("tumbler60 (k_blocks_sweep + hub + joint)", bh.TUMBLER, 120, {}, dict(p0=60)),
         ("tumbler60, launch per colour", bh.TUMBLER, 120, LAUNCHES, dict(p0=60)),
         ("tumbler100, launch per colour", bh.TUMBLER, 120, LAUNCHES, dict(p0=100)),
         ("pyramid90, launch per colour, CCD", bh.PYRAMID, 150, LAUNCHES, dict(p0=90, p1=1, flags=CCD)),
         ("vehicles buried, launch per colour", bh.VEHICLES, 160, LAUNCHES, dict(p0=700, p1=5, seed=3)),
         ("machines buried, launch per colour", bh.MACHINES, 160, LAUNCHES, dict(p0=600, p1=6, seed=3)),
         ("machines buried (k_blocks_sweep)", bh.MACHINES, 160, {}, dict(p0=600, p1=6, seed=3))]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sweep_end_folding_is_bit_identical_to_the_launches_it_replaces(amd, monkeypatch, case):
    name, scene, steps, extra, kw = case
    serial = dict(extra, B2HIP_HUB_SERIAL="1")
    base, _ = run(amd, monkeypatch, scene, steps, dict(serial, B2HIP_NO_SWEEP_END="1"), **kw)
    variants = {"default split": {}, "no rest colours, tail colours up to 100 rows": {"B2HIP_NO_REST": "1", "B2HIP_TAIL_ROWS": "100"},
                "rest colours up to 28 000 rows": {"B2HIP_REST_ROWS": "100000"}, "every colour a tail colour": {"B2HIP_NO_REST": "1", "B2HIP_TAIL_ROWS": "100000"},
                "no rest, no tail": {"B2HIP_NO_REST": "1", "B2HIP_NO_TAIL": "1"},
                "warm start as a sweep of launches": {"B2HIP_NO_BODY_WARM": "1"},
                # (round 6: by default the rest rows and the end of the sweep are ONE launch, k_rest_hub)
                "rest rows and end of the sweep as two launches": {"B2HIP_REST_HUB": "0"},
                "... fused in the velocity sweeps only": {"B2HIP_REST_HUB": "1"},
                "fused, rest colours up to 100 000 rows": {"B2HIP_REST_ROWS": "100000", "B2HIP_REST_HUB": "2"},
                # (round 6: by default ONE launch sorts the hub group's segment in place, k_hub_build)
                "hub list by k_hub_flag + scan + k_hub_fill": {"B2HIP_NO_HUB_BUILD": "1"}}
    for label, env in variants.items():
        other, _ = run(amd, monkeypatch, scene, steps, dict(serial, **env), **kw)
        first = first_diff(base, other)
        assert first is None, "%s, %s: differs from the launch sequence of round 4 at step %d" % (name, label, first)


def test_hub_rows_as_one_fixed_point_agree_with_the_lane_after_lane_sweep(amd, monkeypatch):
    for n in (60, 100):
        # (the same ORDER of the hub rows on both sides - by the partner's highest colour, k_hub_order: what is compared is the
        # fixed point against the lane-after-lane sweep, not one order against another. The boxes reach the container at
        # step ~20; from there a falling pile multiplies any difference by ~1.6 per step - tools/gpu_r06_lockstep.py: 3e-6 at
        # step 20, the fixed point's 2^-21, 1.6e-5 at step 26, 3e-2 at step 40 - so the comparison is made 10 steps in)
        _, serial = run(amd, monkeypatch, bh.TUMBLER, 30, {"B2HIP_HUB_SERIAL": "1", "B2HIP_HUB_ORDER": "1"}, p0=n)
        _, wide = run(amd, monkeypatch, bh.TUMBLER, 30, {}, p0=n)
        assert np.isfinite(wide).all()
        d = np.abs(serial[:, :2] - wide[:, :2]).max()
        assert d < 1e-3, "Tumbler %d: the fixed point over the workgroup is %g away from the lane-after-lane sweep after 30 steps" % (n, d)


def test_hub_list_in_one_launch_is_the_list_of_the_four_launches(amd, monkeypatch):
    """k_hub_build against k_hub_flag + scan + k_hub_fill + k_hub_order, with the fixed point over the workgroup (the default):
    the same rows in the same order, so the same bits."""
    for n, extra in ((60, {}), (100, LAUNCHES)):
        a, _ = run(amd, monkeypatch, bh.TUMBLER, 100, dict(extra), p0=n)
        b, _ = run(amd, monkeypatch, bh.TUMBLER, 100, dict(extra, B2HIP_NO_HUB_BUILD="1"), p0=n)
        first = first_diff(a, b)
        assert first is None, "Tumbler %d: the one-launch hub list differs from the four launches' at step %d" % (n, first)


def test_default_mode_with_hubs_is_run_to_run_deterministic(amd, monkeypatch):
    a, _ = run(amd, monkeypatch, bh.TUMBLER, 200, {}, p0=100)
    b, _ = run(amd, monkeypatch, bh.TUMBLER, 200, {}, p0=100)
    assert a == b
    # launch per colour with rest colours and the end of the sweep in one launch, twice
    a, _ = run(amd, monkeypatch, bh.TUMBLER, 120, LAUNCHES, p0=100)
    b, _ = run(amd, monkeypatch, bh.TUMBLER, 120, LAUNCHES, p0=100)
    assert a == b


def two_hub_world(monkeypatch, env):
    """Two heavy bars side by side on the ground (touching), 40 boxes in a row on each and a second layer on top: one island of
    ~ 330 constraints with TWO hubs (41 + solid contacts each). The hub with the higher degree is the primary one (its rows
    are the fixed point of k_sweep_end); the other hub's 40-odd rows are leftover rows - more than k_sweep_end sweeps lane
    after lane itself, so from the second step on they go to k_large_hub behind the fixed point."""
    import b2hip
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = b2hip.World(allow_sleep=False)  # (the pile must still be an island at the last step, where the counters are read)
    g = w.create_body(b2hip.STATIC, (0.0, 0.0))
    w.create_fixture(g, b2hip.edge_shape((-100.0, 0.0), (100.0, 0.0)))
    for side, extra in ((-1.0, 0), (1.0, 1)):
        bar = w.create_body(b2hip.DYNAMIC, (side * 10.005, 0.5))
        w.create_fixture(bar, b2hip.box_shape(10.0, 0.5), density=20.0, friction=0.6)
        n = 40 + extra
        for i in range(n):
            x = side * 10.005 - 9.6 + i * (19.2 / (n - 1))
            b = w.create_body(b2hip.DYNAMIC, (x, 1.0 + 0.2))
            w.create_fixture(b, b2hip.box_shape(0.2, 0.2), density=1.0, friction=0.4)
        for i in range(n - 1):
            x = side * 10.005 - 9.6 + (i + 0.5) * (19.2 / (n - 1))
            b = w.create_body(b2hip.DYNAMIC, (x, 1.0 + 0.6))
            w.create_fixture(b, b2hip.box_shape(0.2, 0.2), density=1.0, friction=0.4)
    return w


def test_leftover_hub_rows_go_to_k_large_hub_when_there_are_many(monkeypatch):
    def go(env, steps=60):
        w = two_hub_world(monkeypatch, env)
        out = []
        for _ in range(steps):
            w.step()
            out.append(bh.fnv1a64(w.bodies8()))
        s, c = w.bodies8(), w.counters()
        w.close()
        return out, s, c
    a, sa, ca = go(LAUNCHES)
    b, sb, cb = go(LAUNCHES)
    assert a == b, "two hubs: not run-to-run deterministic"
    assert ca["hub_constraints"] > 80 and ca["large_islands"] == 1, ca
    assert np.isfinite(sa).all()
    # the lane-after-lane sweep of all hub rows (one valid order) and the fixed point + k_large_hub (another): the pile rests
    # on the bars either way - nothing sinks in, nothing flies off
    c, sc, _ = go(dict(LAUNCHES, B2HIP_HUB_SERIAL="1"))
    for s in (sa, sc):
        boxes = s[(s[:, 7] == 2)]
        assert boxes[:, 1].min() > 0.3 and boxes[:, 1].max() < 2.2, (boxes[:, 1].min(), boxes[:, 1].max())
    assert np.abs(sa[:, :2] - sc[:, :2]).max() < 5e-2
