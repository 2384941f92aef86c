"""GPU parity tests of the continuous-collision path (b2World::SolveTOI on the device: k_toi_first,
k_toi_loop), driven through the drop-in Box2D API + C ABI with continuous physics ON.

Bars: contact counts, contact sets, feature ids and awake flags bit-exact; floats bit-exact wherever the
discrete solver walks the reference's constraint order (small islands in default mode, every island in
exact-order mode) - the TOI sub-steps themselves always follow the reference's order.
"""
import ctypes as C
import os

import numpy as np
import pytest

import b2harness as bh

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM
SMALL_ISLAND_SCENES = ["ccd_helloworld", "ccd_bullets", "ccd_field", "ccd_rain"]
ALL_SCENES = SMALL_ISLAND_SCENES + ["ccd_pyramid12", "ccd_tumbler6"]


@pytest.fixture(scope="module")
def toi_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "toi_scenes.npz"))


@pytest.fixture()
def exact_mode():
    os.environ["B2HIP_FORCE_LARGE"] = "2"
    yield
    os.environ.pop("B2HIP_FORCE_LARGE", None)


@pytest.fixture()
def default_mode():
    os.environ.pop("B2HIP_FORCE_LARGE", None)
    yield


def check_golden(h, g, name):
    sc, p0, p1, seed, steps = [int(x) for x in g[name + "/params"]]
    f0, f1 = [float(x) for x in g[name + "/fparams"]]
    w = h.world(sc, p0, p1, f0, f1, seed, flags=CCD)
    for s in range(steps):
        w.step(1)
        assert w.contact_count == g[name + "/contact_counts"][s], "contact count differs at step %d" % s
        assert bh.fnv1a64(w.bodies()[:, :3]) == g[name + "/hashes"][s], "pose hash differs at step %d" % s
    assert np.array_equal(w.bodies().view(np.uint32), g[name + "/bodies"].view(np.uint32))
    ids, flags, man = w.contacts()
    assert np.array_equal(ids, g[name + "/contact_ids"])
    assert np.array_equal(flags, g[name + "/contact_flags"])
    assert np.array_equal(man.view(np.uint32), g[name + "/contact_manifolds"].view(np.uint32))
    w.close()


@pytest.mark.parametrize("name", SMALL_ISLAND_SCENES)
def test_ccd_small_island_scenes_bit_exact_vs_golden(amd, toi_golden, default_mode, monkeypatch, name):
    # (the exact-order tier ends at 128 rows by default - above that the block solver wins, DESIGN.md - and the rain / field
    # heaps of these goldens grow to a few hundred rows: the tier is widened to its 512-row limit here)
    monkeypatch.setenv("B2HIP_SMALL_MAX_W", "512")
    check_golden(amd, toi_golden, name)


@pytest.mark.parametrize("name", ALL_SCENES)
def test_ccd_exact_order_mode_bit_exact_vs_golden(amd, toi_golden, exact_mode, name):
    check_golden(amd, toi_golden, name)


@pytest.mark.parametrize("scene,p0,p1,f0,f1,steps", [(bh.BULLETS, 150, 8, 0.0, 0.0, 200), (bh.FIELD, 800, 200, 50.0, 3.0, 120),
                                                      (bh.PILES, 30, 6, 0.0, 0.0, 100)])
def test_ccd_side_by_side_with_c_oracle(amd, oracle, exact_mode, scene, p0, p1, f0, f1, steps):
    """Fresh seeds: HIP path vs the C oracle next to it with continuous physics on, every step, bitwise."""
    a = amd.world(scene, p0, p1, f0, f1, seed=29, flags=CCD)
    o = oracle.world(scene, p0, p1, f0, f1, seed=29, flags=CCD)
    for s in range(steps):
        a.step(1)
        o.step(1)
        assert a.contact_count == o.contact_count, "step %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), o.bodies().view(np.uint32)), "step %d" % s
    ia, fa, ma = a.contacts()
    io, fo, mo = o.contacts()
    assert np.array_equal(ia, io) and np.array_equal(fa, fo)
    assert np.array_equal(ma.view(np.uint32), mo.view(np.uint32))


def test_ccd_chains_create_contacts_between_moving_bodies_in_event_order(amd, default_mode, monkeypatch):
    """A box that lands on the ground through a TOI event next to others: its re-inserted proxy finds a NEW pair with a
    neighbour. The reference creates that contact inside the sub-step; the parallel chains note it (two moving, non-bullet
    bodies: the contact takes no part in the rest of the phase) and their close-out creates what was noted in the reference's
    order - event by event, pairs by proxy ids - instead of sending the whole phase to the serial loop. The serial loop IS the
    reference's order (pinned against the oracle by the tests above), so: the same world with the hand-over switched off
    (B2HIP_TOI_NO_CHAIN_CREATE=1: such steps are replayed serially) must give the same bits - states every step, contact
    ARRAY order and manifolds every tenth (the order is what later steps depend on)."""
    import b2hip

    def run(no_chain_create):
        if no_chain_create:
            monkeypatch.setenv("B2HIP_TOI_NO_CHAIN_CREATE", "1")
        else:
            monkeypatch.delenv("B2HIP_TOI_NO_CHAIN_CREATE", raising=False)
        w = amd.world(bh.RAIN, 1500, 0, seed=7, flags=CCD)
        out = []
        for s in range(300):
            w.step(1)
            h = [bh.fnv1a64(w.bodies()), w.contact_count]
            if s % 10 == 9:
                ids, flags, man = w.contacts()
                h += [bh.fnv1a64(ids), bh.fnv1a64(flags), bh.fnv1a64(man)]
            out.append(tuple(h))
        ctr = b2hip.Counters()
        b2hip.lib().b2hip_get_counters(C.c_void_p(w.device_world()), C.byref(ctr))
        w.close()
        return out, ctr.toi_chain_contacts, ctr.toi_serial_fallbacks

    a, created, fallbacks = run(False)
    b, created_b, fallbacks_b = run(True)
    assert created > 0, "the chains never left a contact to their close-out: the test is vacuous"
    assert created_b == 0 and fallbacks_b > fallbacks
    first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
    assert first is None, "close-out creation and serial replay diverge at step %s" % first


def test_ccd_keeps_projectiles_inside_on_device(amd, default_mode):
    """The point of the TOI phase: no projectile tunnels through the 0.1-wide walls; with it off many do."""
    def escaped(flags):
        w = amd.world(bh.BULLETS, 40, 6, seed=2, flags=flags)
        w.step(60)
        b = w.bodies()
        w.close()
        return int(((np.abs(b[:, 0]) > 20.5) | (b[:, 1] < -0.5) | (b[:, 1] > 30.5)).sum())
    assert escaped(CCD) == 0
    assert escaped(bh.F_SLEEP | bh.F_WARM) > 5


def test_ccd_events_are_counted(amd, default_mode):
    """b2hip_get_counters reports the TOI activity of the last step (events happen in the first steps of the scene)."""
    w = amd.world(bh.BULLETS, 80, 6, seed=5, flags=CCD)
    dev = C.c_void_p(w.device_world())
    hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
    import b2hip
    events = calls = 0
    for _ in range(30):
        w.step(1)
        ctr = b2hip.Counters()
        assert hip.b2hip_get_counters(dev, C.byref(ctr)) == 0
        events += ctr.toi_events
        calls += ctr.toi_calls
        assert ctr.toi_calls >= ctr.toi_pending_first_pass
    assert events > 10 and calls > events
    w.close()


def test_ccd_components_run_side_by_side_and_stay_bit_exact(amd, oracle, default_mode, monkeypatch):
    """Bullets among free bodies: the event loop runs per connected component of the contact graph (k_toi_domains),
    components tied by a new contact are replayed serially (k_toi_loop_partial). Bitwise equal to the oracle every step,
    with and without that path, and the path really is the one that ran (few whole-phase fallbacks)."""
    hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
    import b2hip
    kw = dict(p0=2500, p1=300, f0=0.0, f1=0.0, seed=11, flags=CCD)
    o = oracle.world(bh.FIELD, **kw)
    want = []
    for _ in range(25):
        o.step(1)
        want.append((bh.fnv1a64(o.bodies()), o.contact_count))
    o.close()
    for no_domains in (False, True):
        if no_domains:
            monkeypatch.setenv("B2HIP_TOI_NO_DOMAINS", "1")
        w = amd.world(bh.FIELD, **kw)
        dev = C.c_void_p(w.device_world())
        events = 0
        for s in range(25):
            w.step(1)
            assert (bh.fnv1a64(w.bodies()), w.contact_count) == want[s], "step %d differs (no_domains=%s)" % (s, no_domains)
            ctr = b2hip.Counters()
            assert hip.b2hip_get_counters(dev, C.byref(ctr)) == 0
            events += ctr.toi_events
        assert events > 50
        if not no_domains:
            assert ctr.toi_serial_fallbacks <= 5, "the component path fell back to the serial loop in most steps"
        w.close()


@pytest.mark.parametrize("env", [{"B2HIP_TOI_DOM_WIDE": "1"}, {"B2HIP_NO_SIDE_STREAM": "1"}, {"B2HIP_TOI_NO_SPEC_DOMAINS": "1"},
                                 {"B2HIP_EARLY_ROWS_MIN": "1", "B2HIP_ROW_MARKS_CHECK": "1"}, {"B2HIP_EARLY_ROWS_MIN": "1"},
                                 {"B2HIP_EARLY_ROWS_MIN": "1", "B2HIP_NO_ROW_MARKS": "1"}],
                         ids=["wide component loops", "no side stream", "census first", "early rows + marks checked", "early rows + marks", "early rows, every row compared"])
def test_ccd_component_path_variants_are_bit_exact(amd, oracle, default_mode, monkeypatch, env):
    """Round 5's forms of the component path against the oracle, every step, on the field of the test above: the event
    loops on one wave per component (default) and on 512 lanes (B2HIP_TOI_DOM_WIDE); snapshot / adjacency / components on the
    side stream beside k_toi_first (default from the second step with events on) and on the main stream; the whole path queued
    behind k_toi_first without a look at its census (default while the path has been in use) and after that look; the read-back behind
    an early launch of the rows - forced on this small world with B2HIP_EARLY_ROWS_MIN - looking at marked tiles only
    (DW::b_rowDirty), with the marks CHECKED (a row that differs from the early launch's without a mark fails the step:
    b2hip_host_phases.h, downloadState), and comparing every row."""
    kw = dict(p0=2500, p1=300, f0=0.0, f1=0.0, seed=11, flags=CCD)
    o = oracle.world(bh.FIELD, **kw)
    want = []
    for _ in range(20):
        o.step(1)
        want.append((bh.fnv1a64(o.bodies()), o.contact_count))
    o.close()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = amd.world(bh.FIELD, **kw)
    for s in range(20):
        w.step(1)
        assert (bh.fnv1a64(w.bodies()), w.contact_count) == want[s], "step %d differs (%s)" % (s, env)
    w.close()


def test_ccd_speculative_component_path_survives_a_wrong_guess(amd, oracle, default_mode, monkeypatch):
    """The component path queued without its census (phaseToi) assumes that this step's pair update left the hash grid fresh,
    as the last step's had, and b2hip_step_end checks it. B2HIP_DEBUG_ASSUME_FRESH_GRID=1 makes every such step a wrong guess
    (the device is told the grid is stale): each goes back to the snapshot and through the serial loop - and the states stay
    the oracle's, bit for bit, with the fall-backs counted."""
    hip = C.CDLL(os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so"))
    import b2hip
    kw = dict(p0=2500, p1=300, f0=0.0, f1=0.0, seed=11, flags=CCD)
    o = oracle.world(bh.FIELD, **kw)
    want = []
    for _ in range(14):
        o.step(1)
        want.append((bh.fnv1a64(o.bodies()), o.contact_count))
    o.close()
    monkeypatch.setenv("B2HIP_DEBUG_ASSUME_FRESH_GRID", "1")
    w = amd.world(bh.FIELD, **kw)
    dev = C.c_void_p(w.device_world())
    for s in range(14):
        w.step(1)
        assert (bh.fnv1a64(w.bodies()), w.contact_count) == want[s], "step %d differs" % s
    ctr = b2hip.Counters()
    assert hip.b2hip_get_counters(dev, C.byref(ctr)) == 0
    assert ctr.toi_serial_fallbacks >= 8, "the wrong-guess path was not taken (%d fall-backs)" % ctr.toi_serial_fallbacks
    w.close()


def test_ccd_at_scale_matches_reference_trace(amd, default_mode):
    """30 000 free bodies with 3 000 bullets, 30 steps: per-step contact counts and full-state hashes recorded from the
    reference build (tests/golden/toi_scale.npz). Hundreds of TOI events per step go through the per-component path."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "toi_scale.npz"))
    name = "ccd_field30k"
    sc, p0, p1, seed, steps = [int(x) for x in g[name + "/params"]]
    f0, f1 = [float(x) for x in g[name + "/fparams"]]
    w = amd.world(sc, p0, p1, f0, f1, seed, flags=CCD)
    for s in range(steps):
        w.step(1)
        assert w.contact_count == int(g[name + "/contact_counts"][s]), "contact count differs at step %d" % s
        assert bh.fnv1a64(w.bodies()) == str(g[name + "/hashes"][s]), "state hash differs at step %d" % s
    w.close()
