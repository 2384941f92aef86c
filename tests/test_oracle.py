"""CPU tests: the plain-C oracle (oracle/b2o_*.c) against the golden vectors generated from the real
reference, and - when oracle/_ref is present - against the reference itself. Bit-exact throughout."""
import ctypes as C
import os

import numpy as np
import pytest

import b2harness as bh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

SCENES = ["helloworld", "pyramid12", "pyramid5x3", "pyramid30", "piles", "rain", "circlestack", "field", "tumbler6", "tumbler20", "sensors", "ropes", "machines", "vehicles"]


def run_scene(h, golden, name, check_every_step=True):
    sc, p0, p1, seed, steps = [int(x) for x in golden[name + "/params"]]
    f0, f1 = [float(x) for x in golden[name + "/fparams"]]
    w = h.world(sc, p0, p1, f0, f1, seed)
    counts = np.zeros(steps, np.int32)
    hashes = []
    for s in range(steps):
        w.step(1)
        counts[s] = w.contact_count
        if check_every_step:
            hashes.append(bh.fnv1a64(w.bodies()[:, :3]))
    return w, counts, hashes


@pytest.mark.parametrize("name", SCENES)
def test_oracle_matches_golden_scene(oracle, golden, name):
    w, counts, hashes = run_scene(oracle, golden, name)
    assert np.array_equal(counts, golden[name + "/contact_counts"]), "contact count trace differs"
    assert hashes == list(golden[name + "/hashes"]), "per-step pose hash differs"
    b = w.bodies()
    g = golden[name + "/bodies"]
    assert np.array_equal(b.view(np.uint32), g.view(np.uint32)), "final body state not bit-identical"
    assert np.array_equal(w.mass().view(np.uint32), golden[name + "/mass"].view(np.uint32))
    ids, flags, man = w.contacts()
    assert np.array_equal(ids, golden[name + "/contact_ids"])
    assert np.array_equal(flags, golden[name + "/contact_flags"])
    assert np.array_equal(man.view(np.uint32), golden[name + "/contact_manifolds"].view(np.uint32))
    w.close()


def test_oracle_helloworld_lines(oracle):
    w = oracle.world(bh.HELLO)
    lines = []
    for _ in range(60):
        w.step(1)
        b = w.bodies()[1]
        lines.append("%4.2f %4.2f %4.2f" % (b[0], b[1], b[2]))
    want = open(os.path.join(GOLD, "helloworld.txt")).read().split("\n")[:60]
    assert lines == want


def test_oracle_pyramid141_prefix(oracle, golden):
    """Full-size config 2 on the oracle (brute-force broad-phase): first steps only, contact counts + hash."""
    name = "pyramid141"
    sc, p0, p1, seed, steps = [int(x) for x in golden[name + "/params"]]
    w = oracle.world(sc, p0, p1, 0.0, 0.0, seed)
    for s in range(3):
        w.step(1)
        assert w.contact_count == golden[name + "/contact_counts"][s]
        assert bh.fnv1a64(w.bodies()[:, :3]) == golden[name + "/hashes"][s]
    w.close()


def test_oracle_collide_vectors(built_libs):
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    v = np.load(os.path.join(GOLD, "collide_vectors.npz"))
    fp = C.POINTER(C.c_float)
    bad = 0
    for sa, sb, xa, xb, want in zip(v["shapeA"], v["shapeB"], v["xfA"], v["xfB"], v["manifold"]):
        out = np.zeros(16, np.float32)
        sa = np.ascontiguousarray(sa)
        sb = np.ascontiguousarray(sb)
        xa = np.ascontiguousarray(xa)
        xb = np.ascontiguousarray(xb)
        L.b2o_collide(sa.ctypes.data_as(C.c_void_p), xa.ctypes.data_as(fp), sb.ctypes.data_as(C.c_void_p),
                      xb.ctypes.data_as(fp), out.ctypes.data_as(fp))
        bad += not np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert bad == 0


def test_host_polygon_set_and_mass(oracle):
    """b2PolygonShape::Set / ComputeMass of the drop-in host API (hull, welding, normals, centroid, mass)."""
    v = np.load(os.path.join(GOLD, "polygon_vectors.npz"))
    for i, (inp, want) in enumerate(zip(v["inp"], v["out"])):
        n = int(inp[0])
        # (make_golden.py gave vector i the density 1 + 0.01 i) count, vertices, normals, centroid, mass, mass centre, inertia: bitwise
        got = oracle.polygon(inp[1:1 + 2 * n].reshape(n, 2), density=1.0 + 0.01 * i)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "polygon vector %d" % i


def test_oracle_determinism(oracle):
    """The reference's own consistency rule (TestMT.cpp:91-110): two runs, bitwise equal every step."""
    a = oracle.world(bh.RAIN, 120, seed=3)
    b = oracle.world(bh.RAIN, 120, seed=3)
    for _ in range(80):
        a.step(1)
        b.step(1)
        assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32))


@pytest.mark.parametrize("scene,p0,p1,steps", [(bh.PYRAMID, 20, 1, 120), (bh.PILES, 25, 6, 150), (bh.RAIN, 300, 0, 150),
                                               (bh.FIELD, 800, 0, 80), (bh.CIRCLE_STACK, 6, 5, 150), (bh.TUMBLER, 12, 0, 200)])
def test_oracle_matches_reference_build(oracle, ref, scene, p0, p1, steps):
    """Direct A/B against skitzoid/Box2D-MT compiled from /root/reference (skipped where it is absent)."""
    a = oracle.world(scene, p0, p1, seed=11)
    r = ref.world(scene, p0, p1, seed=11)
    for s in range(steps):
        a.step(1)
        r.step(1)
        assert a.contact_count == r.contact_count, "step %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), r.bodies().view(np.uint32)), "step %d" % s
    ia, fa, ma = a.contacts()
    ir, fr, mr = r.contacts()
    assert np.array_equal(ia, ir) and np.array_equal(fa, fr)
    assert np.array_equal(ma.view(np.uint32), mr.view(np.uint32))
