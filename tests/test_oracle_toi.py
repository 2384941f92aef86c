"""CPU tests for the continuous-collision (TOI) part of the oracle: GJK distance, time of impact and the
TOI event loop of b2World::Step, against golden vectors generated from the real reference
(tests/golden/make_golden_toi.py) and - when oracle/_ref is present - the reference itself. Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import b2harness as bh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM
fp = C.POINTER(C.c_float)

SCENES = ["ccd_helloworld", "ccd_bullets", "ccd_field", "ccd_pyramid12", "ccd_rain", "ccd_tumbler6"]


@pytest.fixture(scope="module")
def toi_golden():
    return np.load(os.path.join(GOLD, "toi_scenes.npz"))


@pytest.fixture(scope="module")
def vectors():
    return np.load(os.path.join(GOLD, "toi_vectors.npz"))


@pytest.fixture(scope="module")
def liboracle(built_libs):
    return C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


@pytest.mark.parametrize("name", SCENES)
def test_oracle_ccd_scene_matches_golden(oracle, toi_golden, name):
    g = toi_golden
    sc, p0, p1, seed, steps = [int(x) for x in g[name + "/params"]]
    f0, f1 = [float(x) for x in g[name + "/fparams"]]
    w = oracle.world(sc, p0, p1, f0, f1, seed, flags=CCD)
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


def test_oracle_distance_vectors(liboracle, vectors):
    v = vectors
    bad = 0
    for i in range(len(v["d_out"])):
        out = np.zeros(6, np.float32)
        a = np.ascontiguousarray(v["d_vertsA"][i]); b = np.ascontiguousarray(v["d_vertsB"][i])
        xa = np.ascontiguousarray(v["d_xfA"][i]); xb = np.ascontiguousarray(v["d_xfB"][i])
        liboracle.b2o_probe_distance(int(v["d_countA"][i]), a.ctypes.data_as(fp), C.c_float(v["d_radiusA"][i]), xa.ctypes.data_as(fp),
                                     int(v["d_countB"][i]), b.ctypes.data_as(fp), C.c_float(v["d_radiusB"][i]), xb.ctypes.data_as(fp),
                                     int(v["d_useRadii"][i]), out.ctypes.data_as(fp))
        bad += not np.array_equal(out.view(np.uint32), v["d_out"][i].view(np.uint32))
    assert bad == 0


def test_oracle_toi_vectors(liboracle, vectors):
    v = vectors
    bad = 0
    for i in range(len(v["t_out"])):
        out = np.zeros(2, np.float32)
        a = np.ascontiguousarray(v["t_vertsA"][i]); b = np.ascontiguousarray(v["t_vertsB"][i])
        sa = np.ascontiguousarray(v["t_sweepA"][i]); sb = np.ascontiguousarray(v["t_sweepB"][i])
        liboracle.b2o_probe_toi(int(v["t_countA"][i]), a.ctypes.data_as(fp), C.c_float(v["t_radiusA"][i]), sa.ctypes.data_as(fp),
                                int(v["t_countB"][i]), b.ctypes.data_as(fp), C.c_float(v["t_radiusB"][i]), sb.ctypes.data_as(fp),
                                C.c_float(1.0), out.ctypes.data_as(fp))
        bad += not np.array_equal(out.view(np.uint32), v["t_out"][i].view(np.uint32))
    assert bad == 0
    # the vectors cover every outcome the TOI loop acts on
    states = v["t_out"][:, 0].astype(int)
    assert (states == 3).sum() > 500 and (states == 4).sum() > 300


def test_ccd_keeps_projectiles_inside(oracle):
    """What continuous collision is for: with it no projectile tunnels out of the thin-walled room, without it many do."""
    def escaped(flags):
        w = oracle.world(bh.BULLETS, 40, 6, seed=2, flags=flags)
        w.step(60)
        b = w.bodies()
        w.close()
        return int(((np.abs(b[:, 0]) > 20.5) | (b[:, 1] < -0.5) | (b[:, 1] > 30.5)).sum())
    assert escaped(CCD) == 0
    assert escaped(bh.F_SLEEP | bh.F_WARM) > 5


@pytest.mark.parametrize("scene,p0,p1,f0,f1,steps", [(bh.BULLETS, 120, 8, 0.0, 0.0, 200), (bh.FIELD, 600, 150, 45.0, 3.0, 150),
                                                      (bh.PILES, 25, 6, 0.0, 0.0, 120), (bh.TUMBLER, 10, 0, 0.0, 0.0, 150)])
def test_oracle_ccd_matches_reference_build(oracle, ref, scene, p0, p1, f0, f1, steps):
    """Direct A/B with continuous physics on against skitzoid/Box2D-MT compiled from /root/reference."""
    a = oracle.world(scene, p0, p1, f0, f1, seed=13, flags=CCD)
    r = ref.world(scene, p0, p1, f0, f1, seed=13, flags=CCD)
    for s in range(steps):
        a.step(1)
        r.step(1)
        assert a.contact_count == r.contact_count, "step %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), r.bodies().view(np.uint32)), "step %d" % s
    ia, fa, ma = a.contacts()
    ir, fr, mr = r.contacts()
    assert np.array_equal(ia, ir) and np.array_equal(fa, fr)
    assert np.array_equal(ma.view(np.uint32), mr.view(np.uint32))


def test_reference_ccd_thread_count_invariance(ref):
    """The reference's own rule (TestMT.cpp): results do not depend on the thread count, TOI included."""
    a = ref.world(bh.BULLETS, 80, 6, seed=4, flags=CCD, threads=1)
    b = ref.world(bh.BULLETS, 80, 6, seed=4, flags=CCD, threads=4)
    for _ in range(100):
        a.step(1)
        b.step(1)
    assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32))
