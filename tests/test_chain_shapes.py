"""b2ChainShape on the hot path (Box2D/Collision/Shapes/b2ChainShape.h:32, b2ChainAndCircleContact.cpp,
b2ChainAndPolygonContact.cpp): every child segment of a chain is one device fixture of type "chain child" - the edge
b2ChainShape::GetChildEdge hands out for the narrow phase and the TOI proxy, an AABB without radius for the broad-phase.

CPU: the C oracle (through the drop-in host layer) against the golden traces generated from the reference build
(tests/golden/make_golden_r3.py) and, where oracle/_ref is present, against the reference side by side incl. ray casts.
GPU: the HIP path against the same goldens - bit for bit in the default mode (the scene's islands stay in the
reference-order tier) and in exact-order mode.
"""
import os

import numpy as np
import pytest

import b2harness as bh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["chains", "ccd_chains"]


@pytest.fixture(scope="module")
def golden_r3():
    return np.load(os.path.join(ROOT, "tests", "golden", "scenes_r3.npz"))


def run_and_compare(h, g, name):
    sc, p0, p1, seed, steps = [int(x) for x in g[name + "/params"]]
    f0, f1 = [float(x) for x in g[name + "/fparams"]]
    w = h.world(sc, p0, p1, f0, f1, seed, flags=int(g[name + "/flags"][0]))
    want_counts, want_hashes = g[name + "/contact_counts"], g[name + "/hashes"]
    for s in range(steps):
        w.step(1)
        assert w.contact_count == want_counts[s], "%s: contact count differs at step %d" % (name, s)
        assert bh.fnv1a64(w.bodies()[:, :3]) == want_hashes[s], "%s: pose hash differs at step %d" % (name, s)
    assert np.array_equal(w.bodies().view(np.uint32), g[name + "/bodies"].view(np.uint32))
    assert np.array_equal(w.mass().view(np.uint32), g[name + "/mass"].view(np.uint32))
    ids, flags, man = w.contacts()
    assert np.array_equal(ids, g[name + "/contact_ids"])  # (fixture index | child index << 16)
    assert (ids[:, 1] >> 16).max() > 0 or (ids[:, 3] >> 16).max() > 0, "no contact on a chain child beyond the first: vacuous"
    assert np.array_equal(flags, g[name + "/contact_flags"])
    assert np.array_equal(man.view(np.uint32), g[name + "/contact_manifolds"].view(np.uint32))
    w.close()


@pytest.mark.parametrize("name", NAMES)
def test_oracle_chain_scene_matches_golden(oracle, golden_r3, name):
    run_and_compare(oracle, golden_r3, name)


def test_oracle_chain_scene_and_ray_casts_match_reference_build(oracle, ref):
    """Side by side with the reference build on a fresh seed, with AABB queries and ray casts through the chains
    (b2ChainShape::RayCast / ComputeAABB per child, b2World::QueryAABB reports the chain fixture once per child)."""
    a = oracle.world(bh.CHAINS, 70, 0, seed=21, flags=bh.DEFAULT_FLAGS | bh.F_CONTINUOUS)
    r = ref.world(bh.CHAINS, 70, 0, seed=21, flags=bh.DEFAULT_FLAGS | bh.F_CONTINUOUS)
    rng = np.random.default_rng(5)
    hits = 0
    for s in range(200):
        a.step(1)
        r.step(1)
        assert a.contact_count == r.contact_count, "step %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), r.bodies().view(np.uint32)), "step %d" % s
        if s % 20 == 19:
            for _ in range(20):
                p1 = rng.uniform([-24, -2], [24, 24])
                p2 = rng.uniform([-24, -2], [24, 24])
                ha, hr = a.raycast_closest(p1, p2), r.raycast_closest(p1, p2)
                assert (ha is None) == (hr is None), "ray cast at step %d" % s
                if ha is not None:
                    hits += 1
                    assert np.array_equal(ha.view(np.uint32), hr.view(np.uint32)), "ray cast at step %d" % s
                lo = rng.uniform([-22, -1], [18, 18])
                hi = (lo[0] + rng.uniform(0.5, 8), lo[1] + rng.uniform(0.5, 6))
                qa, qr = a.query_aabb(lo, hi), r.query_aabb(lo, hi)
                assert sorted(map(tuple, qa)) == sorted(map(tuple, qr)), "AABB query at step %d" % s
    assert hits > 50
    a.close()
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_chain_scene_bit_exact_vs_golden_default_mode(amd, golden_r3, name):
    os.environ.pop("B2HIP_FORCE_LARGE", None)
    run_and_compare(amd, golden_r3, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_chain_scene_bit_exact_vs_golden_exact_order_mode(amd, golden_r3, monkeypatch, name):
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    run_and_compare(amd, golden_r3, name)
