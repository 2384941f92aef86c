"""b2World::QueryAABB / RayCast served from the device's fat AABBs through the drop-in host layer.

The reference answers both from its dynamic tree; here the host walks the fat AABBs of all fixtures (one read-back per
step, on demand) in creation order and uses the same shape ray casts. The SET a box query reports and the closest hit of
a ray do not depend on the visiting order, so they are compared exactly.
"""
import numpy as np
import pytest

import b2harness as bh

CASES = [("rain", bh.RAIN, dict(p0=200, seed=3), 120), ("sensors", bh.SENSORS, dict(p0=40, seed=5), 100),
         ("field", bh.FIELD, dict(p0=600, p1=0, seed=9), 40), ("circles", bh.CIRCLE_STACK, dict(p0=8, p1=6), 80)]


def compare(a, b, queries=150):
    assert np.array_equal(a.bodies(), b.bodies())
    pos = a.bodies()[:, :2]
    lo, hi = pos.min(axis=0) - 2, pos.max(axis=0) + 2
    rng = np.random.default_rng(7)
    reported = hits = 0
    for k in range(queries):
        c, e = rng.uniform(lo, hi), rng.uniform(0.2, 6.0, 2)
        qa, qb = a.query_aabb(c - e, c + e), b.query_aabb(c - e, c + e)
        assert np.array_equal(qa, qb), "QueryAABB %d reports another set" % k
        reported += len(qa)
        p1, p2 = rng.uniform(lo, hi), rng.uniform(lo, hi)
        ra, rb = a.raycast_closest(p1, p2), b.raycast_closest(p1, p2)
        assert (ra is None) == (rb is None), "ray %d: hit / miss differs" % k
        if ra is not None:
            hits += 1
            assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32)), "ray %d: closest hit differs" % k
    assert reported > 0 and hits > 0


@pytest.mark.parametrize("name,scene,kw,steps", CASES)
def test_queries_match_reference(ref, oracle, name, scene, kw, steps):
    """CPU: host layer over the C oracle against the reference's tree queries."""
    a, b = ref.world(scene, **kw), oracle.world(scene, **kw)
    a.step(steps)
    b.step(steps)
    compare(a, b)
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,kw,steps", CASES)
def test_device_queries_match_oracle(monkeypatch, amd, oracle, name, scene, kw, steps):
    """GPU: the fat AABBs read back from the device serve the same answers (exact-order mode: rain piles grow past the
    128-row exact tier, and the coloured order is not the oracle's)."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = amd.world(scene, **kw), oracle.world(scene, **kw)
    a.step(steps)
    b.step(steps)
    compare(a, b)
    # a fixture created after the last step is visible to queries at once (its box is still on the host)
    a.close()
    b.close()
