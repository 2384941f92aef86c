"""CPU checks of tests/golden/settled_windows.npz (aggregates of the reference build's run to the windows bench.py times):
the fixture is what tests/golden/make_golden_settled.py produces - re-made here for the 10 011-box pyramid where the reference
build is present (8 s; the Tumbler's window takes it hours) - and well-formed for every scene it holds."""
import os

import numpy as np
import pytest

import b2harness as bh
import quality_util as qu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "settled_windows.npz")


def test_fixture_is_well_formed():
    g = np.load(GOLDEN)
    scenes = sorted({k.split("/")[0] for k in g.files})
    assert "config2_pyramid141" in scenes and "config4_pyramid316" in scenes
    for name in scenes:
        sc, p0, p1, seed, flags, first, last, every = (int(v) for v in g[name + "/params"])
        steps, table = g[name + "/steps"], g[name + "/table"]
        assert [str(k) for k in g[name + "/keys"]] == list(qu.KEYS)
        assert list(steps) == list(range(first, last + 1, every))
        assert table.shape == (len(steps), len(qu.KEYS)) and np.isfinite(table).all()
        assert (table[:, qu.KEYS.index("contacts")] >= table[:, qu.KEYS.index("touching")]).all()
        assert (table[:, qu.KEYS.index("touching")] > 0).all()


@pytest.mark.skipif(not bh.have_ref(), reason="the reference build (oracle/_ref) is only present in the build container")
def test_the_script_reproduces_the_fixture_for_config_2():
    g = np.load(GOLDEN)
    name = "config2_pyramid141"
    sc, p0, p1, seed, flags, first, last, every = (int(v) for v in g[name + "/params"])
    w = bh.Harness(bh.REF_LIB).world(sc, p0, p1, seed=seed, flags=flags, threads=8)
    w.step(first)
    steps, table = qu.window(w, first, last, every)
    w.close()
    assert list(steps) == list(g[name + "/steps"])
    assert np.array_equal(table, g[name + "/table"]), "the reference build is deterministic whatever the thread count (README.md:161-175): the same table"
