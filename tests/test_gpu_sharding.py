"""Island sharding on the device without a process group: the two "ranks" are two worlds in one
process on one GPU, and the all-reduce(MAX) between them is torch.maximum on their exchange buffers - the kernels that
decide ownership (k_island_classify, k_shard_big), write the records (k_shard_export) and take the other rank's results
(k_shard_import) are the ones a multi-GPU run uses (tests/test_sharding_gloo.py runs the same driver over gloo on the CPU)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import b2harness as bh
import b2hip

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_sharding_gloo import build_field  # noqa: E402
import sharding  # noqa: E402


class TwoRanks:
    """ShardedWorld.step for two worlds side by side; the collective is an element-wise maximum."""

    def __init__(self, worlds):
        import torch
        self.torch = torch
        self.sw = []
        for r, w in enumerate(worlds):
            s = sharding.ShardedWorld(w, dist=None, device="cuda")
            s.rank, s.size = r, len(worlds)
            s._check(s.L.b2hip_set_shard(w.p, r, len(worlds)))
            self.sw.append(s)

    def step(self):
        torch = self.torch
        bufs = []
        for s in self.sw:
            L, p = s.L, s.w.p
            s._check(L.b2hip_step_begin(p, 1.0 / 60.0, 8, 3))
            s._check(L.b2hip_collide(p))
            s._check(L.b2hip_solve(p))
            words = C.c_size_t(0)
            s._check(L.b2hip_shard_exchange_words(p, C.byref(words)))
            buf = torch.empty(int(words.value), dtype=torch.int32, device="cuda")
            s._check(L.b2hip_shard_export(p, C.c_void_p(buf.data_ptr()), int(words.value)))
            bufs.append(buf)
        assert len({b.numel() for b in bufs}) == 1, "the ranks disagree about the size of the world"
        red = bufs[0]
        for b in bufs[1:]:
            red = torch.maximum(red, b)
        torch.cuda.synchronize()
        for s in self.sw:
            L, p = s.L, s.w.p
            s._check(L.b2hip_shard_import(p, C.c_void_p(red.data_ptr()), red.numel()))
            s._check(L.b2hip_sync_fixtures(p))
            s._check(L.b2hip_find_new_contacts(p))
            s._check(L.b2hip_solve_toi(p))
            s._check(L.b2hip_step_end(p))


def snapshot(w):
    return w.body_states().tobytes(), w.contact_count, w.contacts().tobytes()


def test_sharded_field_on_the_device_equals_the_unsharded_run():
    pytest.importorskip("torch")
    worlds = []
    for _ in range(3):
        w = b2hip.World(gravity=(0.0, 0.0), continuous=True)
        build_field(w, 1500, seed=21)
        worlds.append(w)
    ref, pair = worlds[0], TwoRanks(worlds[1:])
    split = 0
    for s in range(120):
        ref.step()
        pair.step()
        want = snapshot(ref)
        for r, w in enumerate(worlds[1:]):
            assert snapshot(w) == want, "rank %d differs from the unsharded world at step %d" % (r, s)
        c0, c1 = worlds[1].counters(), worlds[2].counters()
        # each rank solved only part of the islands (the rest came through the exchange)
        split = max(split, min(c0["small_islands"], c1["small_islands"]))
        assert c0["islands"] == c1["islands"] == ref.counters()["islands"]
    assert split > 10, "the islands were not shared out"
    for w in worlds:
        w.close()


def test_big_islands_are_dealt_round_robin():
    """Three pyramids of 100 rows (5 050 boxes each: above SHARD_BIG_BODIES) over two ranks: one rank solves two of them, the
    other one, and after every exchange the two ranks hold the same world, bit for bit."""
    pytest.importorskip("torch")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from test_gpu_onestep import build_pyramid

    def three(w):
        g = w.create_body(b2hip.STATIC)
        w.create_fixture(g, b2hip.edge_shape((-100.0, 0.0), (500.0, 0.0)))
        box = b2hip.box_shape(0.5, 0.5)
        for k in range(3):
            x = np.array([-7.0 + 140.0 * k, 0.75], np.float32)
            dx = np.array([0.5625, 1.25], np.float32)
            dy = np.array([1.125, 0.0], np.float32)
            for i in range(100):
                y = x.copy()
                for j in range(i, 100):
                    b = w.create_body(b2hip.DYNAMIC, (float(y[0]), float(y[1])))
                    w.create_fixture(b, box, density=5.0)
                    y = y + dy
                x = x + dx

    worlds = []
    for _ in range(2):
        w = b2hip.World()
        three(w)
        worlds.append(w)
    pair = TwoRanks(worlds)
    seen = set()
    for s in range(90):
        pair.step()
        assert snapshot(worlds[0])[0] == snapshot(worlds[1])[0], "the two ranks hold different body states at step %d" % s
        c0, c1 = worlds[0].counters(), worlds[1].counters()
        if c0["large_islands"] + c1["large_islands"] == 3:
            seen.add((c0["large_islands"], c1["large_islands"]))
    assert seen == {(2, 1)}, "big islands were not dealt 2 + 1: %s" % seen
    st = worlds[0].body_states()
    assert np.isfinite(st["px"]).all() and (st["py"][1:] > 0.3).all()
    for w in worlds:
        w.close()
