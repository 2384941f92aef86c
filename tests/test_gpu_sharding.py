"""Island sharding on the device without a process group: the two "ranks" are two worlds in one
process on one GPU, and the all-reduce(MAX) between them is an element-wise maximum of their exchange buffers - the kernels that
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


class Hip:
    """hipMalloc / hipMemcpy of the runtime libb2hip.so itself uses (no second framework in the process)."""

    def __init__(self):
        self.L = C.CDLL("libamdhip64.so")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), max(nbytes, 4)) == 0
        return p

    def free(self, p):
        self.L.hipFree(p)

    def to_host(self, p, words):
        out = np.empty(words, np.int32)
        assert self.L.hipMemcpy(out.ctypes.data, p, 4 * words, 2) == 0
        return out

    def to_device(self, p, arr):
        assert self.L.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0


class TwoRanks:
    """ShardedWorld.step for two worlds side by side; the collective (an all-gather of the ranks' slabs) goes through the host."""

    def __init__(self, worlds):
        self.hip = Hip()
        self.sw = []
        for r, w in enumerate(worlds):
            s = sharding.ShardedWorld(w, dist=None, device="cpu")
            s.rank, s.size = r, len(worlds)
            s._check(s.L.b2hip_set_shard(w.p, r, len(worlds)))
            self.sw.append(s)

    def step(self):
        n = len(self.sw)
        for s in self.sw:
            L, p = s.L, s.w.p
            s._check(L.b2hip_step_begin(p, 1.0 / 60.0, 8, 3))
            s._check(L.b2hip_collide(p))
            s._check(L.b2hip_solve(p))
        # every rank has counted every rank's slab (the island build is replicated): they must agree
        sizes = []
        for s in self.sw:
            words = (C.c_size_t * n)()
            s._check(s.L.b2hip_shard_slab_words(s.w.p, words, n))
            sizes.append(list(words))
        assert all(sz == sizes[0] for sz in sizes), "the ranks disagree about the slab sizes: %s" % sizes
        stride = max(max(sizes[0]), 1)
        self.exchange_words = stride * n
        slabs = []
        for s in self.sw:
            buf = self.hip.alloc(4 * stride)
            s._check(s.L.b2hip_shard_export(s.w.p, buf, stride))
            slabs.append(self.hip.to_host(buf, stride))
            self.hip.free(buf)
        gathered = np.concatenate(slabs)  # (the all-gather)
        for s in self.sw:
            L, p = s.L, s.w.p
            buf = self.hip.alloc(4 * stride * n)
            self.hip.to_device(buf, gathered)
            s._check(L.b2hip_shard_import(p, buf, stride))
            s._check(L.b2hip_sync_fixtures(p))
            s._check(L.b2hip_find_new_contacts(p))
            s._check(L.b2hip_solve_toi(p))
            s._check(L.b2hip_step_end(p))
            self.hip.free(buf)


def snapshot(w):
    return w.body_states().tobytes(), w.contact_count, w.contacts().tobytes()


def test_sharded_field_on_the_device_equals_the_unsharded_run():
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
    assert split >= 5, "the islands were not shared out"
    for w in worlds:
        w.close()


def test_big_islands_are_dealt_round_robin():
    """Three pyramids of 100 rows (5 050 boxes each: above SHARD_BIG_BODIES) over two ranks: one rank solves two of them, the
    other one, and after every exchange the two ranks hold the same world, bit for bit."""
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
        # (while the pyramids are still coming together their islands are smaller than SHARD_BIG_BODIES and go by hash)
        if s >= 60 and c0["large_islands"] + c1["large_islands"] == 3:
            seen.add((c0["large_islands"], c1["large_islands"]))
    assert seen == {(2, 1)}, "big islands were not dealt 2 + 1: %s" % seen
    st = worlds[0].body_states()
    assert np.isfinite(st["px"]).all() and (st["py"][1:] > 0.3).all()
    for w in worlds:
        w.close()


def test_rccl_all_gather_inside_the_library_on_one_rank(monkeypatch):
    """b2hip_shard_connect opens librccl, makes a communicator and from then on b2hip_step queues export -> ncclAllGather ->
    import on the world's own stream. A one-GPU box has no peer, but a communicator of ONE rank (B2HIP_SHARD_LOOPBACK=1)
    runs the same calls: the world must step exactly like an unconnected one, and report the bytes it gathered."""
    monkeypatch.setenv("B2HIP_SHARD_LOOPBACK", "1")
    L = b2hip.lib()
    L.b2hip_shard_unique_id.argtypes = [C.c_void_p]
    L.b2hip_shard_connect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.b2hip_shard_exchange_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    a, b = b2hip.World(gravity=(0.0, 0.0), continuous=True), b2hip.World(gravity=(0.0, 0.0), continuous=True)
    build_field(a, 600, seed=5)
    build_field(b, 600, seed=5)
    ident = (C.c_ubyte * 128)()
    assert L.b2hip_shard_unique_id(ident) == 0, L.b2hip_last_error()
    assert L.b2hip_shard_connect(b.p, ident, 0, 1) == 0, L.b2hip_last_error()
    for s in range(60):
        a.step()
        b.step()
        assert snapshot(a) == snapshot(b), "step %d" % s
    n = C.c_size_t(0)
    assert L.b2hip_shard_exchange_bytes(b.p, C.byref(n)) == 0 and n.value >= 4
    a.close()
    b.close()


def run_selftest(cmd, root, env):
    """tools/shard_selftest.py under torchrun. It takes ~15 s; a rank that is still running after 150 s dumps the stack of
    every thread and exits (B2_SELFTEST_WATCHDOG), so a hang shows where it hangs instead of eating the suite's time."""
    import gc
    import subprocess
    gc.collect()  # (worlds of earlier tests that are only waiting for the collector: their streams go back first)
    env = dict(env, B2_SELFTEST_WATCHDOG="150")
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0 and "SHARD-SELFTEST OK" in r.stdout, "stdout:\n%s\nstderr:\n%s" % (r.stdout[-3000:], r.stderr[-6000:])


def test_sharded_world_over_a_process_group():
    """The driver bench.py --gpus N uses (sharding.ShardedWorld over torch.distributed), two ranks in two processes sharing
    this box's one GPU over gloo (RCCL wants a GPU per rank): every rank holds the unsharded world after every step."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "tools", "shard_selftest.py"), "--backend", "gloo", "--rows", "40",
           "--pyramids", "5", "--steps", "50"]
    # exact-order mode: what a rank computes for an island must not depend on which other islands it owns (in default mode
    # the block partition of the large islands spans the islands a rank holds, so only the ranks agree with each other)
    env = dict(os.environ, B2HIP_FORCE_LARGE="2")
    run_selftest(cmd, root, env)


def test_sharded_world_over_a_process_group_default_mode():
    """Same, default (block solver) mode: the ranks agree with each other bit for bit after every exchange."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29542", os.path.join(root, "tools", "shard_selftest.py"), "--backend", "gloo", "--rows", "40",
           "--pyramids", "5", "--steps", "50", "--no-reference"]
    run_selftest(cmd, root, dict(os.environ))
