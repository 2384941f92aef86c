"""The N > 1 path of bench.py on the CPU: one world over 2 and 4 PROCESSES (torch.distributed, gloo) by spatial ownership
(include/b2hip.h: b2hip_shard_spatial; box2d-mt_amd/python/sharding.SpatialWorld drives it, the all-gather of host memory the
library asks for is torch.distributed's), on the C oracle behind the same C ABI (oracle/b2o_step.c restates the protocol of
box2d-mt_amd/csrc/b2d_kernels_spatial.h serially: owner table, gated phases, the exchange of rows / fat AABBs / new pairs,
migration of components that a new contact joins over an ownership boundary). Every rank must hold, after every step,
exactly the unsharded world: body states and contact counts, bit for bit.

Also here, in one process (threads + a barrier as the collective: tests/spatial_util.py), more scenes on the oracle: joints,
piles, rain. The product's kernels are pinned the same way on the GPU (tests/test_gpu_spatial.py), where the parts this shim
does not restate - contacts created inside TOI sub-steps merged over the ranks, the lean exchange - are covered too.
Reference: results independent of the worker count (README.md:161-175, TestMT.cpp:91-110)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_sharding_gloo import build_field  # noqa: E402  (the bounded zero-gravity field with a few jointed pairs)


def build_pyramids(w, rows, count):
    """`count` pyramids of `rows` rows side by side on one ground edge (small)"""
    import b2hip
    g = w.create_body(b2hip.STATIC)
    w.create_fixture(g, b2hip.edge_shape((-50.0, 0.0), (50.0 + 1.5 * rows * count, 0.0)))
    box = b2hip.box_shape(0.5, 0.5)
    for k in range(count):
        x = np.array([-7.0 + k * (1.125 * rows + 4.0), 0.75], np.float32)
        dx = np.array([0.5625, 1.25], np.float32)
        dy = np.array([1.125, 0.0], np.float32)
        for i in range(rows):
            y = x.copy()
            for j in range(i, rows):
                b = w.create_body(b2hip.DYNAMIC, (float(y[0]), float(y[1])))
                w.create_fixture(b, box, density=5.0)
                y = y + dy
            x = x + dx


SCENES = {
    # name: (builder, args, gravity, steps)
    "field": (build_field, dict(n=260, seed=12), (0.0, 0.0), 120),
    "pyramids": (build_pyramids, dict(rows=9, count=4), (0.0, -10.0), 90),
}


def _make(name, lib):
    import b2hip
    builder, kw, gravity, steps = SCENES[name]
    w = b2hip.World(gravity=gravity, library=lib)
    builder(w, **kw)
    return w, steps


def _worker(rank, world_size, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import b2harness as bh
    import b2hip
    import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    lib = b2hip.load(bh.ORACLE_LIB, optional_ok=True)
    b2hip._lib = lib
    w, steps = _make(name, lib)
    sw = sharding.SpatialWorld(w, dist=dist)
    trace = []
    for s in range(steps):
        sw.step()
        trace.append((w.body_states().tobytes(), w.contact_count))
    owners = np.zeros(w.body_count, np.uint8)
    lib.b2hip_get_body_owners.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int, __import__("ctypes").c_void_p]
    lib.b2hip_get_body_owners(w.p, w.body_count, owners.ctypes.data_as(__import__("ctypes").c_void_p))
    dist.barrier()
    q.put((rank, trace, owners.tobytes(), sw.gathers, sw.gather_bytes))
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world_size", [("field", 2), ("field", 4), ("pyramids", 2), ("pyramids", 4)])
def test_world_over_gloo_ranks_by_spatial_ownership_equals_the_unsharded_run(built_libs, name, world_size):
    import torch.multiprocessing as mp
    import b2harness as bh
    import b2hip
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() * 7 + world_size * 131 + len(name)) % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world_size, port, name, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    got, owners = {}, {}
    for _ in range(world_size):
        rank, trace, own, gathers, nbytes = q.get(timeout=600)
        got[rank] = trace
        owners[rank] = np.frombuffer(own, np.uint8)
        assert gathers > 0 and nbytes > 0
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lib = b2hip.load(bh.ORACLE_LIB, optional_ok=True)
    b2hip._lib = lib
    w, steps = _make(name, lib)
    for s in range(steps):
        w.step()
        want = (w.body_states().tobytes(), w.contact_count)
        for rank in range(world_size):
            assert got[rank][s][1] == want[1], "rank %d: contact count at step %d" % (rank, s)
            assert got[rank][s][0] == want[0], "rank %d: body states differ from the unsharded run at step %d" % (rank, s)
    dyn = w.body_states()["flags"] & 3
    w.close()
    # the owner table is the same on every rank, and every rank owns a share
    for rank in range(1, world_size):
        assert np.array_equal(owners[rank], owners[0])
    counts = np.bincount(owners[0][dyn != 0], minlength=world_size)
    assert counts.sum() == int((dyn != 0).sum()), "owners per rank: %s" % counts.tolist()
    # (the dense bounded field percolates - fat AABBs of 260 bodies in a 28 m box form one component after a while, which one
    # rank then owns: the migration path at work; the pyramids stay one per strip)
    if name == "pyramids":
        assert (counts > 0).all() and counts.max() <= 2 * counts.sum() // world_size, "owners per rank: %s" % counts.tolist()


def test_oracle_ranks_in_one_process_more_scenes(built_libs):
    """The same protocol with threads as ranks (tests/spatial_util.py) on scenes with joints, piles and rain: every rank equals
    the unsharded oracle world after every step; bodies migrate (the rain piles up over the strip boundaries)."""
    import b2harness as bh
    import b2hip
    from spatial_util import SpatialRanks
    orc = bh.Harness(bh.ORACLE_LIB)
    L = b2hip.load(bh.ORACLE_LIB, optional_ok=True)
    plain = bh.F_SLEEP | bh.F_WARM
    migrated = 0
    for scene, p0, p1, ranks, steps, flags in ((bh.RAIN, 300, 0, 3, 150, plain), (bh.VEHICLES, 20, 2, 2, 120, plain), (bh.PILES, 60, 5, 4, 120, plain),
                                                (bh.FIELD, 1500, 0, 4, 60, plain), (bh.MACHINES, 20, 2, 2, 100, plain), (bh.PYRAMID, 10, 2, 2, 60, plain | bh.F_CONTINUOUS)):
        ref = orc.world(scene, p0, p1, seed=3, flags=flags)
        ws = [orc.world(scene, p0, p1, seed=3, flags=flags) for _ in range(ranks)]
        sr = SpatialRanks(L, [(w, w.device_world()) for w in ws])
        for s in range(steps):
            ref.step(1)
            sr.step()
            rb = ref.bodies().view(np.uint32)
            for r, w in enumerate(ws):
                assert w.contact_count == ref.contact_count, "scene %d step %d rank %d" % (scene, s, r)
                assert np.array_equal(w.bodies().view(np.uint32), rb), "scene %d step %d rank %d" % (scene, s, r)
        migrated += sr.stats(0).resolutions
        for w in ws:
            w.close()
        ref.close()
    assert migrated > 0, "no component ever crossed an ownership boundary: the migration path was not exercised"
