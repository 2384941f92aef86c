"""Self-test of the multi-rank driver bench.py uses (box2d-mt_amd/python/sharding.ShardedWorld over torch.distributed):
every rank builds the same world of pyramids, the ranks step it sharded by island, rank 0 also steps an unsharded copy, and
after every step every rank must hold the unsharded world bit for bit.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
        tools/shard_selftest.py [--backend gloo|nccl] [--rows 40] [--pyramids 5] [--steps 60]

backend nccl (RCCL) wants one GPU per rank: the all-gather then runs inside the library on the world's stream
(b2hip_shard_connect); gloo lets several ranks share one GPU (torch.distributed's all-gather over device tensors), which is
how tests/test_gpu_sharding.py runs it on a one-GPU box."""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np
import torch
import torch.distributed as dist


def build(w, b2hip, rows, pyramids):
    g = w.create_body(b2hip.STATIC)
    w.create_fixture(g, b2hip.edge_shape((-100.0, 0.0), (100.0 + 1.2 * rows * pyramids + 12.0 * pyramids, 0.0)))
    box = b2hip.box_shape(0.5, 0.5)
    for k in range(pyramids):
        x = np.array([-7.0 + (1.125 * rows + 10.0) * k, 0.75], np.float32)
        dx = np.array([0.5625, 1.25], np.float32)
        dy = np.array([1.125, 0.0], np.float32)
        for i in range(rows):
            y = x.copy()
            for j in range(i, rows):
                b = w.create_body(b2hip.DYNAMIC, (float(y[0]), float(y[1])))
                w.create_fixture(b, box, density=5.0)
                y = y + dy
            x = x + dx


def main():
    if os.environ.get("B2_SELFTEST_WATCHDOG"):
        # (debugging aid: where is every thread after N seconds?)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["B2_SELFTEST_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--rows", type=int, default=40)
    ap.add_argument("--pyramids", type=int, default=5)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--torch-collective", action="store_true", help="with --backend nccl: torch.distributed's all-gather instead of the library's own RCCL")
    ap.add_argument("--no-reference", action="store_true", help="do not compare with an unsharded world (default mode: the "
                    "block partition of large islands depends on the islands a rank holds; set B2HIP_FORCE_LARGE=2 to compare)")
    a = ap.parse_args()
    rank, size = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local % max(ndev, 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(a.backend, rank=rank, world_size=size)
    import b2hip
    import sharding
    w = b2hip.World(device=dev)
    build(w, b2hip, a.rows, a.pyramids)
    sw = sharding.ShardedWorld(w, dist=dist, device="cuda:%d" % dev)
    if a.backend == "nccl" and not a.torch_collective:
        sw.connect_rccl()  # the all-gather inside the library (RCCL on the world's stream), as bench.py --gpus N runs it
    ref = None
    if rank == 0 and not a.no_reference:
        ref = b2hip.World(device=dev)
        build(ref, b2hip, a.rows, a.pyramids)
    solved = 0
    for s in range(a.steps):
        sw.step()
        c = w.counters()
        solved = c["small_islands"] + c["large_islands"]  # (the last step's census is the one reported)
        total = c["islands"]
        mine = hashlib.sha1(w.body_states().tobytes() + w.contacts().tobytes()).digest()
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8).clone()
        all_t = [torch.empty_like(t) for _ in range(size)]
        dist.all_gather(all_t, t)
        if any(bytes(x.numpy()) != mine for x in all_t):
            st, ct = w.body_states(), w.contacts()
            parts = {f: hashlib.sha1(st[f].tobytes()).hexdigest()[:8] for f in st.dtype.names}
            parts.update({"c:" + f: hashlib.sha1(ct[f].tobytes()).hexdigest()[:8] for f in ct.dtype.names})
            print("rank %d: the ranks hold different worlds after step %d (%d contacts, counters %s)\n   %s" % (rank, s, len(ct), w.counters(), parts), flush=True)
            sys.exit(3)
        if ref is not None:
            ref.step()
            if hashlib.sha1(ref.body_states().tobytes() + ref.contacts().tobytes()).digest() != mine:
                print("the sharded world differs from the unsharded one after step %d" % s, flush=True)
                sys.exit(4)
    counts = [None] * size
    dist.all_gather_object(counts, solved)
    if rank == 0:
        # (exact-order mode lists every island twice - found by the small-island traversal, solved on the large-island lists -
        # so only "nobody idle, nobody has everything" is checked)
        if size > 1 and total >= size and (min(counts) == 0 or (sum(counts) == total and max(counts) == total)):
            print("islands were not shared out: %s of %d" % (counts, total), flush=True)
            sys.exit(5)
        print("SHARD-SELFTEST OK ranks=%d steps=%d islands solved per rank (last step)=%s of %d exchange=%d bytes/step" % (
            size, a.steps, counts, total, sw.exchange_bytes), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
