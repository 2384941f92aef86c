"""Island-granular sharding of a world across the GPUs of one node.

Islands only share static bodies, so a world made of disjoint islands (config 4: several pyramids on one
ground) shards with NO data-path collective: rank r builds and steps the islands it owns. What is
collective is only the optional assembly of the host-visible world state (all ranks see all bodies):
one all-gather of the per-body state rows, RCCL over xGMI for device tensors (backend "nccl"), gloo on
CPU in the tests.

The plan is static here (pyramid k -> rank k % world_size, the same rule scenes.h uses to build a
shard); body ids are mapped back to the ids of the unsharded scene so that a sharded run can be
compared bit for bit with an unsharded one.
"""
import ctypes as C

import numpy as np


def pyramid_bodies(rows):
    return rows * (rows + 1) // 2


def shard_of_pyramid(k, world_size):
    return k % world_size


def global_body_ids(rows, count, rank, world_size):
    """Ids (in the unsharded scene) of the bodies rank `rank` owns, in its local creation order.
    Local body 0 is the shared static ground (global id 0)."""
    per = pyramid_bodies(rows)
    ids = [0]
    for k in range(count):
        if shard_of_pyramid(k, world_size) == rank:
            ids.extend(range(1 + k * per, 1 + (k + 1) * per))
    return np.asarray(ids, np.int64)


def gather_world_state(local_state, rows, count, rank, world_size, dist=None, device=None):
    """All-gather of per-body state rows (any float32 [n_local, k] array): returns the [n_global, k] array of
    the unsharded scene on every rank. With world_size == 1 or dist None it is a local scatter."""
    import torch
    n_global = 1 + count * pyramid_bodies(rows)
    k = local_state.shape[1]
    out = np.zeros((n_global, k), np.float32)
    if dist is None or world_size == 1:
        out[global_body_ids(rows, count, rank, world_size)] = local_state
        return out
    # equal-sized slabs: pad every rank to the largest shard (variable counts -> padded all_gather)
    sizes = [len(global_body_ids(rows, count, r, world_size)) for r in range(world_size)]
    slab = max(sizes)
    send = torch.zeros((slab, k), dtype=torch.float32, device=device)
    send[:local_state.shape[0]] = torch.from_numpy(np.ascontiguousarray(local_state)).to(send.device)
    recv = [torch.empty_like(send) for _ in range(world_size)]
    dist.all_gather(recv, send)
    for r in range(world_size):
        ids = global_body_ids(rows, count, r, world_size)
        out[ids] = recv[r][:sizes[r]].cpu().numpy()
    return out


class ShardedWorld:
    """One world over the ranks of a process group, sharded by island (include/b2hip.h, "One world over the GPUs of a node").

    Every rank builds the same world with the same calls (`world` is a b2hip.World on any library that exports the C ABI:
    libb2hip.so on a GPU, the tests' oracle shim on a CPU) and steps it through this object: Collide, the island build,
    the broad-phase and the TOI phase run replicated; b2Island::Solve runs only for the islands the rank owns; one
    all-reduce(MAX) per step over an int32 buffer - RCCL over xGMI for device tensors (backend "nccl"), gloo for CPU
    tensors - carries every solved island to every rank. With one rank (dist None) it is the plain phase sequence."""

    def __init__(self, world, dist=None, device="cpu"):
        self.w = world
        self.L = world.L
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank() if dist is not None else 0
        self.size = dist.get_world_size() if dist is not None else 1
        self.L.b2hip_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.b2hip_shard_exchange_words.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        self.L.b2hip_shard_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.b2hip_shard_import.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self._check(self.L.b2hip_set_shard(world.p, self.rank, self.size))
        self._buf = None
        self.exchange_bytes = 0

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError("b2hip error %d: %s" % (rc, self.L.b2hip_last_error().decode()))

    def exchange(self):
        """solved islands -> every rank (between b2hip_solve and b2hip_sync_fixtures)"""
        import torch
        words = C.c_size_t(0)
        self._check(self.L.b2hip_shard_exchange_words(self.w.p, C.byref(words)))
        n = int(words.value)
        if self._buf is None or self._buf.numel() < n:
            self._buf = torch.empty(max(n, 1) * 5 // 4 + 64, dtype=torch.int32, device=self.device)
        buf = self._buf[:n]
        self._check(self.L.b2hip_shard_export(self.w.p, C.c_void_p(buf.data_ptr()), n))
        if self.dist is not None and self.size > 1:
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.MAX)
            if buf.is_cuda:
                torch.cuda.current_stream().synchronize()  # the import below runs on the world's own stream
        self._check(self.L.b2hip_shard_import(self.w.p, C.c_void_p(buf.data_ptr()), n))
        self.exchange_bytes = 4 * n

    def step(self, dt=1.0 / 60.0, vel_iters=8, pos_iters=3):
        L, p = self.L, self.w.p
        self._check(L.b2hip_step_begin(p, dt, vel_iters, pos_iters))
        self._check(L.b2hip_collide(p))
        self._check(L.b2hip_solve(p))
        if self.size > 1:
            self.exchange()
        self._check(L.b2hip_sync_fixtures(p))
        self._check(L.b2hip_find_new_contacts(p))
        self._check(L.b2hip_solve_toi(p))
        self._check(L.b2hip_step_end(p))
