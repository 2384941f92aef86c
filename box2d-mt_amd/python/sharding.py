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
