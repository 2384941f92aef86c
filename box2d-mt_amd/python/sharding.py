"""One world over the GPUs of one node, sharded by island (protocol: include/b2hip.h, kernels: csrc/b2d_kernels_shard.h).

Every rank builds the same world with the same calls and steps it through a ShardedWorld: Collide, the island build, the
broad-phase and the TOI phase run replicated; b2Island::Solve runs for the islands a rank owns; after the solve the ranks
all-gather the records of what they own - one slab per rank, sized by what it owns (every rank has counted every rank's slab
during the replicated island build, so no sizes are exchanged).

  * connect_rccl(): the all-gather runs INSIDE the library (RCCL over xGMI on the world's own stream, no host
    synchronisation; torch.distributed - any backend - only carries the 128-byte RCCL id to the ranks once). step() is then
    plain b2hip_step. This is what bench.py --gpus N uses.
  * otherwise the all-gather is torch.distributed's (gloo on CPU tensors in the tests, where `world` is on the oracle's
    ABI shim; nccl on device tensors also works): export -> all_gather_into_tensor -> import, per step.
With one rank it is the plain phase sequence.
"""
import ctypes as C


class ShardedWorld:
    def __init__(self, world, dist=None, device="cpu"):
        self.w = world
        self.L = world.L
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank() if dist is not None else 0
        self.size = dist.get_world_size() if dist is not None else 1
        self.L.b2hip_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.b2hip_shard_slab_words.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_int]
        self.L.b2hip_shard_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.b2hip_shard_import.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self._check(self.L.b2hip_set_shard(world.p, self.rank, self.size))
        self._send = self._recv = None
        self.exchange_bytes = 0
        self.connected = False

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError("b2hip error %d: %s" % (rc, self.L.b2hip_last_error().decode()))

    def connect_rccl(self):
        """RCCL driven from the C layer: rank 0 makes the id, torch.distributed broadcasts its 128 bytes, every rank connects."""
        import torch
        self.L.b2hip_shard_unique_id.argtypes = [C.c_void_p]
        self.L.b2hip_shard_connect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.L.b2hip_shard_exchange_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            self._check(self.L.b2hip_shard_unique_id(ident))
        t = torch.tensor(list(ident), dtype=torch.uint8, device=self.device if self.dist is not None and self.dist.get_backend() == "nccl" else "cpu")
        if self.dist is not None and self.size > 1:
            self.dist.broadcast(t, src=0)
        ident = (C.c_ubyte * 128)(*t.cpu().tolist())
        self._check(self.L.b2hip_shard_connect(self.w.p, ident, self.rank, self.size))
        self.connected = True

    def exchange(self):
        """solved islands -> every rank (between b2hip_solve and b2hip_sync_fixtures), by torch.distributed's all-gather"""
        import torch
        words = (C.c_size_t * self.size)()
        self._check(self.L.b2hip_shard_slab_words(self.w.p, words, self.size))
        stride = max(max(words), 1)
        if self._send is None or self._send.numel() < stride:
            # (the old buffers are dropped only after the world's stream has finished with them: b2hip_shard_import of the
            #  previous step reads _recv on that stream)
            if self._recv is not None and self._recv.is_cuda:
                torch.cuda.synchronize()
            cap = stride * 5 // 4 + 64
            self._send = torch.zeros(cap, dtype=torch.int32, device=self.device)
            self._recv = torch.zeros(cap * self.size, dtype=torch.int32, device=self.device)
            if self._send.is_cuda:
                torch.cuda.synchronize()  # (the fill runs on torch's stream, the export below on the world's: it must have landed)
        send = self._send[:stride]
        recv = self._recv[:stride * self.size]
        self._check(self.L.b2hip_shard_export(self.w.p, C.c_void_p(send.data_ptr()), stride))  # (returns with the slab written)
        if self.dist is not None and self.size > 1:
            self.dist.all_gather_into_tensor(recv, send)
            if recv.is_cuda:
                torch.cuda.current_stream().synchronize()  # the import below runs on the world's own stream
        else:
            recv[:stride] = send
        self._check(self.L.b2hip_shard_import(self.w.p, C.c_void_p(recv.data_ptr()), stride))
        self.exchange_bytes = 4 * stride * self.size

    def step(self, dt=1.0 / 60.0, vel_iters=8, pos_iters=3):
        L, p = self.L, self.w.p
        if self.connected:
            self._check(L.b2hip_step(p, dt, vel_iters, pos_iters))  # (the library exchanges on its own stream)
            n = C.c_size_t(0)
            L.b2hip_shard_exchange_bytes(p, C.byref(n))
            self.exchange_bytes = int(n.value)
            return
        self._check(L.b2hip_step_begin(p, dt, vel_iters, pos_iters))
        self._check(L.b2hip_collide(p))
        self._check(L.b2hip_solve(p))
        if self.size > 1:
            self.exchange()
        self._check(L.b2hip_sync_fixtures(p))
        self._check(L.b2hip_find_new_contacts(p))
        self._check(L.b2hip_solve_toi(p))
        self._check(L.b2hip_step_end(p))


GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class SpatialWorld:
    """One world over the ranks of torch.distributed by SPATIAL OWNERSHIP (include/b2hip.h: b2hip_shard_spatial;
    box2d-mt_amd/csrc/b2d_kernels_spatial.h): every rank builds the same world, owns the bodies of its strip along x, and
    evaluates / solves / moves those only; the library exchanges what the others need inside b2hip_step - over RCCL on the
    world's own stream (`nccl` backend: b2hip_shard_connect), or through an all-gather of host memory that this class hands
    it (`gloo`: the CPU tests over the oracle's shim, functional runs of several ranks on one GPU)."""

    def __init__(self, world, dist, owners=None, device="cpu"):
        self.w = world
        self.L = world.L
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.size = dist.get_world_size() if dist is not None else 1
        L = self.L
        L.b2hip_shard_spatial.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.b2hip_set_shard_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.connected = False
        self.gathers = 0
        self.gather_bytes = 0
        self._group = None
        if dist is not None and self.size > 1 and dist.get_backend() == "nccl":
            import torch
            s = ShardedWorld.__new__(ShardedWorld)  # (only for its RCCL hand-shake: rank 0's id broadcast, every rank connects)
            s.w, s.L, s.dist, s.device, s.rank, s.size, s.connected = world, L, dist, device, self.rank, self.size, False
            why = ""
            try:
                s.connect_rccl()
            except Exception as e:  # noqa: BLE001
                why = str(e)
            # (the library's own communicator cannot be had - librccl not found ... - on every rank alike: the exchange goes
            # through an all-gather of host memory over a gloo group instead; a rank on its own with the problem is an error)
            ok = torch.tensor([0 if why else 1], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.SUM)
            if int(ok.item()) == self.size:
                self.connected = True
            elif int(ok.item()) == 0:
                # (loud: from here on "RCCL saw N ranks" is false for this world - the exchange is an all-gather of HOST memory)
                import sys
                sys.stderr.write("sharding.SpatialWorld: b2hip_shard_connect failed on every rank (%s): the exchange falls back to torch.distributed over gloo - NOT RCCL\n" % why)
                self._group = dist.new_group(backend="gloo")
            else:
                raise RuntimeError("b2hip_shard_connect failed on some ranks only: %s" % why)
        if not self.connected:
            self._cb = GATHER_FN(self._gather)
            rc = L.b2hip_set_shard_gather(world.p, C.cast(self._cb, C.c_void_p), None)
            if rc < 0:
                raise RuntimeError("b2hip error %d: %s" % (rc, L.b2hip_last_error().decode()))
        own = None
        if owners is not None:
            import numpy as np
            self._owners = np.ascontiguousarray(owners, np.uint8)
            own = self._owners.ctypes.data_as(C.c_void_p)
        rc = L.b2hip_shard_spatial(world.p, self.rank, self.size, own)
        if rc < 0:
            raise RuntimeError("b2hip error %d: %s" % (rc, L.b2hip_last_error().decode()))

    def _gather(self, user, send, nbytes, recv):
        import numpy as np
        import torch
        try:
            self.gathers += 1
            self.gather_bytes += nbytes * self.size
            src = torch.from_numpy(np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(send)))
            dst = torch.from_numpy(np.ctypeslib.as_array((C.c_ubyte * (nbytes * self.size)).from_address(recv)))
            if self.dist is None or self.size == 1:
                dst[:nbytes] = src
            else:
                self.dist.all_gather_into_tensor(dst, src, group=self._group)
            return 0
        except Exception:  # noqa: BLE001 (reported by the library as a failed collective)
            return 1

    def step(self, dt=1.0 / 60.0, vel_iters=8, pos_iters=3):
        rc = self.L.b2hip_step(self.w.p, C.c_float(dt), vel_iters, pos_iters)
        if rc < 0:
            raise RuntimeError("b2hip error %d: %s" % (rc, self.L.b2hip_last_error().decode()))
