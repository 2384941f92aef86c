"""N ranks of a spatially sharded world inside ONE process (test infrastructure): one world per rank on the same device (or
on the CPU oracle's shim), one thread per rank, and an all-gather over host memory made of a barrier - what
b2hip_set_shard_gather asks of the caller (include/b2hip.h). Every rank steps in its own thread; ctypes releases the GIL
while the library runs and takes it again for the callback."""
import ctypes as C
import threading

import numpy as np

GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class ShardStats(C.Structure):
    _fields_ = [("rank", C.c_int32), ("count", C.c_int32), ("owned_bodies", C.c_int32), ("owned_proxies", C.c_int32),
                ("owned_contacts", C.c_int32), ("islands_solved", C.c_int32), ("constraint_rows", C.c_int32), ("pad", C.c_int32),
                ("migrated_bodies", C.c_int64), ("resolutions", C.c_int64), ("bytes_received_last_step", C.c_int64),
                ("pairs_sent", C.c_int64), ("toi_redos", C.c_int64)]


class ThreadGather:
    """The collective of `n` ranks that live in one process."""

    def __init__(self, n, timeout=120.0):
        self.n = n
        self.slots = [None] * n
        self.barrier = threading.Barrier(n, timeout=timeout)
        self.calls = 0
        self.bytes = 0
        # the ranks share ONE device here: they take turns on it (a rank's stream is idle when its all-gather is called and
        # when its step returns), so that a kernel whose workgroups must all be resident - the block solver - has the device
        # to itself as it would on a rank's own GPU
        self.device = threading.Lock()

    def callback(self, rank):
        def fn(user, send, nbytes, recv):
            try:
                self.slots[rank] = C.string_at(send, nbytes)
                self.device.release()
                self.barrier.wait()
                for r in range(self.n):
                    C.memmove(recv + r * nbytes, self.slots[r], nbytes)
                if rank == 0:
                    self.calls += 1
                    self.bytes += nbytes * self.n
                self.barrier.wait()
                self.device.acquire()
                return 0
            except Exception:  # (a broken barrier: another rank failed)
                return 1
        return GATHER_FN(fn)


class SpatialRanks:
    """worlds[r] = (stepper, device pointer): the same world built n times; rank r owns what b2hip_shard_spatial deals it."""

    def __init__(self, L, worlds, owners=None):
        self.L = L
        self.worlds = worlds
        self.n = len(worlds)
        self.gather = ThreadGather(self.n)
        self.cbs = []
        L.b2hip_shard_spatial.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.b2hip_set_shard_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.b2hip_get_shard_stats.argtypes = [C.c_void_p, C.POINTER(ShardStats)]
        L.b2hip_get_body_owners.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for r, (_, dev) in enumerate(worlds):
            own = None if owners is None else np.ascontiguousarray(owners, np.uint8).ctypes.data_as(C.c_void_p)
            rc = L.b2hip_shard_spatial(C.c_void_p(dev), r, self.n, own)
            assert rc == 0, L.b2hip_last_error()
            cb = self.gather.callback(r)
            self.cbs.append(cb)
            assert L.b2hip_set_shard_gather(C.c_void_p(dev), C.cast(cb, C.c_void_p), None) == 0

    def step(self, fn=None):
        """fn(rank, stepper) steps one rank's world once (default: stepper.step(1)); returns when all ranks have."""
        errs = [None] * self.n

        def run(r):
            self.gather.device.acquire()
            try:
                (fn or (lambda _r, s: s.step(1)))(r, self.worlds[r][0])
            except BaseException as e:  # noqa: B902
                errs[r] = e
                self.gather.barrier.abort()
            finally:
                try:
                    self.gather.device.release()
                except RuntimeError:
                    pass

        ts = [threading.Thread(target=run, args=(r,)) for r in range(self.n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in errs:
            if e is not None:
                raise e

    def stats(self, r):
        st = ShardStats()
        assert self.L.b2hip_get_shard_stats(C.c_void_p(self.worlds[r][1]), C.byref(st)) == 0
        return st

    def owners(self, r, n):
        out = np.zeros(n, np.uint8)
        got = self.L.b2hip_get_body_owners(C.c_void_p(self.worlds[r][1]), n, out.ctypes.data_as(C.c_void_p))
        assert got >= 0, self.L.b2hip_last_error()
        return out

    def own_rows(self, r, n):
        """(ids, rows[k, 10] as uint32): the packed rows of rank r's bodies as the last step brought them to the host"""
        ids = np.zeros(n, np.int32)
        rows = np.zeros((n, 10), np.uint32)
        self.L.b2hip_get_own_body_states.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        got = self.L.b2hip_get_own_body_states(C.c_void_p(self.worlds[r][1]), n, ids.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p))
        assert got >= 0, self.L.b2hip_last_error()
        return ids[:got], rows[:got]
