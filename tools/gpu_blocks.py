"""Block solver on one pyramid: per-step solver kernel time, partition figures, phase stamps of workgroup 0.
Usage (GPU box): python tools/gpu_blocks.py [rows=141] [steps=200] [report_every=20]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import numpy as np  # noqa: E402
import b2harness as bh  # noqa: E402
import b2hip  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 141
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
every = int(sys.argv[3]) if len(sys.argv) > 3 else 20
amd = bh.Harness(bh.AMD_LIB)
L = b2hip.lib()
L.b2hip_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
L.b2hip_get_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_char), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)]
L.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(bh.PYRAMID, rows, 1, flags=bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM)
dev = C.c_void_p(w.device_world())
L.b2hip_set_kernel_timing(dev, 1)
t_acc = 0.0
for s in range(steps):
    t0 = time.perf_counter()
    w.step(1)
    t_acc += time.perf_counter() - t0
    if s % every == every - 1:
        buf = C.create_string_buffer(64)
        ms, launches, nbytes = C.c_float(), C.c_int(), C.c_double()
        L.b2hip_get_kernel_timing(dev, buf, 64, C.byref(ms), C.byref(launches), C.byref(nbytes))
        ctr = b2hip.Counters()
        L.b2hip_get_counters(dev, C.byref(ctr))
        stamps = np.zeros(16, np.int32)
        L.b2hip_debug_read(dev, 11, 0, 16, stamps.ctypes.data)
        print("step %d: %.3f ms/step | %s %.1f us x%d (%.0f GB/s alg) | rows %d cut %d blocks %d maxrows %d partitions %d blocksteps %d colors %d posIters %d | stamps(10ns) %s" % (
            s + 1, 1e3 * t_acc / every, buf.value.decode(), 1e3 * ms.value, launches.value, nbytes.value / max(ms.value, 1e-9) / 1e6,
            ctr.large_island_contacts, ctr.cut_constraints, ctr.blocks, ctr.block_max_rows, ctr.partitions, ctr.block_solver_steps, ctr.colors,
            ctr.pos_iterations_large, stamps[8:14].tolist()), flush=True)
        t_acc = 0.0
# partition / colouring dump for offline analysis
ctr = b2hip.Counters(); L.b2hip_get_counters(dev, C.byref(ctr))
nb = w.body_count
act = np.zeros(nb, np.uint64); L.b2hip_debug_read(dev, 16, 0, nb, act.ctypes.data)
blk = np.zeros(nb, np.int32); L.b2hip_debug_read(dev, 17, 0, nb, blk.ctypes.data)
cc = np.zeros(65, np.int32); L.b2hip_debug_read(dev, 12, 0, 65, cc.ctypes.data)
nrows = ctr.large_island_contacts
ref = np.zeros((max(nrows, 1), 4), np.int32); L.b2hip_debug_read(dev, 18, 0, nrows, ref.ctypes.data)
rcol = np.zeros(max(nrows, 1), np.int32); L.b2hip_debug_read(dev, 19, 0, nrows, rcol.ctypes.data)
rstart = np.zeros(ctr.blocks + 1, np.int32); L.b2hip_debug_read(dev, 20, 0, ctr.blocks + 1, rstart.ctypes.data)
cutdeg = np.array([bin(int(x)).count("1") for x in act])
print("colour census", {i: int(c) for i, c in enumerate(cc) if c})
print("cut degree histogram of the bodies", np.bincount(cutdeg).tolist())
os.makedirs(os.path.join(ROOT, "gpurun_out", "blocks"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "blocks", "dump_%d.npz" % rows), bodies=w.bodies(), act=act, blk=blk, colorCount=cc, ref=ref, rowColor=rcol, rowStart=rstart)
b = w.bodies()
print("finite", bool(np.isfinite(b).all()), "min y", float(b[1:, 1].min()))
