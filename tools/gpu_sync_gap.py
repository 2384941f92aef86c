"""What the host's census read-back in the middle of b2World::Step costs on the device's clock: time from the end of
k_block_census (the last kernel before the read-back) to the start of the first kernels the host launches after it.
usage: gpu_sync_gap.py [rows] [steps]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2hip
from test_gpu_onestep import build_pyramid
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 141
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
w = b2hip.World(continuous=True)
build_pyramid(w, rows)
for _ in range(150):
    w.step()
L = b2hip.lib()
L.b2hip_debug_gap_clocks.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
gaps = []
t0 = time.perf_counter()
for _ in range(steps):
    w.step()
    c = (ctypes.c_ulonglong * 4)()
    L.b2hip_debug_gap_clocks(w.p, c)
    gaps.append([(int(c[k]) - int(c[0])) / 100.0 for k in range(1, 4)])
dt = (time.perf_counter() - t0) / steps
g = np.array(gaps)
print("pyramid %d: %.1f us/step wall; from the end of k_block_census to the start of k_island_dfs / k_color_small / k_solve_blocks:" % (rows, dt * 1e6))
print("  mean us", np.round(g.mean(axis=0), 1), " p50", np.round(np.median(g, axis=0), 1), " p99", np.round(np.percentile(g, 99, axis=0), 1))
