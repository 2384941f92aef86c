"""How deep a sweep of the block solver is: per block of the current partition the number of interior and of cut colours in
use (every interior colour is a workgroup barrier per sweep, every cut colour a hand-over through memory), and the colour
census of the whole island.   usage: python tools/gpu_block_colours.py [rows] [settle steps]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh, b2hip
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 316
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 340
amd = bh.Harness(bh.AMD_LIB); L = b2hip.lib()
L.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(bh.PYRAMID, rows, 1, seed=3, flags=bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM)
dev = C.c_void_p(w.device_world())
w.step(settle)
c = b2hip.Counters(); L.b2hip_get_counters(dev, C.byref(c))
nb = c.blocks
start = np.zeros(nb + 2, np.int32); L.b2hip_debug_read(dev, 20, 0, nb + 2, start.ctypes.data)
nrows = int(start[nb])
col = np.zeros(max(nrows, 1), np.int32); L.b2hip_debug_read(dev, 19, 0, nrows, col.ctypes.data)
col &= 63
cc = np.zeros(65, np.int32); L.b2hip_debug_read(dev, 12, 0, 65, cc.ctypes.data)
inter = [len(set(col[start[b]:start[b + 1]][col[start[b]:start[b + 1]] < 32].tolist())) for b in range(nb)]
cut = [len(set(col[start[b]:start[b + 1]][(col[start[b]:start[b + 1]] >= 32) & (col[start[b]:start[b + 1]] < 63)].tolist())) for b in range(nb)]
print("pyramid %d after %d steps: %d constraints, %d blocks of %d..%d rows, %d cut rows" % (rows, settle, c.large_island_contacts, nb, int(np.diff(start[:nb + 1]).min()), int(np.diff(start[:nb + 1]).max()), c.cut_constraints))
print("interior colours in use per block: min %d median %d max %d | cut colours per block: min %d median %d max %d" % (min(inter), int(np.median(inter)), max(inter), min(cut), int(np.median(cut)), max(cut)))
print("island census by colour (interior 0..31):", cc[:32].tolist())
print("                         (cut 32..62):", cc[32:63].tolist())
