"""Debug: where a reference Testbed scene parts between the device-backed and the oracle-backed drop-in layer."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]; steps = int(sys.argv[2])
if len(sys.argv) > 3: os.environ["B2HIP_FORCE_LARGE"] = sys.argv[3]
which = sys.argv[4].split(",") if len(sys.argv) > 4 else ["amd", "oracle"]
libs = {}
for k in which:
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtestbed_%s.so" % k))
    L.testbed_trace.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    L.testbed_states.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    libs[k] = L
tr = {}
for k, L in libs.items():
    a = np.zeros((steps, 6)); r = L.testbed_trace(name.encode(), steps, a.ctypes.data); tr[k] = (r, a)
    print(k, "result", r, "last", a[-1], flush=True)
if len(which) == 2:
    A, B = tr[which[0]][1], tr[which[1]][1]
    d = np.nonzero((A != B).any(axis=1))[0]
    print(name, "first differing steps", d[:8])
    if len(d):
        i = int(d[0])
        st = {}
        for k, L in libs.items():
            rows = np.zeros((4096, 8)); n = L.testbed_states(name.encode(), i + 1, rows.ctypes.data, 4096); st[k] = rows[:n]
        X, Y = st[which[0]], st[which[1]]
        bad = np.nonzero((X != Y).any(axis=1))[0] if X.shape == Y.shape else []
        print(" after", i + 1, "steps: bodies", X.shape[0], Y.shape[0], "differing rows", bad[:10])
        np.set_printoptions(linewidth=200, precision=7)
        for j in bad[:6]:
            print("  row", j, "\n   ", which[0], X[j], "\n   ", which[1], Y[j])
