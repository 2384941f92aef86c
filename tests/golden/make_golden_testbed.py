"""Golden traces of the reference's own large Testbed scenes (Testbed/Tests/ManyBodies.h: ManyBodies1 .. 5, 10 000 - 50 000
bodies), generated from the REAL reference (oracle/_ref/libtestbed_ref.so = tests/testbed/scenes_main.cpp compiled against
/root/reference's Box2D and its unmodified scene headers). The C oracle's brute-force broad-phase cannot follow at these
sizes, so the GPU is pinned against these traces directly. Run in the build container:

    python tests/golden/make_golden_testbed.py

Output: testbed_big.npz - per scene and step the six summary figures (bodies, contacts, sum |x| + |y|, top speed^2, finite,
awake) and the FNV-1a hash of every body's full state. Fixtures are data; no reference source text is stored."""
import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
STEPS = 40


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtestbed_ref.so"), mode=C.RTLD_LOCAL)
    L.testbed_trace.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    L.testbed_trace_hash.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    out = {}
    for name in ("ManyBodies1", "ManyBodies2", "ManyBodies3", "ManyBodies4", "ManyBodies5"):
        t0 = time.time()
        summ = np.zeros((STEPS, 6))
        hashes = np.zeros(STEPS, np.uint64)
        L.testbed_trace(name.encode(), STEPS, summ.ctypes.data)
        L.testbed_trace_hash(name.encode(), STEPS, hashes.ctypes.data)
        out[name + "/summaries"] = summ
        out[name + "/hashes"] = hashes
        print(name, summ[-1].tolist(), hex(int(hashes[-1])), "%.1f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "testbed_big.npz"), **out)


if __name__ == "__main__":
    main()
