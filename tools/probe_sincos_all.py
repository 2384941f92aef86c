"""Exhaustive check of the device sin/cos (box2d-mt_amd/csrc/b2d_math.h compiled for the CPU: tests/probe/host_probe.cpp)
against this machine's libm for ALL 2^32 float bit patterns (the reference calls sinf / cosf: b2Math.h:294-299).
Eight processes, a few minutes. Prints the number of mismatches (0 expected)."""
import ctypes as C, os, sys
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

def run(k):
    import probe_util as pu
    P = pu.build_probe()
    P.probe_sincos_vs_libm.restype = C.c_long
    n = 1 << 29
    lo = k * n
    return P.probe_sincos_vs_libm(C.c_uint(lo), C.c_uint(lo + n - 1), C.c_uint(1))

if __name__ == "__main__":
    import probe_util as pu
    pu.build_probe()
    with Pool(8) as p:
        bad = p.map(run, range(8))
    print("mismatches per eighth of the bit patterns:", bad, "total", sum(bad))
    sys.exit(1 if sum(bad) else 0)
