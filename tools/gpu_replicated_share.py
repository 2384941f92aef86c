"""What a rank of a sharded world pays for the phases it runs replicated: ONE unsharded world of N pyramids (141 rows each, one
ground) on one GPU, ms per step and the device profile by phase, N = 1, 2, 4, 8. A rank of an N-GPU run (bench.py --gpus N)
runs collide, the island build, the pair update and the TOI phase for all N pyramids and solves one of them.
usage: gpu_replicated_share.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 141
amd = H.Harness(H.AMD_LIB)
fl = H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM
for n in (1, 2, 4, 8):
    w = amd.world(H.PYRAMID, rows, n, flags=fl)
    w.step(240)
    w.reset_profile()
    t0 = time.perf_counter(); w.step(100); dt = (time.perf_counter() - t0) / 100
    p = w.profile()
    rep = p["collide"] + p["solveTraversal"] + p["broadphase"] + p["solveTOI"]
    print("%d pyramids (%d bodies, %d contacts): %.3f ms/step; collide %.3f, island build + colouring %.3f, solver %.3f, pair update %.3f, TOI %.3f -> replicated phases %.3f ms" % (
        n, w.body_count, w.contact_count, dt * 1e3, p["collide"], p["solveTraversal"], p["solveInit"] + p["solveVelocity"] + p["solvePosition"], p["broadphase"], p["solveTOI"], rep), flush=True)
    w.close()
