"""bench.py's small-island sample on its own (100 000 piles of 5 boxes, 40 settle steps, 10 steps) for the rocprofv3 PMC
passes of tools/gpu_profile_round.sh: the last 10 dispatches of k_solve_small are the measured state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
w = amd.world(H.PILES, 100000, 5, seed=3, flags=H.F_CONTINUOUS | H.F_SLEEP | H.F_WARM)
w.step(50)
print(w.body_count, w.contact_count)
w.close()
