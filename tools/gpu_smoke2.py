"""GPU bring-up 2: exact-order large path must be bit-exact with the reference (dev script)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["B2HIP_FORCE_LARGE"] = "2"
import importlib.util
spec = importlib.util.spec_from_file_location("s1", os.path.join(ROOT, "tools", "gpu_smoke1.py"))
src = open(os.path.join(ROOT, "tools", "gpu_smoke1.py")).read().split("w = hello(); rw")[0]
exec(src)
w = hello(); rw = ref.world(bh.HELLO); compare("hello-exactL", w, rw, 60, 6, 2); w.close()
for rows in (3, 10, 20, 40):
    w = pyramid(rows); rw = ref.world(bh.PYRAMID, rows, 1); compare("pyr%d-exactL" % rows, w, rw, 90, 8, 3); w.close()
