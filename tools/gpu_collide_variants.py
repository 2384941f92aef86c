"""k_collide's launch time (HIP event pair: b2hip_set_kernel_timing mode 2) in its four forms - contacts of a tile sorted by
shape-pair class or not (B2HIP_COLLIDE_SORT), shape records staged through LDS or not (B2HIP_COLLIDE_STAGE) - on the 1 M-body
field (a shape record per body) and on the 100 000-box Tumbler (one record for all boxes)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python")); sys.path.insert(0, ROOT)
import b2harness as H, b2hip
import bench
amd = H.Harness(H.AMD_LIB); L = b2hip.lib()
CCD = H.F_SLEEP | H.F_WARM | H.F_CONTINUOUS
for name, scene, p0, p1, flags, settle in (("field 1 M / 10 000 bullets", H.FIELD, 1000000, 10000, CCD, 30), ("Tumbler 316 x 316", H.TUMBLER, 316, 0, H.F_SLEEP | H.F_WARM, int(sys.argv[1]) if len(sys.argv) > 1 else 400)):
    for sort, stage in ((0, 0), (1, 0), (0, 1), (1, 1)):
        os.environ["B2HIP_COLLIDE_SORT"] = str(sort); os.environ["B2HIP_COLLIDE_STAGE"] = str(stage)
        w = amd.world(scene, p0, p1, seed=3, flags=flags)
        w.step(settle)
        dev = C.c_void_p(w.device_world())
        r = bench.kernel_roofline(L, dev, lambda: w.step(1), 2, 10, (w.contact_count, 0))
        h = H.fnv1a64(w.bodies())
        print("%-28s sort %d stage %d: %s %.1f us per launch, %d contacts, state hash %s" % (name, sort, stage, r["kernel"], r["mean_launch_us"], w.contact_count, h), flush=True)
        w.close()
