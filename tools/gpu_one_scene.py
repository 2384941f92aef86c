"""One harness scene for N steps (for rocprofv3 + tools/trace_steady.py): gpu_one_scene.py <scene id> <p0> <p1> <steps> [ccd]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
scene, p0, p1, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
fl = (H.F_CONTINUOUS if len(sys.argv) > 5 else 0) | H.F_SLEEP | H.F_WARM
w = amd.world(scene, p0, p1, flags=fl)
w.step(steps)
print(w.body_count, w.contact_count, {k: round(v, 3) for k, v in w.profile().items() if v})
w.close()
