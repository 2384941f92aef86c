"""Aggregates of one harness scene every <every> steps (tests/quality_util.py: contacts, touching contacts, penetration, energy,
speeds): what the pile looks like, whatever order solved it. usage: gpu_aggregates_series.py <lib: amd|ref|oracle> <scene> <p0> <p1> <steps> <every> [ccd]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import b2harness as H, quality_util as qu
lib = {"amd": H.AMD_LIB, "ref": H.REF_LIB, "oracle": H.ORACLE_LIB}[sys.argv[1]]
scene, p0, p1, steps, every = (int(a) for a in sys.argv[2:7])
fl = (H.F_CONTINUOUS if len(sys.argv) > 7 else 0) | H.F_SLEEP | H.F_WARM
w = H.Harness(lib).world(scene, p0, p1, flags=fl, threads=8) if sys.argv[1] == "ref" else H.Harness(lib).world(scene, p0, p1, flags=fl)
mass = w.mass()
done = 0
t0 = time.time()
while done < steps:
    w.step(every); done += every
    a = qu.aggregates(w, mass)
    print("step %4d (%.0f s): contacts %d touching %d pen max %.3f p99 %.3f mean %.4f impulse %.4g KE %.4g speed max %.2f mean %.3f" % (
        done, time.time() - t0, a["contacts"], a["touching"], a["penetration_max"], a["penetration_p99"], a["penetration_mean"], a["impulse_sum"], a["kinetic_energy"], a["speed_max"], a["speed_mean"]), flush=True)
w.close()
