"""How much of the coloured solver's one-step deviation is the visiting ORDER, and can a better order remove it? (CPU only.)

Builds the C oracle with -DB2O_ORDER_EXPERIMENT (oracle/b2o_step.c: ex_reorder) into tools/_order_exp/, steps config 2
(Pyramid 141 rows, continuous physics on) in the reference's order to the steps given by STEPS (default 245,300: the state
bench.py times), and there forks one child per candidate order (MODES, "k" or "k:D", see b2o_step.c) that takes ONE step in
that order; the parent takes the same step in the reference's order and prints every child's deviation from it, in the
metrics of tests/test_gpu_onestep.py. Usage: python tools/order_experiment.py        (about 40 s)"""
import os, sys, time, pickle, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2hip
import test_gpu_onestep as t
OUT = os.path.join(ROOT, "tools", "_order_exp")
os.makedirs(OUT, exist_ok=True)
O = os.path.join(ROOT, "oracle")
subprocess.check_call(["gcc", "-std=c99", "-O2", "-fPIC", "-ffp-contract=off", "-w", "-DB2O_ORDER_EXPERIMENT", "-shared", "-o", os.path.join(OUT, "liboexp.so")] +
                      [os.path.join(O, f) for f in ("b2o_step.c", "b2o_collide.c", "b2o_joint.c", "b2o_toi.c", "b2o_abi_shim.c")] + ["-I" + os.path.join(ROOT, "include"), "-lm"])
L = b2hip.load(os.path.join(OUT, "liboexp.so"), optional_ok=True)
rows = int(os.environ.get("ROWS", "141"))
ks = [int(x) for x in os.environ.get("STEPS", "245,300").split(",")]
modes = [x for x in os.environ.get("MODES", "1,2,3,4:64,4:256,5:64,6").split(",")]
a = b2hip.World(library=L, continuous=True)
t.build_pyramid(a, rows)
step = 0
t0 = time.time()

class Snap:
    def __init__(self, st, ct): self.st, self.ct = st, ct
    def body_states(self): return self.st
    def contacts(self): return self.ct

for k in ks:
    while step < k:
        a.step(); step += 1
    for m in modes:
        mode, D = (m.split(":") + ["64"])[:2]
        pid = os.fork()
        if pid == 0:
            os.environ["B2O_ORDER"] = mode; os.environ["B2O_ORDER_D"] = D
            a.step()
            pickle.dump((a.body_states(), a.contacts()), open(os.path.join(OUT, "out_%s.pkl" % m), "wb"))
            os._exit(0)
        os.waitpid(pid, 0)
    a.step(); step += 1
    for m in modes:
        st, ct = pickle.load(open(os.path.join(OUT, "out_%s.pkl" % m), "rb"))
        dev = t.one_step_deviation(Snap(st, ct), a)
        print("step %d mode %s: pos %.3g (|dp| max %.3g m p99 %.3g p50 %.3g) vel %.3g (|dv| max %.3g p99 %.3g) angle %.3g spin %.3g; contact diff %d  [%.0f s]" % (
            step, m, dev["pos"], dev["pos_m"], dev["pos_m_p99"], dev["pos_m_p50"], dev["vel"], dev["vel_mps"], dev["vel_mps_p99"], dev["angle"], dev["spin"], dev["contact_set_diff"], time.time() - t0), flush=True)
