"""A/B of an environment variable the HIP RUNTIME reads at start-up (so: one fresh process per value, unlike tools/gpu_ab.py,
whose switches the library reads at world creation): wall-clock ms per step of one harness scene in its settled window.
usage: python tools/gpu_env_ab.py VAR=a,b[,c] <scene id> <p0> <p1> <settle> <timed> [ccd]
e.g.   python tools/gpu_env_ab.py HIP_FORCE_DEV_KERNARG=0,1 2 316 0 405 40      (the Tumbler of the bench line)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time
sys.path.insert(0, os.path.join(%r, "tests"))
import b2harness as H
amd = H.Harness(H.AMD_LIB)
scene, p0, p1, settle, timed = [int(x) for x in sys.argv[1:6]]
fl = (H.F_CONTINUOUS if len(sys.argv) > 6 else 0) | H.F_SLEEP | H.F_WARM
w = amd.world(scene, p0, p1, flags=fl)
w.step(settle)
ms = []
for _ in range(timed):
    t = time.perf_counter(); w.step(1); ms.append(1000.0 * (time.perf_counter() - t))
ms.sort()
print("%%.4f mean  %%.4f p50  %%.4f p99  (%%d bodies, %%d contacts, hash %%s)" %% (sum(ms) / len(ms), ms[len(ms) // 2], ms[min(len(ms) - 1, int(0.99 * len(ms)))], w.body_count, w.contact_count, H.fnv1a64(w.bodies())))
w.close()
""" % ROOT


def main():
    var, values = sys.argv[1].split("=", 1)
    args = sys.argv[2:]
    for rep in range(2):
        for v in values.split(","):
            env = dict(os.environ)
            if v == "unset":
                env.pop(var, None)
            else:
                env[var] = v
            out = subprocess.run([sys.executable, "-c", CHILD] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout.strip().split("\n")[-1]
            print("%s=%-6s run %d: %s" % (var, v, rep, out), flush=True)


if __name__ == "__main__":
    main()
