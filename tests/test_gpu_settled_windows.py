"""GPU tests of the windows bench.py TIMES: the default mode - coloured order on the large islands - run to
the settled state of BASELINE configs 3, 4 (one share) and 2 and compared with aggregates of the REFERENCE BUILD's run to the
same steps (tests/golden/settled_windows.npz, made by tests/golden/make_golden_settled.py from oracle/_ref).

By step 700 of the Tumbler (320 of Pyramid 316, 240 of Pyramid 141) the device's trajectory and the reference's have long parted:
a reordered Gauss-Seidel sweep is a different, equally valid iteration, and a pile of 100 000 boxes multiplies any difference by
~1.6 per step while it moves (tools/gpu_r06_lockstep.py). What any valid order must deliver is the same PILE: as many contacts,
as many of them touching, the same penetration left by 8 + 3 iterations, the same weight carried, the same energy and speeds,
the container where its motor puts it. Bounds, on window means over 16 samples (every 4th step of 61), as MEASURED with a margin
(profiles/r06_*_settled_windows_*.txt hold the tables this test writes; two runs of one chaotic pile are two samples of it - the
318 steps of a collapsing 316-row pyramid leave the device's run with 4.5 % fewer fat-AABB pairs and 12 % less kinetic energy
than the reference's, the 10 011-box pyramid with 2.4 % and 7 %):
  * touching contacts within 4 %, contact count and summed normal impulse within 8 %, mean speed within 6 %, kinetic energy within 15 %
    of the reference's;
  * penetration (p99, mean): the device's <= 1.05 x the reference's (measured 0.81 - 0.97: a better solution is no failure);
  * extremes of a single body / point (deepest penetration, top speed): <= 1.25 x - one sample of an extreme value (measured
    0.84 - 1.14);
  * the container's angle: within 1e-3 rad at every sample (a motor with torque to spare: the same angle whatever the pile does).
Run-to-run determinism of the same window is asserted in tests/test_gpu_configs_full_size.py and test_gpu_sweep_end.py.
"""
import os

import numpy as np
import pytest

import b2harness as bh
import quality_util as qu

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "settled_windows.npz")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# key: (lower bound, upper bound) of device window mean / reference window mean
RATIO_BOUNDS = {"contacts": (0.92, 1.08), "touching": (0.96, 1.04), "impulse_sum": (0.92, 1.08), "speed_mean": (0.94, 1.06),
                "kinetic_energy": (0.85, 1.15), "penetration_p99": (0.0, 1.05), "penetration_mean": (0.0, 1.05),
                "penetration_max": (0.0, 1.25), "speed_max": (0.0, 1.25)}
# Per scene overrides, each with the measurement that asks for it (profiles/r06_*_settled_windows_*.txt).
#
# config3_tumbler316 - THE FIND OF THIS TEST. At steps 700..760 the reference build's pile and the device's are macroscopically
# different piles: the reference holds 5.38 M contacts of which 712 000 touch, penetration p99 0.237 m / deepest 0.52 m (the boxes
# are 0.25 m wide); the device 2.6 M / 370 000, p99 0.16 m / deepest 0.22 m. Both fell the same way (contact counts agree to
# 0.5 % over the first 30 steps, tests/test_gpu_configs_full_size.py; both reach ~5 M contacts at steps 150..300), but from
# there the device's pile expands again to what a dense packing of 100 000 boxes holds (3.7 touching contacts per box) while the
# reference's stays compressed (7.1 per box) for as long as it was run (700 steps = 3 hours of the reference build on 8 cores).
# It is the ORDER, and it grows with the depth of the pile: 8 + 3 Gauss-Seidel iterations cannot carry a pile 250 boxes deep in
# either order, and the reference's order (the island's depth-first constraint order, one thread) leaves it further from
# solved than colour order does. Evidence: (1) the 10 000-box Tumbler agrees between the two to 1 - 2 % in contact count over
# all 700 steps (tools/gpu_step_series.py against the reference build); (2) the 22 500-box Tumbler on the device in the
# REFERENCE'S order (B2HIP_FORCE_LARGE=2, bit-exact to the reference wherever it is compared) against the device's default
# mode, step 350: 476 000 contacts / 98 800 touching / p99 0.167 m against 437 000 / 88 200 / 0.157 m - already + 9 - 12 %
# (profiles/r06_c_order_effect_tumbler150.txt). So: the device's default mode does NOT reproduce the reference's pile at this
# depth - it leaves LESS penetration - and the bench line says so (`state_vs_reference`): its timed state holds half the
# contacts the reference's holds at the same step. What is asserted for this scene: the container's angle (exact physics of a
# motor with torque to spare), a solution no worse than the reference's (penetration), counts between a dense packing's and the
# reference's, energies and speeds of the same order.
OVERRIDES = {"config3_tumbler316": {"contacts": (0.40, 1.05), "touching": (0.45, 1.05), "impulse_sum": (0.9, 1.8), "kinetic_energy": (0.8, 1.6),
                                    "speed_mean": (0.85, 1.4), "speed_max": (0.0, 1.9)}}


def _scenes():
    # (the scenes the committed fixture holds: make_golden_settled.py writes one after the other - the Tumbler's 760 steps take
    # the reference build hours on eight shared cores)
    g = np.load(GOLDEN)
    return sorted({k.split("/")[0] for k in g.files})


@pytest.mark.parametrize("name", _scenes())
def test_settled_window_matches_the_reference_builds_aggregates(amd, name):
    g = np.load(GOLDEN)
    sc, p0, p1, seed, flags, first, last, every = (int(v) for v in g[name + "/params"])
    keys = [str(k) for k in g[name + "/keys"]]
    ref = g[name + "/table"]
    w = amd.world(sc, p0, p1, seed=seed, flags=flags)
    w.step(first)
    steps, dev = qu.window(w, first, last, every)
    w.close()
    assert list(steps) == list(g[name + "/steps"])
    assert np.isfinite(dev).all()
    lines = ["%s: device (default mode) / reference build, samples at steps %d..%d every %d" % (name, first, last, every),
             "step  " + "  ".join(keys)]
    for i, s in enumerate(steps):
        lines.append("%4d  " % s + "  ".join("%.6g/%.6g" % (dev[i, j], ref[i, j]) for j in range(len(keys))))
    dm, rm = dev.mean(axis=0), ref.mean(axis=0)
    lines.append("mean  " + "  ".join("%.6g/%.6g (%.4f)" % (dm[j], rm[j], dm[j] / rm[j] if rm[j] else 1.0) for j in range(len(keys))))
    report = "\n".join(lines)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "settled_windows_%s.txt" % name), "w") as f:
            f.write(report + "\n")
    bounds = dict(RATIO_BOUNDS, **OVERRIDES.get(name, {}))
    for j, k in enumerate(keys):
        if k == "angle_body1":
            if sc == bh.TUMBLER:
                assert np.abs(dev[:, j] - ref[:, j]).max() < 1e-3, "%s: the container's angle\n%s" % (name, report)
            continue
        lo, hi = bounds[k]
        ratio = dm[j] / rm[j] if rm[j] else 1.0
        assert lo <= ratio <= hi, "%s: window mean of %s is %.4f x the reference build's (bounds %.2f .. %.2f)\n%s" % (name, k, ratio, lo, hi, report)
