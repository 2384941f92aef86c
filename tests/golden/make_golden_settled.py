"""Golden AGGREGATES of the windows bench.py times, from the REAL reference (oracle/_ref/libb2ref_harness.so, the reference's own
sources compiled where they lie, 8 threads). Run in the build container (about 8 minutes):

    python tests/golden/make_golden_settled.py

Output (committed, small): settled_windows.npz - per scene the sampled steps and, per sample, the aggregates of
tests/quality_util.py (contact count, touching contacts, deepest / p99 / mean penetration, summed normal impulse, kinetic
energy, top and mean speed, the container's angle). Fixtures are data; no reference source text is stored.

  config3_tumbler316  99 856 boxes in the revolving container, continuous physics off (BASELINE configs[2], the N = 1 bench
                      line): steps 700 .. 760 every 4th - the bench settles 700 steps and times the steps behind them (the
                      splash of the start has died down by then: tools/gpu_step_series.py, profiles/r06_*_step_series.txt).
  config4_pyramid316  one 316-row pyramid, continuous physics on (one GPU's share of BASELINE configs[3]): steps 320 .. 380.
  config2_pyramid141  the 10 011-box pyramid, continuous physics on (BASELINE configs[1]): steps 240 .. 300.

The device's default mode sweeps large islands in coloured order: by step 700 its trajectory and the reference's are two
samples of the same chaotic pile. What must agree is what ANY valid order delivers - these aggregates, as window means
(tests/test_gpu_settled_windows.py states the bounds).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import b2harness as bh  # noqa: E402
import quality_util as qu  # noqa: E402

CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM
# name, scene, p0, p1, seed, flags, first, last, every
SCENES = [
    ("config2_pyramid141", bh.PYRAMID, 141, 1, 1, CCD, 240, 300, 4),
    ("config4_pyramid316", bh.PYRAMID, 316, 1, 1, CCD, 320, 380, 4),
    ("config3_tumbler316", bh.TUMBLER, 316, 0, 1, bh.F_SLEEP | bh.F_WARM, 700, 760, 4),
]


def main():
    ref = bh.Harness(bh.REF_LIB)
    # (--out <file>: write somewhere else - the Tumbler's 760 steps take the reference build over an hour on the build container's
    # eight cores and ~10 minutes on the GPU box's host, where oracle/_ref/libb2ref_harness.so travels with the tree; the scenes
    # already in settled_windows.npz are kept)
    path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(HERE, "settled_windows.npz")
    if "--out" in sys.argv and not os.path.exists(path) and os.path.exists(os.path.join(HERE, "settled_windows.npz")):
        import shutil
        shutil.copy(os.path.join(HERE, "settled_windows.npz"), path)
    out = {}
    if os.path.exists(path) and "--all" not in sys.argv:
        old = np.load(path)
        out = {k: old[k] for k in old.files}
    for name, sc, p0, p1, seed, flags, first, last, every in SCENES:
        if name + "/table" in out:
            continue
        t0 = time.time()
        w = ref.world(sc, p0, p1, seed=seed, flags=flags, threads=8)
        w.step(first)
        print(name, "settled %d steps in %.1f s" % (first, time.time() - t0), flush=True)
        steps, table = qu.window(w, first, last, every, on_sample=lambda s, a: print("  step", s, {k: round(v, 5) for k, v in a.items()}, flush=True))
        out[name + "/params"] = np.array([sc, p0, p1, seed, flags, first, last, every], np.int64)
        out[name + "/steps"] = steps
        out[name + "/table"] = table
        out[name + "/keys"] = np.array(qu.KEYS)
        print(name, "%.1f s" % (time.time() - t0), "window means:", dict(zip(qu.KEYS, table.mean(axis=0).round(5).tolist())), flush=True)
        w.close()
        np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
