"""Golden traces of BASELINE.json's configs at FULL size, generated from the REAL reference (oracle/_ref/libb2ref_harness.so,
the reference's own sources compiled where they lie). Run in the build container (minutes):

    python tests/golden/make_golden_configs.py

Output (committed, small): config_scale.npz - per step the contact count (b2World::GetContactCount), the number of awake
bodies and the FNV-1a hash of every body's full state row (position, angle, velocities, awake, type: 8 floats), plus the
scene parameters that produced them. Fixtures are data; no reference source text is stored.

  config5_field1m     1 000 000 bodies (circles / boxes / n-gons), 10 000 bullets, continuous physics ON, seed 3 - the world
                      bench.py's `extra_configs` times on one GPU. Every island of it lies in the reference-order tier, so
                      the device must reproduce these hashes bit for bit in its DEFAULT mode.
  config4_pyramid316  one 316-row pyramid (50 086 boxes: one GPU's share of config 4): contact counts while it comes down.
  config3_tumbler316  99 856 boxes in the revolving container, continuous physics off: contact counts of the first steps.
  config4_4pyramids316  four such pyramids in one world (200 344 boxes): config 4 as BASELINE.json states it.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import b2harness as bh  # noqa: E402

CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM
# name, scene, p0, p1, seed, flags, steps
SCENES = [
    ("config5_field1m", bh.FIELD, 1000000, 10000, 3, CCD, 12),
    ("config4_pyramid316", bh.PYRAMID, 316, 1, 3, CCD, 60),
    ("config3_tumbler316", bh.TUMBLER, 316, 0, 3, bh.F_SLEEP | bh.F_WARM, 30),
    # config 4 as stated: four disjoint 316-row pyramids in ONE world (what the 4 ranks of tests/test_gpu_spatial.py share)
    ("config4_4pyramids316", bh.PYRAMID, 316, 4, 3, CCD, 40),
]


def main():
    ref = bh.Harness(bh.REF_LIB)
    out = {}
    path = os.path.join(HERE, "config_scale.npz")
    if os.path.exists(path) and "--all" not in sys.argv:
        old = np.load(path)
        out = {k: old[k] for k in old.files}  # (scenes already in the file are kept: `--all` regenerates everything)
    for name, sc, p0, p1, seed, flags, steps in SCENES:
        if name + "/hashes" in out:
            continue
        t0 = time.time()
        w = ref.world(sc, p0, p1, seed=seed, flags=flags, threads=8)
        counts = np.zeros(steps, np.int32)
        awake = np.zeros(steps, np.int32)
        hashes = []
        for s in range(steps):
            w.step(1)
            b = w.bodies()
            counts[s] = w.contact_count
            awake[s] = int((b[:, 6] != 0).sum())
            hashes.append(bh.fnv1a64(b))
        out[name + "/params"] = np.array([sc, p0, p1, seed, steps, flags], np.int64)
        out[name + "/contact_counts"] = counts
        out[name + "/awake"] = awake
        out[name + "/hashes"] = np.array(hashes)
        out[name + "/bodies"] = np.int64(w.body_count)
        print(name, w.body_count, counts.tolist(), hashes[-1], "%.1f s" % (time.time() - t0), flush=True)
        w.close()
    np.savez_compressed(os.path.join(HERE, "config_scale.npz"), **out)


if __name__ == "__main__":
    main()
