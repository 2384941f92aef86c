"""Golden fixtures added in round 3, generated from the REAL reference (oracle/_ref/libb2ref_harness.so, i.e.
skitzoid/Box2D-MT compiled from /root/reference by oracle/Makefile). Run in the build container:

    python tests/golden/make_golden_r3.py

Output (committed, small): scenes_r3.npz, same per-scene layout as scenes.npz / toi_scenes.npz (final body states, masses,
per-step contact counts and pose hashes, final contact ids / flags / manifolds; `flags` = the world flags of the run):
  chains        b2ChainShape scene (loop, open chains with and without ghost vertices, a chain on a kinematic body),
                continuous physics off
  ccd_chains    the same scene with continuous physics on (bullets against chain children)
Fixtures are data (inputs and expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import b2harness as bh  # noqa: E402

CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM

# name, scene, p0, p1, f0, f1, seed, steps, world flags
SCENES = [
    ("chains", bh.CHAINS, 90, 0, 0.0, 0.0, 4, 300, bh.DEFAULT_FLAGS),
    ("ccd_chains", bh.CHAINS, 90, 0, 0.0, 0.0, 6, 300, CCD),
]


def main():
    ref = bh.Harness(bh.REF_LIB)
    out = {}
    for name, sc, p0, p1, f0, f1, seed, steps, flags in SCENES:
        w = ref.world(sc, p0, p1, f0, f1, seed, flags=flags)
        counts = np.zeros(steps, np.int32)
        hashes = []
        for s in range(steps):
            w.step(1)
            counts[s] = w.contact_count
            hashes.append(bh.fnv1a64(w.bodies()[:, :3]))
        ids, cflags, man = w.contacts()
        out[name + "/params"] = np.array([sc, p0, p1, seed, steps], np.int64)
        out[name + "/fparams"] = np.array([f0, f1], np.float32)
        out[name + "/flags"] = np.array([flags], np.int64)
        out[name + "/bodies"] = w.bodies()
        out[name + "/mass"] = w.mass()
        out[name + "/contact_counts"] = counts
        out[name + "/hashes"] = np.array(hashes)
        out[name + "/contact_ids"] = ids
        out[name + "/contact_flags"] = cflags
        out[name + "/contact_manifolds"] = man
        print(name, w.body_count, counts[-1], counts.max(), hashes[-1])
        w.close()
    np.savez_compressed(os.path.join(HERE, "scenes_r3.npz"), **out)


if __name__ == "__main__":
    main()
