"""Golden fixtures for the continuous-collision (TOI) part of the hot path, generated from the REAL
reference (oracle/_ref/libb2ref_harness.so). Run in the build container:

    python tests/golden/make_golden_toi.py

Outputs (committed, small):
  toi_scenes.npz    scenes stepped with continuous physics ON: final body states, per-step contact
                    counts and pose hashes, final contact ids / flags / manifolds
  toi_scale.npz     at-scale traces (per-step contact counts and full-state hashes only)
  toi_vectors.npz   b2Distance and b2TimeOfImpact inputs (vertex proxies, transforms / sweeps) and the
                    reference's outputs
Fixtures are data (inputs and expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import b2harness as bh  # noqa: E402

CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM

# name, scene, p0, p1, f0, f1, seed, steps
SCENES = [
    ("ccd_helloworld", bh.HELLO, 0, 0, 0.0, 0.0, 1, 60),
    ("ccd_bullets", bh.BULLETS, 60, 6, 0.0, 0.0, 2, 180),
    ("ccd_field", bh.FIELD, 400, 60, 40.0, 3.0, 5, 150),
    ("ccd_pyramid12", bh.PYRAMID, 12, 1, 0.0, 0.0, 1, 120),
    ("ccd_rain", bh.RAIN, 150, 0, 0.0, 0.0, 7, 150),
    ("ccd_tumbler6", bh.TUMBLER, 6, 0, 0.0, 0.0, 1, 200),
]

# at-scale traces (hashes and contact counts only): thousands of bullets among free bodies, the workload of the
# per-component TOI path
SCALE_SCENES = [
    ("ccd_field30k", bh.FIELD, 30000, 3000, 0.0, 0.0, 17, 30),
]

POLY_RADIUS = 0.01


def proxy(ref, rng, kind):
    """(verts[n,2], radius, centroid)"""
    if kind == 0:  # circle: one vertex
        return np.array([[rng.uniform(-.2, .2), rng.uniform(-.2, .2)]], np.float32), float(np.float32(rng.uniform(.1, .8))), (0.0, 0.0)
    if kind == 1:  # edge: two vertices
        return np.array([[-rng.uniform(.5, 3), rng.uniform(-.2, .2)], [rng.uniform(.5, 3), rng.uniform(-.2, .2)]], np.float32), POLY_RADIUS, (0.0, 0.0)
    n = rng.integers(3, 9)
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = rng.uniform(0.2, 1.2)
    pts = [(r * np.cos(a) * rng.uniform(0.7, 1), r * np.sin(a) * rng.uniform(0.7, 1)) for a in ang]
    o = ref.polygon(pts)
    cnt = int(o[0])
    return o[1:1 + 2 * cnt].reshape(cnt, 2).copy(), POLY_RADIUS, (float(o[33]), float(o[34]))


def main():
    ref = bh.Harness(bh.REF_LIB)
    out = {}
    for name, sc, p0, p1, f0, f1, seed, steps in SCENES:
        w = ref.world(sc, p0, p1, f0, f1, seed, flags=CCD)
        counts = np.zeros(steps, np.int32)
        hashes = []
        for s in range(steps):
            w.step(1)
            counts[s] = w.contact_count
            hashes.append(bh.fnv1a64(w.bodies()[:, :3]))
        ids, flags, man = w.contacts()
        out[name + "/params"] = np.array([sc, p0, p1, seed, steps], np.int64)
        out[name + "/fparams"] = np.array([f0, f1], np.float32)
        out[name + "/bodies"] = w.bodies()
        out[name + "/contact_counts"] = counts
        out[name + "/hashes"] = np.array(hashes)
        out[name + "/contact_ids"] = ids
        out[name + "/contact_flags"] = flags
        out[name + "/contact_manifolds"] = man
        print(name, w.body_count, counts[-1], hashes[-1], "solveTOI ms", w.profile()["solveTOI"])
        w.close()
    np.savez_compressed(os.path.join(HERE, "toi_scenes.npz"), **out)

    big = {}
    for name, sc, p0, p1, f0, f1, seed, steps in SCALE_SCENES:
        w = ref.world(sc, p0, p1, f0, f1, seed, flags=CCD, threads=8)
        counts = np.zeros(steps, np.int32)
        hashes = []
        for s in range(steps):
            w.step(1)
            counts[s] = w.contact_count
            hashes.append(bh.fnv1a64(w.bodies()))
        big[name + "/params"] = np.array([sc, p0, p1, seed, steps], np.int64)
        big[name + "/fparams"] = np.array([f0, f1], np.float32)
        big[name + "/contact_counts"] = counts
        big[name + "/hashes"] = np.array(hashes)
        print(name, w.body_count, counts[-1], hashes[-1], "solveTOI ms", w.profile()["solveTOI"])
        w.close()
    np.savez_compressed(os.path.join(HERE, "toi_scale.npz"), **big)

    rng = np.random.default_rng(20240917)
    # ---- b2Distance ---------------------------------------------------------------------------
    N = 1200
    vA = np.zeros((N, 16), np.float32); vB = np.zeros((N, 16), np.float32)
    nA = np.zeros(N, np.int32); nB = np.zeros(N, np.int32)
    rA = np.zeros(N, np.float32); rB = np.zeros(N, np.float32)
    xfA = np.zeros((N, 3), np.float32); xfB = np.zeros((N, 3), np.float32)
    useR = np.zeros(N, np.int32)
    dist = np.zeros((N, 6), np.float32)
    for i in range(N):
        a, ra, _ = proxy(ref, rng, [2, 2, 0, 1][i % 4])
        b, rb, _ = proxy(ref, rng, [2, 0, 0, 2][(i // 4) % 4])
        nA[i], nB[i] = len(a), len(b)
        vA[i, :a.size] = a.reshape(-1); vB[i, :b.size] = b.reshape(-1)
        rA[i], rB[i] = ra, rb
        xfA[i] = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-7, 7)]
        spread = [0.5, 2.0, 5.0][i % 3]
        xfB[i] = [xfA[i, 0] + rng.uniform(-spread, spread), xfA[i, 1] + rng.uniform(-spread, spread), rng.uniform(-7, 7)]
        useR[i] = (i // 2) % 2
        dist[i] = ref.distance(a, ra, xfA[i], b, rb, xfB[i], bool(useR[i]))
    # ---- b2TimeOfImpact -----------------------------------------------------------------------
    M = 2000
    tvA = np.zeros((M, 16), np.float32); tvB = np.zeros((M, 16), np.float32)
    tnA = np.zeros(M, np.int32); tnB = np.zeros(M, np.int32)
    trA = np.zeros(M, np.float32); trB = np.zeros(M, np.float32)
    swA = np.zeros((M, 9), np.float32); swB = np.zeros((M, 9), np.float32)
    toi = np.zeros((M, 2), np.float32)
    for i in range(M):
        a, ra, ca = proxy(ref, rng, [2, 2, 0, 1, 2][i % 5])
        b, rb, cb = proxy(ref, rng, [2, 0, 0, 2, 2][(i // 5) % 5])
        tnA[i], tnB[i] = len(a), len(b)
        tvA[i, :a.size] = a.reshape(-1); tvB[i, :b.size] = b.reshape(-1)
        trA[i], trB[i] = ra, rb
        # A mostly static / slow, B flies towards (or past) A; angles up to a few turns to exercise Normalize
        c0a = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1)])
        da = rng.uniform(-0.3, 0.3, 2) * (i % 3 == 0)
        a0a = rng.uniform(-20, 20)
        ang = rng.uniform(0, 2 * np.pi)
        start = rng.uniform(1.5, 8.0)
        c0b = c0a + start * np.array([np.cos(ang), np.sin(ang)])
        aim = c0a + rng.uniform(-1.0, 1.0, 2)
        travel = rng.uniform(0.3, 2.5)
        cb1 = c0b + travel * (aim - c0b)
        a0b = rng.uniform(-20, 20)
        swA[i] = [ca[0], ca[1], c0a[0], c0a[1], c0a[0] + da[0], c0a[1] + da[1], a0a, a0a + rng.uniform(-1, 1) * (i % 2), 0.0]
        swB[i] = [cb[0], cb[1], c0b[0], c0b[1], cb1[0], cb1[1], a0b, a0b + rng.uniform(-3, 3), 0.0]
        toi[i] = ref.toi(a, ra, swA[i], b, rb, swB[i], 1.0)
    states = toi[:, 0].astype(int)
    print("distance vectors", N, "toi vectors", M, "states", np.bincount(states, minlength=5).tolist())
    np.savez_compressed(os.path.join(HERE, "toi_vectors.npz"), d_vertsA=vA, d_vertsB=vB, d_countA=nA, d_countB=nB,
                        d_radiusA=rA, d_radiusB=rB, d_xfA=xfA, d_xfB=xfB, d_useRadii=useR, d_out=dist,
                        t_vertsA=tvA, t_vertsB=tvB, t_countA=tnA, t_countB=tnB, t_radiusA=trA, t_radiusB=trB,
                        t_sweepA=swA, t_sweepB=swB, t_out=toi)


if __name__ == "__main__":
    main()
