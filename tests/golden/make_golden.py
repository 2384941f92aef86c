"""Generates the golden fixtures from the REAL reference (oracle/_ref/libb2ref_harness.so, i.e.
skitzoid/Box2D-MT compiled from /root/reference by oracle/Makefile). Run in the build container:

    python tests/golden/make_golden.py

Outputs (committed, small):
  helloworld.txt        the 60 lines HelloWorld/HelloWorld.cpp prints (x y angle)
  scenes.npz            per scene: final body states, per-step contact counts and FNV pose hashes,
                        final contact ids / flags / manifolds
  collide_vectors.npz   inputs + reference manifolds for the five narrow-phase routines
  polygon_vectors.npz   b2PolygonShape::Set / ComputeMass inputs + outputs
  sincos_vectors.npz    b2Rot::Set inputs + outputs
Fixtures are data (inputs and expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import b2harness as bh  # noqa: E402
import probe_util as pu  # noqa: E402

# name, scene, p0, p1, f0, f1, seed, steps
SCENES = [
    ("helloworld", bh.HELLO, 0, 0, 0.0, 0.0, 1, 60),
    ("pyramid12", bh.PYRAMID, 12, 1, 0.0, 0.0, 1, 150),
    ("pyramid5x3", bh.PYRAMID, 5, 3, 0.0, 0.0, 1, 150),
    ("pyramid30", bh.PYRAMID, 30, 1, 0.0, 0.0, 1, 90),
    ("piles", bh.PILES, 40, 5, 0.0, 0.0, 7, 200),
    ("rain", bh.RAIN, 200, 0, 0.0, 0.0, 7, 200),
    ("circlestack", bh.CIRCLE_STACK, 8, 4, 0.0, 0.0, 1, 200),
    ("field", bh.FIELD, 1500, 0, 0.0, 0.0, 7, 100),
    ("tumbler6", bh.TUMBLER, 6, 0, 0.0, 0.0, 1, 300),
    ("tumbler20", bh.TUMBLER, 20, 0, 0.0, 0.0, 1, 150),
    ("sensors", bh.SENSORS, 40, 0, 0.0, 0.0, 5, 240),
    ("ropes", bh.ROPES, 80, 14, 0.0, 0.0, 9, 240),
    ("machines", bh.MACHINES, 120, 6, 0.0, 0.0, 3, 300),
    ("vehicles", bh.VEHICLES, 150, 5, 0.0, 0.0, 3, 300),
    ("pyramid141", bh.PYRAMID, 141, 1, 0.0, 0.0, 1, 30),
]


def main():
    ref = bh.Harness(bh.REF_LIB)
    out = {}
    for name, sc, p0, p1, f0, f1, seed, steps in SCENES:
        w = ref.world(sc, p0, p1, f0, f1, seed)
        counts = np.zeros(steps, np.int32)
        hashes = []
        lines = []
        for s in range(steps):
            w.step(1)
            counts[s] = w.contact_count
            b = w.bodies()
            hashes.append(bh.fnv1a64(b[:, :3]))
            if name == "helloworld":
                lines.append("%4.2f %4.2f %4.2f" % (b[1, 0], b[1, 1], b[1, 2]))
        ids, flags, man = w.contacts()
        out[name + "/params"] = np.array([sc, p0, p1, seed, steps], np.int64)
        out[name + "/fparams"] = np.array([f0, f1], np.float32)
        out[name + "/bodies"] = w.bodies()
        out[name + "/mass"] = w.mass()
        out[name + "/contact_counts"] = counts
        out[name + "/hashes"] = np.array(hashes)
        if name != "pyramid141":
            out[name + "/contact_ids"] = ids
            out[name + "/contact_flags"] = flags
            out[name + "/contact_manifolds"] = man
        else:
            out[name + "/bodies"] = w.bodies()[:, :3]  # keep the fixture small
            out[name + "/mass"] = w.mass()[:16]
        if name == "helloworld":
            open(os.path.join(HERE, "helloworld.txt"), "w").write("\n".join(lines) + "\n")
        print(name, w.body_count, counts[-1], hashes[-1])
        w.close()
    np.savez_compressed(os.path.join(HERE, "scenes.npz"), **out)

    # ---- narrow-phase vectors ---------------------------------------------------------------
    rng = np.random.default_rng(20240601)

    def rpoly():
        n = rng.integers(3, 9)
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        r = rng.uniform(0.3, 1.0)
        return [(r * np.cos(a) * rng.uniform(0.7, 1), r * np.sin(a) * rng.uniform(0.7, 1)) for a in ang]

    shapesA, shapesB, xfA, xfB, expect = [], [], [], [], []
    for i in range(1500):
        xa = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-7, 7)]
        xb = [xa[0] + rng.uniform(-1.6, 1.6), xa[1] + rng.uniform(-1.6, 1.6), rng.uniform(-7, 7)]
        k = i % 6
        if k == 0:
            a = ("box", rng.uniform(.2, 1), rng.uniform(.2, 1))
            b = ("box", rng.uniform(.2, 1), rng.uniform(.2, 1))
            r = ref.collide_polygons(a, xa, b, xb)
            sa, sb = pu.box_rec(a[1], a[2]), pu.box_rec(b[1], b[2])
        elif k == 1:
            va, vb = rpoly(), rpoly()
            r = ref.collide_polygons(("verts", va), xa, ("verts", vb), xb)
            sa, sb = pu.polygon_from_ref(ref, va), pu.polygon_from_ref(ref, vb)
        elif k == 2:
            va = rpoly()
            c = [rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)]
            r = ref.collide_polygon_circle(("verts", va), xa, c, xb)
            sa, sb = pu.polygon_from_ref(ref, va), pu.circle_rec(*c)
        elif k == 3:
            c1 = [rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)]
            c2 = [rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)]
            r = ref.collide_circles(c1, xa, c2, xb)
            sa, sb = pu.circle_rec(*c1), pu.circle_rec(*c2)
        else:
            e = [-1, rng.uniform(-.2, .2), 1, rng.uniform(-.2, .2), rng.integers(0, 2), -2, rng.uniform(-1, 1),
                 rng.integers(0, 2), 2, rng.uniform(-1, 1)]
            if k == 4:
                vb = rpoly()
                r = ref.collide_edge_polygon(e, xa, ("verts", vb), xb)
                sa, sb = pu.edge_rec(e), pu.polygon_from_ref(ref, vb)
            else:
                c = [rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)]
                r = ref.collide_edge_circle(e, xa, c, xb)
                sa, sb = pu.edge_rec(e), pu.circle_rec(*c)
        shapesA.append(sa); shapesB.append(sb); xfA.append(xa); xfB.append(xb); expect.append(r)
    np.savez_compressed(os.path.join(HERE, "collide_vectors.npz"), shapeA=np.array(shapesA), shapeB=np.array(shapesB),
                        xfA=np.array(xfA, np.float32), xfB=np.array(xfB, np.float32), manifold=np.array(expect))
    print("collide vectors", len(expect), "touching", int(sum(e[1] > 0 for e in expect)))

    # ---- polygon build / mass vectors ---------------------------------------------------------
    pin, pout = [], []
    for i in range(200):
        v = rpoly()
        if i % 5 == 0:  # duplicates / collinear points exercise welding and the collinearity rule
            v = v + [v[0], ((v[0][0] + v[1][0]) / 2, (v[0][1] + v[1][1]) / 2)]
            v = v[:8]
        buf = np.zeros(17, np.float32)
        buf[0] = len(v)
        buf[1:1 + 2 * len(v)] = np.asarray(v, np.float32).reshape(-1)
        pin.append(buf)
        pout.append(ref.polygon(v, density=1.0 + 0.01 * i))
    np.savez_compressed(os.path.join(HERE, "polygon_vectors.npz"), inp=np.array(pin), out=np.array(pout))

    # ---- sin / cos ------------------------------------------------------------------------------
    a = np.concatenate([rng.uniform(-8, 8, 20000), rng.uniform(-130, 130, 20000), rng.uniform(-1e5, 1e5, 2000),
                        np.array([0.0, -0.0, 1e-7, 0.785398, 0.7853982, 1.5707963, 3.1415927, 6.2831855, 119.99, 120.0, 120.01])]).astype(np.float32)
    s, c = ref.sincos(a)
    np.savez_compressed(os.path.join(HERE, "sincos_vectors.npz"), angle=a, sin=s, cos=c)
    print("done")


if __name__ == "__main__":
    main()
