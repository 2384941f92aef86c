"""Aggregates of a world's state that any valid Gauss-Seidel order must reproduce statistically - what the large piles of
BASELINE configs 3 and 4 are compared by in the window bench.py times, where the device's coloured order and the reference's
order have long left each other's trajectory (tests/test_gpu_settled_windows.py; goldens from the reference build:
tests/golden/make_golden_settled.py). Works on any tests/b2harness.py world (reference build, oracle, device).

Per sample: contact count (b2World::GetContactCount), touching contacts with points, the separation of every manifold point
from the body poses and manifolds as they stand after the step (b2PositionSolverManifold::Initialize, b2ContactSolver.cpp:620-673;
polygon / edge manifolds, radii 2 x b2_polygonRadius) as deepest / 99th percentile / mean penetration, the summed normal
impulse, the kinetic energy and top speed of the dynamic bodies, and the angle of body 1 (the Tumbler's container).
"""
import ctypes as C

import numpy as np

KEYS = ("contacts", "touching", "penetration_max", "penetration_p99", "penetration_mean", "impulse_sum", "kinetic_energy",
        "speed_max", "speed_mean", "angle_body1")


def _contacts_unsorted(w):
    cap = max(w.contact_count, 1)
    ids = np.zeros((cap, 4), np.int32)
    flags = np.zeros(cap, np.int32)
    man = np.zeros((cap, 16), np.float32)
    n = w.L.b2h_get_contacts(w.ptr, cap, ids.ctypes.data_as(C.POINTER(C.c_int)), flags.ctypes.data_as(C.POINTER(C.c_int)),
                             man.ctypes.data_as(C.POINTER(C.c_float)))
    return ids[:n], flags[:n], man[:n]


def aggregates(w, mass=None):
    b = w.bodies().astype(np.float64)
    if mass is None:
        mass = w.mass()
    ids, flags, man = _contacts_unsorted(w)
    n_contacts = len(ids)
    pc = man[:, 1].astype(np.int32)
    sel = ((flags & 1) != 0) & (pc > 0)
    ids, man, pc = ids[sel], man[sel].astype(np.float64), pc[sel]
    px, py, qs, qc = b[:, 0], b[:, 1], np.sin(b[:, 2]), np.cos(b[:, 2])

    def to_world(body, lx, ly):
        return px[body] + qc[body] * lx - qs[body] * ly, py[body] + qs[body] * lx + qc[body] * ly

    mtype = man[:, 0].astype(np.int32)
    poly = (mtype == 1) | (mtype == 2)  # e_faceA / e_faceB (b2Collision.h:96-101); circles do not occur in these scenes
    ids, man, pc, mtype = ids[poly], man[poly], pc[poly], mtype[poly]
    face_a = mtype == 1
    ref = np.where(face_a, ids[:, 0], ids[:, 2])
    inc = np.where(face_a, ids[:, 2], ids[:, 0])
    nx = qc[ref] * man[:, 2] - qs[ref] * man[:, 3]
    ny = qs[ref] * man[:, 2] + qc[ref] * man[:, 3]
    plx, ply = to_world(ref, man[:, 4], man[:, 5])
    sep = np.full(len(ids), np.inf)
    imp = np.zeros(len(ids))
    for k in range(2):
        has = pc > k
        cx, cy = to_world(inc, man[:, 6 + 5 * k], man[:, 7 + 5 * k])
        d = (cx - plx) * nx + (cy - ply) * ny - 0.02
        sep = np.where(has, np.minimum(sep, d), sep)
        imp += np.where(has, man[:, 8 + 5 * k], 0.0)
    pen = -sep[np.isfinite(sep)]
    dyn = b[:, 7] == 2
    m, inertia = mass[:, 0].astype(np.float64), mass[:, 1].astype(np.float64)
    v2 = b[:, 3] ** 2 + b[:, 4] ** 2
    ke = float((0.5 * m[dyn] * v2[dyn] + 0.5 * inertia[dyn] * b[dyn, 5] ** 2).sum())
    speed = np.sqrt(v2[dyn])
    return {"contacts": float(n_contacts), "touching": float(len(ids)),
            "penetration_max": float(pen.max()) if pen.size else 0.0,
            "penetration_p99": float(np.percentile(pen, 99)) if pen.size else 0.0,
            "penetration_mean": float(np.maximum(pen, 0.0).mean()) if pen.size else 0.0,
            "impulse_sum": float(imp.sum()), "kinetic_energy": ke,
            "speed_max": float(speed.max()) if speed.size else 0.0, "speed_mean": float(speed.mean()) if speed.size else 0.0,
            "angle_body1": float(b[1, 2]) if len(b) > 1 else 0.0}


def window(w, first, last, every, on_sample=None):
    """Steps the world (which stands at step `first` - 1 ... i.e. has made `first` steps) to `last`, sampling the aggregates
    after steps first, first + every, ... <= last. Returns (steps, table[len(steps), len(KEYS)])."""
    mass = w.mass()
    steps, rows = [], []
    done = first
    while True:
        a = aggregates(w, mass)
        steps.append(done)
        rows.append([a[k] for k in KEYS])
        if on_sample is not None:
            on_sample(done, a)
        if done + every > last:
            break
        w.step(every)
        done += every
    return np.array(steps, np.int32), np.array(rows, np.float64)
