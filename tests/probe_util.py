"""Helpers to build ShapeRec blobs (box2d-mt_amd/csrc/b2d_collide.h) for the CPU probe of the device math."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE_SRC = os.path.join(ROOT, "tests", "probe", "host_probe.cpp")
PROBE_LIB = os.path.join(ROOT, "tests", "probe", "libhost_probe.so")

CIRCLE, EDGE, POLYGON = 0, 1, 2


def build_probe():
    hdrs = [os.path.join(ROOT, "box2d-mt_amd", "csrc", h) for h in ("b2d_math.h", "b2d_collide.h", "b2d_solver.h", "b2d_toi.h")]
    newest = max(os.path.getmtime(p) for p in hdrs + [PROBE_SRC])
    if not os.path.exists(PROBE_LIB) or os.path.getmtime(PROBE_LIB) < newest:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                               "-I", os.path.join(ROOT, "box2d-mt_amd", "csrc"), "-o", PROBE_LIB, PROBE_SRC])
    return C.CDLL(PROBE_LIB)


def shape_rec(kind, count=0, radius=0.0, centroid=(0, 0), verts=(), normals=()):
    buf = np.zeros(38, np.float32)
    iv = buf.view(np.int32)
    iv[0] = kind
    iv[1] = count
    buf[2] = radius
    buf[4:6] = centroid
    v = np.asarray(verts, np.float32).reshape(-1)
    buf[6:6 + v.size] = v
    n = np.asarray(normals, np.float32).reshape(-1)
    buf[22:22 + n.size] = n
    return buf


def polygon_from_ref(ref, verts):
    """Build the polygon record from the reference's own b2PolygonShape::Set output."""
    o = ref.polygon(verts)
    cnt = int(o[0])
    return shape_rec(POLYGON, cnt, 0.01, o[33:35], o[1:1 + 2 * cnt], o[17:17 + 2 * cnt])


def box_rec(hx, hy):
    return shape_rec(POLYGON, 4, 0.01, (0, 0), [-hx, -hy, hx, -hy, hx, hy, -hx, hy], [0, -1, 1, 0, 0, 1, -1, 0])


def circle_rec(px, py, r):
    return shape_rec(CIRCLE, 0, r, (0, 0), [px, py])


def edge_rec(edge10):
    e = np.asarray(edge10, np.float32)
    flags = (1 if e[4] != 0 else 0) | (2 if e[7] != 0 else 0)
    return shape_rec(EDGE, flags, 0.01, (0, 0), [e[0], e[1], e[2], e[3], e[5], e[6], e[8], e[9]])
