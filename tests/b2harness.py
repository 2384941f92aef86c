"""ctypes binding of the headless harness C ABI (box2d-mt_amd/harness/harness.cpp).

TEST INFRASTRUCTURE.  The same ABI is exported by two shared libraries built from one source:
  oracle/_ref/libb2ref_harness.so      the real reference (skitzoid/Box2D-MT) compiled where it lies
  box2d-mt_amd/libb2amd_harness.so     this repo's drop-in Box2D API on top of the HIP C-ABI (libb2hip.so)
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libb2ref_harness.so")
AMD_LIB = os.path.join(ROOT, "box2d-mt_amd", "libb2amd_harness.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libb2oracle_harness.so")

HELLO, PYRAMID, TUMBLER, FIELD, PILES, RAIN, CIRCLE_STACK, BULLETS, SENSORS, ROPES, MACHINES, VEHICLES, LIFECYCLE, CHAINS, PROPS = range(15)
F_CONTINUOUS, F_SLEEP, F_WARM, F_SUBSTEP = 1, 2, 4, 8
DEFAULT_FLAGS = F_SLEEP | F_WARM  # CCD off unless a test asks for it

PROFILE_FIELDS = ["step", "collide", "solve", "solveTraversal", "solveInit", "solveVelocity",
                  "solvePosition", "solveTOI", "solveTOIFindMinContact", "broadphase",
                  "broadphaseSyncFixtures", "broadphaseFindContacts", "locking"]

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _fptr(a):
    return a.ctypes.data_as(_fp)


def _iptr(a):
    return a.ctypes.data_as(_ip)


class Harness:
    """One loaded backend library."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        L = self.lib
        L.b2h_backend.restype = C.c_char_p
        L.b2h_create.restype = C.c_void_p
        L.b2h_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint, C.c_int, C.c_int]
        L.b2h_destroy.argtypes = [C.c_void_p]
        L.b2h_default_iters.argtypes = [C.c_void_p, _ip, _ip]
        L.b2h_step.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int]
        L.b2h_body_count.argtypes = [C.c_void_p]
        L.b2h_get_bodies.argtypes = [C.c_void_p, _fp]
        L.b2h_get_mass.argtypes = [C.c_void_p, _fp]
        L.b2h_contact_count.argtypes = [C.c_void_p]
        L.b2h_get_contacts.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _fp]
        L.b2h_get_profile.argtypes = [C.c_void_p, _fp]
        L.b2h_reset_profile.argtypes = [C.c_void_p]
        L.b2h_query_aabb.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int)]
        L.b2h_query_aabb.restype = C.c_int
        L.b2h_raycast_closest.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.b2h_raycast_closest.restype = C.c_int
        L.b2h_record_events.argtypes = [C.c_void_p, C.c_int]
        L.b2h_get_events.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.b2h_get_events.restype = C.c_int
        L.b2h_get_events_ex.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.b2h_get_events_ex.restype = C.c_int
        L.b2h_set_filter.argtypes = [C.c_void_p, C.c_int]

    @property
    def backend(self):
        return self.lib.b2h_backend().decode()

    def world(self, scene, p0=0, p1=0, f0=0.0, f1=0.0, seed=1, flags=DEFAULT_FLAGS, threads=1):
        return World(self, scene, p0, p1, f0, f1, seed, flags, threads)

    # ---- per-function probes -------------------------------------------------------------
    def _poly_args(self, poly):
        """poly = ('box', hx, hy) or ('verts', [(x,y)...])"""
        v = np.zeros(16, np.float32)
        if poly[0] == "box":
            v[0], v[1] = poly[1], poly[2]
            return 4, v, 1
        pts = np.asarray(poly[1], np.float32).reshape(-1)
        v[:pts.size] = pts
        return pts.size // 2, v, 0

    def collide_polygons(self, polyA, xfA, polyB, xfB):
        ca, va, ba = self._poly_args(polyA)
        cb, vb, bb = self._poly_args(polyB)
        out = np.zeros(16, np.float32)
        a = np.asarray(xfA, np.float32)
        b = np.asarray(xfB, np.float32)
        self.lib.b2h_probe_collide_polygons(ca, _fptr(va), ba, _fptr(a), cb, _fptr(vb), bb, _fptr(b), _fptr(out))
        return out

    def collide_polygon_circle(self, polyA, xfA, circleB, xfB):
        ca, va, ba = self._poly_args(polyA)
        out = np.zeros(16, np.float32)
        a = np.asarray(xfA, np.float32)
        b = np.asarray(xfB, np.float32)
        c = np.asarray(circleB, np.float32)
        self.lib.b2h_probe_collide_polygon_circle(ca, _fptr(va), ba, _fptr(a), _fptr(c), _fptr(b), _fptr(out))
        return out

    def collide_circles(self, circleA, xfA, circleB, xfB):
        out = np.zeros(16, np.float32)
        ca = np.asarray(circleA, np.float32)
        cb = np.asarray(circleB, np.float32)
        a = np.asarray(xfA, np.float32)
        b = np.asarray(xfB, np.float32)
        self.lib.b2h_probe_collide_circles(_fptr(ca), _fptr(a), _fptr(cb), _fptr(b), _fptr(out))
        return out

    def collide_edge_polygon(self, edgeA, xfA, polyB, xfB):
        cb, vb, bb = self._poly_args(polyB)
        out = np.zeros(16, np.float32)
        e = np.asarray(edgeA, np.float32)
        a = np.asarray(xfA, np.float32)
        b = np.asarray(xfB, np.float32)
        self.lib.b2h_probe_collide_edge_polygon(_fptr(e), _fptr(a), cb, _fptr(vb), bb, _fptr(b), _fptr(out))
        return out

    def collide_edge_circle(self, edgeA, xfA, circleB, xfB):
        out = np.zeros(16, np.float32)
        e = np.asarray(edgeA, np.float32)
        a = np.asarray(xfA, np.float32)
        b = np.asarray(xfB, np.float32)
        c = np.asarray(circleB, np.float32)
        self.lib.b2h_probe_collide_edge_circle(_fptr(e), _fptr(a), _fptr(c), _fptr(b), _fptr(out))
        return out

    def polygon(self, verts, density=1.0):
        pts = np.zeros(16, np.float32)
        flat = np.asarray(verts, np.float32).reshape(-1)
        pts[:flat.size] = flat
        out = np.zeros(39, np.float32)
        self.lib.b2h_probe_polygon(flat.size // 2, _fptr(pts), C.c_float(density), _fptr(out))
        return out

    # ---- continuous-collision probes (reference build only) ---------------------------------
    def distance(self, vertsA, radiusA, xfA, vertsB, radiusB, xfB, use_radii=False):
        """b2Distance on raw vertex proxies -> [pointA.x, pointA.y, pointB.x, pointB.y, distance, iterations]"""
        va = np.ascontiguousarray(vertsA, np.float32).reshape(-1)
        vb = np.ascontiguousarray(vertsB, np.float32).reshape(-1)
        a = np.asarray(xfA, np.float32)
        b = np.asarray(xfB, np.float32)
        out = np.zeros(6, np.float32)
        self.lib.b2h_probe_distance(va.size // 2, _fptr(va), C.c_float(radiusA), _fptr(a), vb.size // 2, _fptr(vb),
                                    C.c_float(radiusB), _fptr(b), int(use_radii), _fptr(out))
        return out

    def toi(self, vertsA, radiusA, sweepA, vertsB, radiusB, sweepB, t_max=1.0):
        """b2TimeOfImpact on raw vertex proxies; sweep = [lcx, lcy, c0x, c0y, cx, cy, a0, a, alpha0] -> [state, t]"""
        va = np.ascontiguousarray(vertsA, np.float32).reshape(-1)
        vb = np.ascontiguousarray(vertsB, np.float32).reshape(-1)
        a = np.ascontiguousarray(sweepA, np.float32)
        b = np.ascontiguousarray(sweepB, np.float32)
        out = np.zeros(2, np.float32)
        self.lib.b2h_probe_toi(va.size // 2, _fptr(va), C.c_float(radiusA), _fptr(a), vb.size // 2, _fptr(vb),
                               C.c_float(radiusB), _fptr(b), C.c_float(t_max), _fptr(out))
        return out

    def sincos(self, angles):
        a = np.ascontiguousarray(angles, np.float32)
        s = np.empty_like(a)
        c = np.empty_like(a)
        self.lib.b2h_probe_sincos(a.size, _fptr(a), _fptr(s), _fptr(c))
        return s, c


class World:
    def __init__(self, h, scene, p0, p1, f0, f1, seed, flags, threads):
        self.h = h
        self.L = h.lib
        self.ptr = self.L.b2h_create(scene, p0, p1, C.c_float(f0), C.c_float(f1), seed, flags, threads)
        if not self.ptr:
            raise RuntimeError("b2h_create failed")
        vi, pi = C.c_int(), C.c_int()
        self.L.b2h_default_iters(self.ptr, C.byref(vi), C.byref(pi))
        self.vel_iters, self.pos_iters = vi.value, pi.value

    def close(self):
        if self.ptr:
            self.L.b2h_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        # a test that fails (or forgets) must not keep its world: on the GPU a world is two streams, and a pytest process
        # that piles them up leaves the queues of other processes (the torchrun self-tests) waiting for a slot
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def step(self, n=1, dt=1.0 / 60.0, vel_iters=None, pos_iters=None):
        self.L.b2h_step(self.ptr, n, C.c_float(dt), vel_iters or self.vel_iters, pos_iters or self.pos_iters)

    @property
    def body_count(self):
        return self.L.b2h_body_count(self.ptr)

    def bodies(self):
        out = np.zeros((self.body_count, 8), np.float32)
        self.L.b2h_get_bodies(self.ptr, _fptr(out))
        return out

    def mass(self):
        out = np.zeros((self.body_count, 6), np.float32)
        self.L.b2h_get_mass(self.ptr, _fptr(out))
        return out

    @property
    def contact_count(self):
        return self.L.b2h_contact_count(self.ptr)

    def contacts(self):
        """Returns (ids[n,4], flags[n], manifold[n,16]) sorted by ids so backends are comparable."""
        cap = max(self.contact_count, 1)
        ids = np.zeros((cap, 4), np.int32)
        flags = np.zeros(cap, np.int32)
        man = np.zeros((cap, 16), np.float32)
        n = self.L.b2h_get_contacts(self.ptr, cap, _iptr(ids), _iptr(flags), _fptr(man))
        ids, flags, man = ids[:n], flags[:n], man[:n]
        order = np.lexsort((ids[:, 3], ids[:, 2], ids[:, 1], ids[:, 0]))
        return ids[order], flags[order], man[order]

    def debug_draw(self, flags=0x1f):
        """b2World::DrawDebugData into a counting b2Draw: (calls per primitive kind [7], checksum of everything it was handed)"""
        out = (C.c_longlong * 8)()
        self.L.b2h_debug_draw.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_longlong)]
        self.L.b2h_debug_draw(self.ptr, flags, out)
        return list(out[:7]), int(out[7])

    def contact_materials(self):
        """(ids[n,4], material[n,3]: friction, restitution, tangent speed) of every contact, sorted by ids."""
        cap = max(self.contact_count, 1)
        ids = np.zeros((cap, 4), np.int32)
        flags = np.zeros(cap, np.int32)
        man = np.zeros((cap, 16), np.float32)
        mat = np.zeros((cap, 3), np.float32)
        n = self.L.b2h_get_contacts(self.ptr, cap, _iptr(ids), _iptr(flags), _fptr(man))
        self.L.b2h_get_contact_materials.argtypes = [C.c_void_p, C.c_int, _fp]
        self.L.b2h_get_contact_materials(self.ptr, cap, _fptr(mat))
        ids, mat = ids[:n], mat[:n]
        order = np.lexsort((ids[:, 3], ids[:, 2], ids[:, 1], ids[:, 0]))
        return ids[order], mat[order]

    def device_world(self):
        """b2hip_world* behind the drop-in b2World (AMD backend only), for the C-ABI measurement hooks."""
        self.L.b2h_device_world.restype = C.c_void_p
        self.L.b2h_device_world.argtypes = [C.c_void_p]
        return self.L.b2h_device_world(self.ptr)

    def profile(self):
        out = np.zeros(13, np.float32)
        n = self.L.b2h_get_profile(self.ptr, _fptr(out))
        d = dict(zip(PROFILE_FIELDS, out.tolist()))
        d["steps"] = n
        return d

    def query_aabb(self, lo, hi, cap=1 << 16):
        """b2World::QueryAABB: sorted rows (body, fixture index in body) of every fixture whose fat AABB overlaps the box."""
        out = np.zeros((cap, 2), np.int32)
        n = self.L.b2h_query_aabb(self.ptr, lo[0], lo[1], hi[0], hi[1], cap, _iptr(out))
        return out[:min(n, cap)].copy()

    def raycast_closest(self, p1, p2):
        """b2World::RayCast with a closest-hit callback: None, or (body, fixture, point.xy, normal.xy, fraction)."""
        out = np.zeros(7, np.float32)
        if not self.L.b2h_raycast_closest(self.ptr, p1[0], p1[1], p2[0], p2[1], _fptr(out)):
            return None
        return out

    def record_events(self, enable=True, mode=None):
        """Install (or remove) the harness's recording b2ContactListener. mode bits: 1 begin / end (the default), 2 PreSolve,
        4 PostSolve, 8 PreSolve disables the contacts a fixed rule of the body indices picks (SetEnabled(false))."""
        self.L.b2h_record_events(self.ptr, (mode if mode is not None else 1) if enable else 0)

    def events_ex(self, cap=1 << 18):
        """Every recorded callback since the last call, in call order: rows of 10 ints (see box2d-mt_amd/harness/harness.cpp):
        kind (0 begin, 1 end, 2 PreSolve, 3 PostSolve), bodyA, fixtureA, bodyB, fixtureB, five payload words."""
        out = np.zeros((cap, 10), np.int32)
        n = self.L.b2h_get_events_ex(self.ptr, cap, _iptr(out))
        return out[:min(n, cap)].copy()

    def set_filter(self, enable=True):
        """Install (or remove) the harness's user b2ContactFilter (default rule AND a fixed rule of the body indices)."""
        self.L.b2h_set_filter(self.ptr, 1 if enable else 0)

    def events(self, cap=1 << 16):
        """BeginContact / EndContact callbacks since the last call, in call order: rows (kind, bodyA, fixtureA, bodyB, fixtureB)."""
        out = np.zeros((cap, 5), np.int32)
        n = self.L.b2h_get_events(self.ptr, cap, _iptr(out))
        return out[:min(n, cap)].copy()

    def reset_profile(self):
        self.L.b2h_reset_profile(self.ptr)


def fnv1a64(arr):
    """FNV-1a over the raw bytes of an array (pose hashes in the fixtures)."""
    h = 0xcbf29ce484222325
    for b in np.ascontiguousarray(arr).view(np.uint8).reshape(-1).tolist():
        h ^= b
        h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def have_ref():
    return os.path.exists(REF_LIB)


def have_oracle():
    return os.path.exists(ORACLE_LIB)


def have_amd():
    return os.path.exists(AMD_LIB)
