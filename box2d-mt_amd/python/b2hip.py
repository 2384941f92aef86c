"""ctypes binding of include/b2hip.h (libb2hip.so).

Plumbing only: every call goes straight to the C ABI; there is no Python or CPU implementation of the
step behind it.  Loading fails loudly if the HIP library is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libb2hip.so")

STATIC, KINEMATIC, DYNAMIC = 0, 1, 2
CIRCLE, EDGE, POLYGON = 0, 1, 2
POLYGON_RADIUS = np.float32(2.0) * np.float32(0.005)


class WorldDef(C.Structure):
    _fields_ = [("gravity_x", C.c_float), ("gravity_y", C.c_float), ("allow_sleep", C.c_int),
                ("warm_starting", C.c_int), ("continuous", C.c_int), ("sub_stepping", C.c_int),
                ("auto_clear_forces", C.c_int), ("device", C.c_int)]


class BodyDef(C.Structure):
    _fields_ = [("type", C.c_int), ("px", C.c_float), ("py", C.c_float), ("angle", C.c_float),
                ("vx", C.c_float), ("vy", C.c_float), ("w", C.c_float),
                ("linear_damping", C.c_float), ("angular_damping", C.c_float), ("gravity_scale", C.c_float),
                ("allow_sleep", C.c_int), ("awake", C.c_int), ("fixed_rotation", C.c_int),
                ("bullet", C.c_int), ("active", C.c_int)]


class Shape(C.Structure):
    _fields_ = [("type", C.c_int32), ("count", C.c_int32), ("radius", C.c_float), ("pad", C.c_float),
                ("centroid", C.c_float * 2), ("verts", C.c_float * 16), ("normals", C.c_float * 16)]


class FixtureDef(C.Structure):
    _fields_ = [("density", C.c_float), ("friction", C.c_float), ("restitution", C.c_float),
                ("category_bits", C.c_uint16), ("mask_bits", C.c_uint16), ("group_index", C.c_int16),
                ("pad", C.c_int16), ("is_sensor", C.c_int), ("thick_shape", C.c_int)]


class RevoluteJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("local_anchor_a", C.c_float * 2),
                ("local_anchor_b", C.c_float * 2), ("reference_angle", C.c_float), ("enable_limit", C.c_int),
                ("lower_angle", C.c_float), ("upper_angle", C.c_float), ("enable_motor", C.c_int),
                ("motor_speed", C.c_float), ("max_motor_torque", C.c_float), ("collide_connected", C.c_int)]


class DistanceJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("local_anchor_a", C.c_float * 2),
                ("local_anchor_b", C.c_float * 2), ("length", C.c_float), ("frequency_hz", C.c_float),
                ("damping_ratio", C.c_float), ("collide_connected", C.c_int)]


class PrismaticJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("local_anchor_a", C.c_float * 2),
                ("local_anchor_b", C.c_float * 2), ("local_axis_a", C.c_float * 2), ("reference_angle", C.c_float),
                ("enable_limit", C.c_int), ("lower_translation", C.c_float), ("upper_translation", C.c_float),
                ("enable_motor", C.c_int), ("motor_speed", C.c_float), ("max_motor_force", C.c_float),
                ("collide_connected", C.c_int)]


class WeldJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("local_anchor_a", C.c_float * 2),
                ("local_anchor_b", C.c_float * 2), ("reference_angle", C.c_float), ("frequency_hz", C.c_float),
                ("damping_ratio", C.c_float), ("collide_connected", C.c_int)]


class WheelJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("local_anchor_a", C.c_float * 2),
                ("local_anchor_b", C.c_float * 2), ("local_axis_a", C.c_float * 2), ("frequency_hz", C.c_float),
                ("damping_ratio", C.c_float), ("enable_motor", C.c_int), ("motor_speed", C.c_float),
                ("max_motor_torque", C.c_float), ("collide_connected", C.c_int)]


class RopeJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("local_anchor_a", C.c_float * 2),
                ("local_anchor_b", C.c_float * 2), ("max_length", C.c_float), ("collide_connected", C.c_int)]


class FrictionJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("local_anchor_a", C.c_float * 2),
                ("local_anchor_b", C.c_float * 2), ("max_force", C.c_float), ("max_torque", C.c_float),
                ("collide_connected", C.c_int)]


class MotorJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("linear_offset", C.c_float * 2), ("angular_offset", C.c_float),
                ("max_force", C.c_float), ("max_torque", C.c_float), ("correction_factor", C.c_float),
                ("collide_connected", C.c_int)]


class PulleyJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("ground_anchor_a", C.c_float * 2), ("ground_anchor_b", C.c_float * 2),
                ("local_anchor_a", C.c_float * 2), ("local_anchor_b", C.c_float * 2), ("length_a", C.c_float),
                ("length_b", C.c_float), ("ratio", C.c_float), ("collide_connected", C.c_int)]


class MouseJointDef(C.Structure):
    _fields_ = [("body_a", C.c_int), ("body_b", C.c_int), ("target", C.c_float * 2), ("max_force", C.c_float),
                ("frequency_hz", C.c_float), ("damping_ratio", C.c_float), ("collide_connected", C.c_int)]


class GearJointDef(C.Structure):
    _fields_ = [("joint1", C.c_int), ("joint2", C.c_int), ("ratio", C.c_float), ("collide_connected", C.c_int)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "bodies", "proxies", "contacts", "touching_contacts", "islands", "small_islands", "large_islands",
        "small_island_bodies", "small_island_contacts", "large_island_bodies", "large_island_contacts",
        "colors", "moved_proxies", "new_contacts", "destroyed_contacts", "solver_chunks",
        "pos_iterations_large", "overflow_flags", "toi_events", "toi_calls", "toi_pending_first_pass", "toi_serial_fallbacks",
        "blocks", "cut_constraints", "block_max_rows", "partitions", "block_solver_steps", "free_islands", "sweep_solver_steps",
        "hub_constraints", "hub_fixpoint_rounds", "hub_serial_chunks", "toi_chain_contacts", "toi_pre_solve_reruns", "solver_recoveries")]


BODY_STATE_DTYPE = np.dtype([("px", "f4"), ("py", "f4"), ("angle", "f4"), ("vx", "f4"), ("vy", "f4"), ("w", "f4"),
                             ("cx", "f4"), ("cy", "f4"), ("flags", "u4"), ("sleep_time", "f4")])

CONTACT_DTYPE = np.dtype([("fixture_a", "i4"), ("fixture_b", "i4"), ("body_a", "i4"), ("body_b", "i4"),
                          ("flags", "u4"), ("manifold_type", "i4"), ("point_count", "i4"),
                          ("local_normal", "f4", 2), ("local_point", "f4", 2), ("point_local", "f4", (2, 2)),
                          ("normal_impulse", "f4", 2), ("tangent_impulse", "f4", 2), ("id_key", "u4", 2),
                          ("friction", "f4"), ("restitution", "f4"), ("tangent_speed", "f4")])

_lib = None


def _configure(L, optional_ok=False):
    """Declares the C-ABI signatures on a loaded library. `optional_ok`: the library may lack the measurement / phase
    entry points (the oracle's ABI shim used by the tests exports the world-building and stepping calls only)."""
    L.b2hip_last_error.restype = C.c_char_p
    L.b2hip_version.restype = C.c_char_p
    sigs = {
        "b2hip_world_create": [C.POINTER(WorldDef), C.POINTER(C.c_void_p)],
        "b2hip_world_destroy": [C.c_void_p],
        "b2hip_set_gravity": [C.c_void_p, C.c_float, C.c_float],
        "b2hip_shift_origin": [C.c_void_p, C.c_float, C.c_float],
        "b2hip_set_body_damping": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float],
        "b2hip_fixture_set_material": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float],
        "b2hip_set_flags": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int],
        "b2hip_create_body": [C.c_void_p, C.POINTER(BodyDef)],
        "b2hip_create_fixture": [C.c_void_p, C.c_int, C.POINTER(FixtureDef), C.POINTER(Shape)],
        "b2hip_create_revolute_joint": [C.c_void_p, C.POINTER(RevoluteJointDef)],
        "b2hip_create_distance_joint": [C.c_void_p, C.POINTER(DistanceJointDef)],
        "b2hip_create_prismatic_joint": [C.c_void_p, C.POINTER(PrismaticJointDef)],
        "b2hip_create_weld_joint": [C.c_void_p, C.POINTER(WeldJointDef)],
        "b2hip_create_wheel_joint": [C.c_void_p, C.POINTER(WheelJointDef)],
        "b2hip_create_rope_joint": [C.c_void_p, C.POINTER(RopeJointDef)],
        "b2hip_create_friction_joint": [C.c_void_p, C.POINTER(FrictionJointDef)],
        "b2hip_create_motor_joint": [C.c_void_p, C.POINTER(MotorJointDef)],
        "b2hip_create_pulley_joint": [C.c_void_p, C.POINTER(PulleyJointDef)],
        "b2hip_create_mouse_joint": [C.c_void_p, C.POINTER(MouseJointDef)],
        "b2hip_create_gear_joint": [C.c_void_p, C.POINTER(GearJointDef)],
        "b2hip_destroy_joint": [C.c_void_p, C.c_int],
        "b2hip_joint_set_target": [C.c_void_p, C.c_int, C.c_float, C.c_float],
        "b2hip_joint_set_offsets": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float],
        "b2hip_joint_set_motor": [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float],
        "b2hip_joint_set_limits": [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float],
        "b2hip_body_count": [C.c_void_p],
        "b2hip_fixture_count": [C.c_void_p],
        "b2hip_step": [C.c_void_p, C.c_float, C.c_int, C.c_int],
        "b2hip_step_begin": [C.c_void_p, C.c_float, C.c_int, C.c_int],
        "b2hip_collide": [C.c_void_p], "b2hip_solve": [C.c_void_p], "b2hip_sync_fixtures": [C.c_void_p],
        "b2hip_find_new_contacts": [C.c_void_p], "b2hip_solve_toi": [C.c_void_p], "b2hip_step_end": [C.c_void_p],
        "b2hip_get_body_states": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
        "b2hip_contact_count": [C.c_void_p],
        "b2hip_get_contacts": [C.c_void_p, C.c_int, C.c_void_p],
        "b2hip_save_snapshot": [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)],
        "b2hip_load_snapshot": [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)],
        "b2hip_enable_contact_events": [C.c_void_p, C.c_int],
        "b2hip_get_contact_events": [C.c_void_p, C.c_int, C.c_void_p],
        "b2hip_get_island_labels": [C.c_void_p, C.c_int, C.c_void_p],
        "b2hip_get_fat_aabb": [C.c_void_p, C.c_int, C.POINTER(C.c_float)],
        "b2hip_get_fat_aabbs": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
        "b2hip_get_profile": [C.c_void_p, C.POINTER(C.c_float)],
        "b2hip_get_counters": [C.c_void_p, C.POINTER(Counters)],
        "b2hip_get_solver_timing": [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "b2hip_apply_force": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int],
        "b2hip_set_velocity": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float],
        "b2hip_destroy_body": [C.c_void_p, C.c_int],
        "b2hip_destroy_fixture": [C.c_void_p, C.c_int],
        "b2hip_set_bullet": [C.c_void_p, C.c_int, C.c_int],
        "b2hip_set_awake": [C.c_void_p, C.c_int, C.c_int],
        "b2hip_fixture_set_sensor": [C.c_void_p, C.c_int, C.c_int],
        "b2hip_fixture_refilter": [C.c_void_p, C.c_int],
        "b2hip_set_lazy_readback": [C.c_void_p, C.c_int],
    }
    for name, argtypes in sigs.items():
        try:
            getattr(L, name).argtypes = argtypes
        except AttributeError:
            if not optional_ok:
                raise
    return L


def load(path, optional_ok=False):
    """Any library that exports the b2hip C ABI (the product, or the tests' oracle shim)."""
    return _configure(C.CDLL(path), optional_ok)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libb2hip.so is not built (%s): run __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
        _lib = load(LIB_PATH)
    return _lib


class B2HipError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise B2HipError("b2hip error %d: %s" % (rc, (_lib.b2hip_last_error().decode() if _lib is not None else "?")))
    return rc


def box_shape(hx, hy):
    s = Shape()
    s.type, s.count, s.radius = POLYGON, 4, POLYGON_RADIUS
    hx, hy = np.float32(hx), np.float32(hy)
    v = [-hx, -hy, hx, -hy, hx, hy, -hx, hy]
    n = [0, -1, 1, 0, 0, 1, -1, 0]
    for i in range(8):
        s.verts[i] = v[i]
        s.normals[i] = n[i]
    return s


def circle_shape(radius, px=0.0, py=0.0):
    s = Shape()
    s.type, s.count, s.radius = CIRCLE, 0, radius
    s.verts[0], s.verts[1] = px, py
    return s


def edge_shape(v1, v2):
    s = Shape()
    s.type, s.count, s.radius = EDGE, 0, POLYGON_RADIUS
    s.verts[0], s.verts[1], s.verts[2], s.verts[3] = v1[0], v1[1], v2[0], v2[1]
    return s


class World:
    def __init__(self, gravity=(0.0, -10.0), allow_sleep=True, warm_starting=True, continuous=False, device=-1, library=None):
        self.L = library if library is not None else lib()
        d = WorldDef(gravity[0], gravity[1], int(allow_sleep), int(warm_starting), int(continuous), 0, 1, device)
        p = C.c_void_p()
        _check(self.L.b2hip_world_create(C.byref(d), C.byref(p)))
        self.p = p

    def save_snapshot(self):
        """The world as bytes (b2hip_save_snapshot): bodies, fixtures, joints, contacts with warm-start state, counters."""
        need = C.c_size_t(0)
        _check(self.L.b2hip_save_snapshot(self.p, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _check(self.L.b2hip_save_snapshot(self.p, buf, need.value, C.byref(need)))
        return buf.raw[:need.value]

    @classmethod
    def from_snapshot(cls, blob, device=-1, library=None):
        self = cls.__new__(cls)
        self.L = library if library is not None else lib()
        p = C.c_void_p()
        _check(self.L.b2hip_load_snapshot(blob, len(blob), device, C.byref(p)))
        self.p = p
        return self

    def close(self):
        if self.p:
            self.L.b2hip_world_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def create_body(self, type=STATIC, position=(0.0, 0.0), angle=0.0, velocity=(0.0, 0.0), omega=0.0,
                    linear_damping=0.0, angular_damping=0.0, gravity_scale=1.0, allow_sleep=True, awake=True,
                    fixed_rotation=False, bullet=False):
        d = BodyDef(type, position[0], position[1], angle, velocity[0], velocity[1], omega, linear_damping,
                    angular_damping, gravity_scale, int(allow_sleep), int(awake), int(fixed_rotation), int(bullet), 1)
        return _check(self.L.b2hip_create_body(self.p, C.byref(d)))

    def create_fixture(self, body, shape, density=0.0, friction=0.2, restitution=0.0, category=1, mask=0xFFFF,
                       group=0, sensor=False, thick=False):
        d = FixtureDef(density, friction, restitution, category, mask, group, 0, int(sensor), int(thick))
        return _check(self.L.b2hip_create_fixture(self.p, body, C.byref(d), C.byref(shape)))

    def create_revolute_joint(self, body_a, body_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), reference_angle=0.0,
                              enable_limit=False, lower=0.0, upper=0.0, enable_motor=False, motor_speed=0.0,
                              max_motor_torque=0.0, collide_connected=False):
        d = RevoluteJointDef()
        d.body_a, d.body_b = body_a, body_b
        d.local_anchor_a[0], d.local_anchor_a[1] = anchor_a
        d.local_anchor_b[0], d.local_anchor_b[1] = anchor_b
        d.reference_angle = reference_angle
        d.enable_limit, d.lower_angle, d.upper_angle = int(enable_limit), lower, upper
        d.enable_motor, d.motor_speed, d.max_motor_torque = int(enable_motor), motor_speed, max_motor_torque
        d.collide_connected = int(collide_connected)
        return _check(self.L.b2hip_create_revolute_joint(self.p, C.byref(d)))

    def create_distance_joint(self, body_a, body_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), length=1.0,
                              frequency_hz=0.0, damping_ratio=0.0, collide_connected=False):
        d = DistanceJointDef()
        d.body_a, d.body_b = body_a, body_b
        d.local_anchor_a[0], d.local_anchor_a[1] = anchor_a
        d.local_anchor_b[0], d.local_anchor_b[1] = anchor_b
        d.length, d.frequency_hz, d.damping_ratio = length, frequency_hz, damping_ratio
        d.collide_connected = int(collide_connected)
        return _check(self.L.b2hip_create_distance_joint(self.p, C.byref(d)))

    def create_prismatic_joint(self, body_a, body_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), axis=(1.0, 0.0),
                               reference_angle=0.0, enable_limit=False, lower=0.0, upper=0.0, enable_motor=False,
                               motor_speed=0.0, max_motor_force=0.0, collide_connected=False):
        d = PrismaticJointDef()
        d.body_a, d.body_b = body_a, body_b
        d.local_anchor_a[0], d.local_anchor_a[1] = anchor_a
        d.local_anchor_b[0], d.local_anchor_b[1] = anchor_b
        d.local_axis_a[0], d.local_axis_a[1] = axis
        d.reference_angle = reference_angle
        d.enable_limit, d.lower_translation, d.upper_translation = int(enable_limit), lower, upper
        d.enable_motor, d.motor_speed, d.max_motor_force = int(enable_motor), motor_speed, max_motor_force
        d.collide_connected = int(collide_connected)
        return _check(self.L.b2hip_create_prismatic_joint(self.p, C.byref(d)))

    def create_weld_joint(self, body_a, body_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), reference_angle=0.0,
                          frequency_hz=0.0, damping_ratio=0.0, collide_connected=False):
        d = WeldJointDef()
        d.body_a, d.body_b = body_a, body_b
        d.local_anchor_a[0], d.local_anchor_a[1] = anchor_a
        d.local_anchor_b[0], d.local_anchor_b[1] = anchor_b
        d.reference_angle, d.frequency_hz, d.damping_ratio = reference_angle, frequency_hz, damping_ratio
        d.collide_connected = int(collide_connected)
        return _check(self.L.b2hip_create_weld_joint(self.p, C.byref(d)))

    def _joint_def(self, cls, body_a, body_b, anchor_a, anchor_b, collide_connected):
        d = cls()
        d.body_a, d.body_b = body_a, body_b
        if anchor_a is not None:
            d.local_anchor_a[0], d.local_anchor_a[1] = anchor_a
            d.local_anchor_b[0], d.local_anchor_b[1] = anchor_b
        d.collide_connected = int(collide_connected)
        return d

    def create_wheel_joint(self, body_a, body_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), axis=(1.0, 0.0), frequency_hz=2.0,
                           damping_ratio=0.7, enable_motor=False, motor_speed=0.0, max_motor_torque=0.0, collide_connected=False):
        d = self._joint_def(WheelJointDef, body_a, body_b, anchor_a, anchor_b, collide_connected)
        d.local_axis_a[0], d.local_axis_a[1] = axis
        d.frequency_hz, d.damping_ratio = frequency_hz, damping_ratio
        d.enable_motor, d.motor_speed, d.max_motor_torque = int(enable_motor), motor_speed, max_motor_torque
        return _check(self.L.b2hip_create_wheel_joint(self.p, C.byref(d)))

    def create_rope_joint(self, body_a, body_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), max_length=1.0, collide_connected=False):
        d = self._joint_def(RopeJointDef, body_a, body_b, anchor_a, anchor_b, collide_connected)
        d.max_length = max_length
        return _check(self.L.b2hip_create_rope_joint(self.p, C.byref(d)))

    def create_friction_joint(self, body_a, body_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), max_force=0.0, max_torque=0.0,
                              collide_connected=False):
        d = self._joint_def(FrictionJointDef, body_a, body_b, anchor_a, anchor_b, collide_connected)
        d.max_force, d.max_torque = max_force, max_torque
        return _check(self.L.b2hip_create_friction_joint(self.p, C.byref(d)))

    def create_motor_joint(self, body_a, body_b, linear_offset=(0.0, 0.0), angular_offset=0.0, max_force=1.0, max_torque=1.0,
                           correction_factor=0.3, collide_connected=False):
        d = self._joint_def(MotorJointDef, body_a, body_b, None, None, collide_connected)
        d.linear_offset[0], d.linear_offset[1] = linear_offset
        d.angular_offset = angular_offset
        d.max_force, d.max_torque, d.correction_factor = max_force, max_torque, correction_factor
        return _check(self.L.b2hip_create_motor_joint(self.p, C.byref(d)))

    def create_pulley_joint(self, body_a, body_b, ground_a, ground_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), length_a=1.0,
                            length_b=1.0, ratio=1.0, collide_connected=True):
        d = self._joint_def(PulleyJointDef, body_a, body_b, anchor_a, anchor_b, collide_connected)
        d.ground_anchor_a[0], d.ground_anchor_a[1] = ground_a
        d.ground_anchor_b[0], d.ground_anchor_b[1] = ground_b
        d.length_a, d.length_b, d.ratio = length_a, length_b, ratio
        return _check(self.L.b2hip_create_pulley_joint(self.p, C.byref(d)))

    def create_mouse_joint(self, body_a, body_b, target, max_force, frequency_hz=5.0, damping_ratio=0.7, collide_connected=False):
        d = self._joint_def(MouseJointDef, body_a, body_b, None, None, collide_connected)
        d.target[0], d.target[1] = target
        d.max_force, d.frequency_hz, d.damping_ratio = max_force, frequency_hz, damping_ratio
        return _check(self.L.b2hip_create_mouse_joint(self.p, C.byref(d)))

    def create_gear_joint(self, joint1, joint2, ratio=1.0, collide_connected=False):
        d = GearJointDef(joint1, joint2, ratio, int(collide_connected))
        return _check(self.L.b2hip_create_gear_joint(self.p, C.byref(d)))

    def destroy_joint(self, joint):
        _check(self.L.b2hip_destroy_joint(self.p, joint))

    def destroy_body(self, body):
        """b2World::DestroyBody: the body, its fixtures, its joints and its contacts (ids are never reused)."""
        _check(self.L.b2hip_destroy_body(self.p, body))

    def destroy_fixture(self, fixture):
        _check(self.L.b2hip_destroy_fixture(self.p, fixture))

    def shift_origin(self, x, y):
        _check(self.L.b2hip_shift_origin(self.p, x, y))

    def set_bullet(self, body, flag=True):
        _check(self.L.b2hip_set_bullet(self.p, body, int(flag)))

    def set_awake(self, body, flag=True):
        _check(self.L.b2hip_set_awake(self.p, body, int(flag)))

    def fixture_set_sensor(self, fixture, flag=True):
        _check(self.L.b2hip_fixture_set_sensor(self.p, fixture, int(flag)))

    def fixture_refilter(self, fixture):
        _check(self.L.b2hip_fixture_refilter(self.p, fixture))

    def joint_set_target(self, joint, target):
        _check(self.L.b2hip_joint_set_target(self.p, joint, target[0], target[1]))

    def joint_set_offsets(self, joint, linear_offset, angular_offset):
        _check(self.L.b2hip_joint_set_offsets(self.p, joint, linear_offset[0], linear_offset[1], angular_offset))

    def joint_set_motor(self, joint, enable_motor, motor_speed, max_motor):
        _check(self.L.b2hip_joint_set_motor(self.p, joint, int(enable_motor), motor_speed, max_motor))

    def joint_set_limits(self, joint, enable_limit, lower, upper):
        _check(self.L.b2hip_joint_set_limits(self.p, joint, int(enable_limit), lower, upper))

    def apply_force(self, body, force=(0.0, 0.0), torque=0.0, wake=True):
        _check(self.L.b2hip_apply_force(self.p, body, force[0], force[1], torque, int(wake)))

    def set_flags(self, allow_sleep=True, warm_starting=True, continuous=False, sub_stepping=False):
        _check(self.L.b2hip_set_flags(self.p, int(allow_sleep), int(warm_starting), int(continuous), int(sub_stepping)))

    def set_velocity(self, body, velocity=(0.0, 0.0), omega=0.0):
        _check(self.L.b2hip_set_velocity(self.p, body, velocity[0], velocity[1], omega))

    def set_lazy_readback(self, enable=True):
        """The 40 B per body of a step's read-back stay in HBM until body_states() (or an edit) asks for them."""
        _check(self.L.b2hip_set_lazy_readback(self.p, int(enable)))

    def step(self, dt=1.0 / 60.0, vel_iters=8, pos_iters=3):
        _check(self.L.b2hip_step(self.p, dt, vel_iters, pos_iters))

    @property
    def body_count(self):
        return self.L.b2hip_body_count(self.p)

    def body_states(self):
        n = self.body_count
        out = np.zeros(n, BODY_STATE_DTYPE)
        _check(self.L.b2hip_get_body_states(self.p, 0, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def bodies8(self):
        """Same 8-column layout as the harness: x, y, angle, vx, vy, w, awake, type."""
        s = self.body_states()
        out = np.zeros((s.size, 8), np.float32)
        out[:, 0], out[:, 1], out[:, 2] = s["px"], s["py"], s["angle"]
        out[:, 3], out[:, 4], out[:, 5] = s["vx"], s["vy"], s["w"]
        out[:, 6] = (s["flags"] & 4) != 0
        out[:, 7] = s["flags"] & 3
        return out

    @property
    def contact_count(self):
        return self.L.b2hip_contact_count(self.p)

    def contacts(self):
        cap = max(self.contact_count, 1)
        out = np.zeros(cap, CONTACT_DTYPE)
        n = _check(self.L.b2hip_get_contacts(self.p, cap, out.ctypes.data_as(C.c_void_p)))
        return out[:n]

    def enable_contact_events(self, enable=True):
        _check(self.L.b2hip_enable_contact_events(self.p, 1 if enable else 0))

    def contact_events(self):
        """Begin / end events of the last step: rows (fixture_a, fixture_b, kind 0 begin / 1 end, contact_index or -1)."""
        n = _check(self.L.b2hip_get_contact_events(self.p, 0, None))
        out = np.zeros((max(n, 1), 4), np.int32)
        n = _check(self.L.b2hip_get_contact_events(self.p, n, out.ctypes.data_as(C.c_void_p)))
        return out[:n]

    def island_labels(self):
        n = self.body_count
        out = np.zeros(n, np.int32)
        _check(self.L.b2hip_get_island_labels(self.p, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def counters(self):
        c = Counters()
        _check(self.L.b2hip_get_counters(self.p, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def profile(self):
        ms = (C.c_float * 13)()
        _check(self.L.b2hip_get_profile(self.p, ms))
        return list(ms)

    def solver_timing(self):
        ms, by, ct, b = C.c_float(), C.c_double(), C.c_int(), C.c_int()
        _check(self.L.b2hip_get_solver_timing(self.p, C.byref(ms), C.byref(by), C.byref(ct), C.byref(b)))
        return ms.value, by.value, ct.value, b.value
