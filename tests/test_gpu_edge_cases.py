"""GPU edge cases through the plain C ABI (include/b2hip.h): the same world is built twice with the same calls, once on
libb2hip.so (HIP) and once on the oracle's ABI shim, and every step is compared bitwise. Covers what the scene harness
does not: empty and static-only worlds, dt = 0, bodies without fixtures, forces / velocities set between steps, waking a
sleeping pile, sensors, collision filters, fixed rotation, damping, gravity scale, bullets, zero iterations."""
import os

import numpy as np
import pytest

import b2harness as bh
import b2hip

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs(built_libs):
    if not bh.have_amd():
        pytest.fail("libb2hip.so missing (no CPU fallback)")
    return b2hip.lib(), b2hip.load(bh.ORACLE_LIB, optional_ok=True)


def both(libs, **kw):
    return b2hip.World(library=libs[0], **kw), b2hip.World(library=libs[1], **kw)


def same(a, b, what="", skip=()):
    sa, sb = a.body_states(), b.body_states()
    if skip:  # (destroyed bodies: their ids stay, what their rows hold is not defined)
        keep = np.ones(len(sa), bool)
        keep[list(skip)] = False
        sa, sb = sa[keep], sb[keep]
    for f in ("px", "py", "angle", "vx", "vy", "w", "cx", "cy", "sleep_time"):
        assert np.array_equal(sa[f].view(np.uint32), sb[f].view(np.uint32)), "%s: %s differs" % (what, f)
    assert np.array_equal(sa["flags"] & 0x7f, sb["flags"] & 0x7f), "%s: flags differ" % what
    assert a.contact_count == b.contact_count, "%s: contact count" % what
    ca, cb = a.contacts(), b.contacts()
    for f in ("fixture_a", "fixture_b", "flags", "point_count", "id_key", "normal_impulse", "tangent_impulse"):
        assert np.array_equal(ca[f].view(np.uint32) if ca[f].dtype.kind == "f" else ca[f],
                              cb[f].view(np.uint32) if cb[f].dtype.kind == "f" else cb[f]), "%s: contact %s differs" % (what, f)


def run(a, b, steps, what, dt=1.0 / 60.0, vi=8, pi=3, between=None):
    for s in range(steps):
        if between:
            between(s, a)
            between(s, b)
        a.step(dt, vi, pi)
        b.step(dt, vi, pi)
        same(a, b, "%s step %d" % (what, s))


def test_empty_and_static_only_worlds(libs):
    a, b = both(libs)
    run(a, b, 3, "empty")
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, -1.0))
        w.create_fixture(g, b2hip.box_shape(10.0, 1.0))
        g2 = w.create_body(b2hip.STATIC, (3.0, 0.5), angle=0.3)
        w.create_fixture(g2, b2hip.circle_shape(0.5))
    run(a, b, 3, "static only")
    assert a.contact_count == 0
    a.close(); b.close()


def test_zero_dt_and_zero_iterations(libs):
    a, b = both(libs)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, -1.0))
        w.create_fixture(g, b2hip.box_shape(10.0, 1.0))
        d = w.create_body(b2hip.DYNAMIC, (0.0, 0.4), velocity=(1.0, 0.0))
        w.create_fixture(d, b2hip.box_shape(0.5, 0.5), density=1.0)
    run(a, b, 2, "dt=0", dt=0.0)
    run(a, b, 20, "normal")
    run(a, b, 2, "dt=0 again", dt=0.0)
    run(a, b, 10, "no iterations", vi=0, pi=0)
    run(a, b, 10, "one iteration", vi=1, pi=1)
    a.close(); b.close()


def test_body_without_fixture_and_body_options(libs):
    a, b = both(libs)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, -1.0))
        w.create_fixture(g, b2hip.box_shape(20.0, 1.0), friction=0.6)
        w.create_body(b2hip.DYNAMIC, (5.0, 5.0))                      # no fixture: unit mass, falls forever
        d1 = w.create_body(b2hip.DYNAMIC, (-3.0, 2.0), angle=0.4, fixed_rotation=True)
        w.create_fixture(d1, b2hip.box_shape(0.4, 0.7), density=2.0, friction=0.1)
        d2 = w.create_body(b2hip.DYNAMIC, (-1.0, 3.0), linear_damping=0.8, angular_damping=0.5, omega=5.0)
        w.create_fixture(d2, b2hip.circle_shape(0.3), density=1.0, restitution=0.6)
        d3 = w.create_body(b2hip.DYNAMIC, (1.0, 3.0), gravity_scale=0.25, allow_sleep=False)
        w.create_fixture(d3, b2hip.box_shape(0.3, 0.3), density=1.0)
        d4 = w.create_body(b2hip.DYNAMIC, (3.0, 2.0), gravity_scale=-0.5)   # floats upwards
        w.create_fixture(d4, b2hip.circle_shape(0.2, 0.1, 0.0), density=1.0)  # off-centre circle: non-zero local centre
        k = w.create_body(b2hip.KINEMATIC, (0.0, 1.0), velocity=(0.5, 0.0), omega=0.3)
        w.create_fixture(k, b2hip.box_shape(1.0, 0.1))
    run(a, b, 150, "options")
    a.close(); b.close()


def test_sensor_and_filters(libs):
    a, b = both(libs)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, -1.0))
        w.create_fixture(g, b2hip.box_shape(20.0, 1.0))
        s = w.create_body(b2hip.STATIC, (0.0, 1.0))
        w.create_fixture(s, b2hip.box_shape(2.0, 0.2), sensor=True)                    # bodies fall through
        for i in range(6):
            d = w.create_body(b2hip.DYNAMIC, (-2.5 + i, 3.0 + 0.3 * i))
            # categories: even boxes ignore odd boxes (mask), group -1 never collides within the group
            w.create_fixture(d, b2hip.box_shape(0.45, 0.45), density=1.0, category=1 << (i & 1), mask=0xFFFF ^ (2 >> (i & 1)) if i < 4 else 0xFFFF,
                             group=-1 if i >= 4 else 0)
    run(a, b, 120, "sensor/filter")
    a.close(); b.close()


def test_forces_velocities_and_waking(libs):
    a, b = both(libs)
    ids = {}
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, -1.0))
        w.create_fixture(g, b2hip.box_shape(20.0, 1.0))
        ids[w] = []
        for i in range(4):
            d = w.create_body(b2hip.DYNAMIC, (0.0, 0.5 + 1.0 * i))
            w.create_fixture(d, b2hip.box_shape(0.5, 0.5), density=1.0)
            ids[w].append(d)

    def between(s, w):
        top = ids[w][3]
        if 5 <= s < 15:
            w.apply_force(top, (30.0, 0.0), torque=2.0)
        if s == 200:
            w.apply_force(ids[w][0], (0.0, 0.0), torque=0.0, wake=False)   # no-op on a sleeping body
        if s == 220:
            w.set_velocity(ids[w][1], (0.0, 4.0), omega=1.0)                # wakes the pile

    run(a, b, 300, "forces", between=between)
    a.close(); b.close()


def test_pile_falls_asleep_and_all_flags_match(libs):
    a, b = both(libs)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 0.0))
        w.create_fixture(g, b2hip.edge_shape((-10.0, 0.0), (10.0, 0.0)))
        for i in range(5):
            d = w.create_body(b2hip.DYNAMIC, (0.01 * i, 0.51 + 1.01 * i))
            w.create_fixture(d, b2hip.box_shape(0.5, 0.5), density=1.0, friction=0.5)
    run(a, b, 260, "sleep")
    assert (a.body_states()["flags"][1:] & 4).sum() == 0, "the settled pile must be asleep"
    a.close(); b.close()


def test_bullet_through_thin_wall_with_ccd(libs):
    a, b = both(libs, continuous=True, gravity=(0.0, 0.0))
    for w in (a, b):
        wall = w.create_body(b2hip.STATIC, (5.0, 0.0))
        w.create_fixture(wall, b2hip.box_shape(0.05, 5.0))
        for i in range(5):
            d = w.create_body(b2hip.DYNAMIC, (0.0, -2.0 + i), velocity=(150.0 + 20.0 * i, 3.0 * i), bullet=(i % 2 == 0))
            w.create_fixture(d, b2hip.circle_shape(0.1) if i % 2 else b2hip.box_shape(0.1, 0.1), density=1.0, restitution=0.3)
    run(a, b, 60, "ccd")
    assert (a.body_states()["px"][1:] < 5.0).all(), "nothing may tunnel through the wall with continuous physics on"
    a.close(); b.close()


def test_revolute_pendulum_chain(libs, monkeypatch):
    """Several joints in one island, limit and motor rows: exact-order mode (islands with joints take the coloured path by
    default, where joints are visited in id order instead of the reference's traversal order)."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 10.0))
        prev = g
        for i in range(4):
            d = w.create_body(b2hip.DYNAMIC, (0.5 + i, 10.0))
            w.create_fixture(d, b2hip.box_shape(0.5, 0.1), density=2.0)
            w.create_revolute_joint(prev, d, anchor_a=(0.0, 0.0) if i == 0 else (0.5, 0.0), anchor_b=(-0.5, 0.0),
                                    enable_limit=(i == 2), lower=-0.5, upper=0.5, enable_motor=(i == 0), motor_speed=1.0, max_motor_torque=50.0)
            prev = d
    run(a, b, 200, "pendulum")
    a.close(); b.close()


def test_distance_joints_rods_and_springs(libs, monkeypatch):
    """Distance joints (b2DistanceJoint.cpp): a rope of circles on rigid rods swinging into a wall, a weight on a soft
    spring, and a box held by a rod and a revolute joint at once (two joint types in one island), collide_connected both
    ways. Exact-order mode, as for every island with joints."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 0.0))
        w.create_fixture(g, b2hip.box_shape(30.0, 0.5))
        wall = w.create_body(b2hip.STATIC, (6.0, 6.0))
        w.create_fixture(wall, b2hip.box_shape(0.5, 6.0))
        prev, prev_anchor = g, (0.0, 12.0)
        for i in range(8):
            d = w.create_body(b2hip.DYNAMIC, (-1.0 - i, 12.0))
            w.create_fixture(d, b2hip.circle_shape(0.3), density=1.0 + 0.5 * i, restitution=0.2)
            w.create_distance_joint(prev, d, anchor_a=prev_anchor, anchor_b=(0.0, 0.0), length=1.0, collide_connected=bool(i & 1))
            prev, prev_anchor = d, (0.0, 0.0)
        m = w.create_body(b2hip.DYNAMIC, (12.0, 6.0))
        w.create_fixture(m, b2hip.box_shape(0.5, 0.5), density=3.0)
        w.create_distance_joint(g, m, anchor_a=(12.0, 12.0), anchor_b=(0.2, 0.5), length=4.0, frequency_hz=1.5, damping_ratio=0.2)
        k = w.create_body(b2hip.DYNAMIC, (16.0, 8.0), angle=0.4)
        w.create_fixture(k, b2hip.box_shape(1.0, 0.25), density=1.0)
        w.create_revolute_joint(g, k, anchor_a=(15.0, 8.0), anchor_b=(-1.0, 0.0))
        w.create_distance_joint(m, k, anchor_a=(0.0, 0.0), anchor_b=(1.0, 0.0), length=5.0, frequency_hz=3.0, damping_ratio=0.7,
                                collide_connected=True)
    run(a, b, 300, "distance joints")
    a.close(); b.close()


def test_distance_joints_without_warm_starting(libs, monkeypatch):
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs, warm_starting=False)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 0.0))
        w.create_fixture(g, b2hip.box_shape(10.0, 0.5))
        d = w.create_body(b2hip.DYNAMIC, (3.0, 5.0))
        w.create_fixture(d, b2hip.box_shape(0.4, 0.4), density=1.0)
        w.create_distance_joint(g, d, anchor_a=(0.0, 5.0), anchor_b=(0.0, 0.0), length=3.0)
        e = w.create_body(b2hip.DYNAMIC, (3.0, 3.0))
        w.create_fixture(e, b2hip.circle_shape(0.3), density=1.0)
        w.create_distance_joint(d, e, anchor_a=(0.0, -0.4), anchor_b=(0.0, 0.0), length=1.6, frequency_hz=5.0, damping_ratio=0.1)
    run(a, b, 150, "distance joints, no warm start")
    a.close(); b.close()


def test_prismatic_and_weld_joints(libs, monkeypatch):
    """Prismatic joints (b2PrismaticJoint.cpp) in each limit state with and without motor, on a moving (dynamic) parent
    too, and weld joints (b2WeldJoint.cpp) rigid, soft and with a fixed-rotation pair (singular angular row). Exact-order mode."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs)
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 0.0))
        w.create_fixture(g, b2hip.box_shape(40.0, 0.5))
        cart = w.create_body(b2hip.DYNAMIC, (0.0, 1.5))
        w.create_fixture(cart, b2hip.box_shape(2.0, 0.5), density=2.0, friction=0.1)
        w.create_prismatic_joint(g, cart, anchor_a=(0.0, 1.5), anchor_b=(0.0, 0.0), axis=(2.0, 0.0), enable_limit=True, lower=-5.0,
                                 upper=5.0, enable_motor=True, motor_speed=4.0, max_motor_force=400.0)
        mast = w.create_body(b2hip.DYNAMIC, (0.0, 4.0))
        w.create_fixture(mast, b2hip.box_shape(0.2, 2.0), density=0.5)
        w.create_weld_joint(cart, mast, anchor_a=(0.0, 0.5), anchor_b=(0.0, -2.0))
        flag = w.create_body(b2hip.DYNAMIC, (1.0, 6.0))
        w.create_fixture(flag, b2hip.box_shape(0.8, 0.1), density=0.5)
        w.create_weld_joint(mast, flag, anchor_a=(0.2, 2.0), anchor_b=(-0.8, 0.0), frequency_hz=3.0, damping_ratio=0.3)
        bob = w.create_body(b2hip.DYNAMIC, (3.0, 3.0))
        w.create_fixture(bob, b2hip.circle_shape(0.4), density=1.0)
        w.create_prismatic_joint(cart, bob, anchor_a=(1.5, 0.5), anchor_b=(0.0, -1.0), axis=(0.5, 1.0), reference_angle=0.1,
                                 enable_limit=True, lower=-0.5, upper=1.5)
        pin = w.create_body(b2hip.DYNAMIC, (-10.0, 5.0), fixed_rotation=True)
        w.create_fixture(pin, b2hip.box_shape(0.3, 0.3), density=1.0)
        nail = w.create_body(b2hip.DYNAMIC, (-10.0, 6.0), fixed_rotation=True)
        w.create_fixture(nail, b2hip.circle_shape(0.3), density=1.0)
        w.create_weld_joint(pin, nail, anchor_a=(0.0, 0.5), anchor_b=(0.0, -0.5))
        w.create_prismatic_joint(g, pin, anchor_a=(-10.0, 5.0), anchor_b=(0.0, 0.0), axis=(0.0, 1.0), enable_limit=True,
                                 lower=-0.001, upper=0.001)
        for i in range(20):
            d = w.create_body(b2hip.DYNAMIC, (-4.0 + 0.45 * i, 9.0 + 0.7 * (i % 5)), angle=0.1 * i)
            w.create_fixture(d, b2hip.box_shape(0.2, 0.2) if i % 2 else b2hip.circle_shape(0.2), density=1.0)
    run(a, b, 250, "prismatic + weld")
    a.close(); b.close()


def test_wheel_rope_friction_motor_joints(libs, monkeypatch):
    """The four remaining simple joint types through the plain C ABI: a cart on sprung / rigid wheel joints whose motor is
    reversed mid-run, a rope that goes taut, a friction joint between two dynamic bodies, a motor joint whose target offsets
    are moved between steps and one between two dynamic bodies. Warm starting off for the second half (set_flags)."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs)
    ids = {}
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 0.0))
        w.create_fixture(g, b2hip.box_shape(60.0, 0.5))
        cart = w.create_body(b2hip.DYNAMIC, (0.0, 1.6))
        w.create_fixture(cart, b2hip.box_shape(1.5, 0.3), density=1.0)
        for k, x in enumerate((-1.0, 1.0)):
            wh = w.create_body(b2hip.DYNAMIC, (x, 0.95))
            w.create_fixture(wh, b2hip.circle_shape(0.45), density=1.0, friction=0.9)
            ids["wheel%d" % k] = w.create_wheel_joint(cart, wh, anchor_a=(x, -0.65), anchor_b=(0.0, 0.0), axis=(0.0, 1.0),
                                                      frequency_hz=4.0 if k == 0 else 0.0, damping_ratio=0.7, enable_motor=(k == 0),
                                                      motor_speed=-8.0, max_motor_torque=15.0)
        bob = w.create_body(b2hip.DYNAMIC, (10.0, 8.0))
        w.create_fixture(bob, b2hip.circle_shape(0.3), density=2.0)
        w.create_rope_joint(g, bob, anchor_a=(10.0, 10.0), anchor_b=(0.0, 0.3), max_length=4.0)
        w.set_velocity(bob, (5.0, 0.0), 0.0)
        sled = w.create_body(b2hip.DYNAMIC, (-10.0, 1.0))
        w.create_fixture(sled, b2hip.box_shape(2.0, 0.5), density=1.0, friction=0.0)
        rider = w.create_body(b2hip.DYNAMIC, (-10.0, 1.9))
        w.create_fixture(rider, b2hip.box_shape(0.5, 0.4), density=1.0, friction=0.0)
        w.create_friction_joint(sled, rider, anchor_a=(0.0, 0.9), anchor_b=(0.0, 0.0), max_force=3.0, max_torque=1.0, collide_connected=True)
        w.set_velocity(rider, (4.0, 0.0), 0.0)
        plat = w.create_body(b2hip.DYNAMIC, (20.0, 5.0))
        w.create_fixture(plat, b2hip.box_shape(1.5, 0.2), density=1.0)
        ids["servo"] = w.create_motor_joint(g, plat, linear_offset=(20.0, 5.0), max_force=500.0, max_torque=200.0)
        tail = w.create_body(b2hip.DYNAMIC, (23.0, 5.0))
        w.create_fixture(tail, b2hip.box_shape(0.5, 0.2), density=0.5)
        w.create_motor_joint(plat, tail, linear_offset=(3.0, 0.5), angular_offset=0.5, max_force=50.0, max_torque=20.0, correction_factor=0.8)
        for i in range(12):
            d = w.create_body(b2hip.DYNAMIC, (18.0 + 0.4 * i, 8.0 + 0.6 * (i % 4)))
            w.create_fixture(d, b2hip.box_shape(0.2, 0.2) if i % 2 else b2hip.circle_shape(0.2), density=1.0)

    def between(s, w):
        import math
        w.joint_set_offsets(ids["servo"], (20.0 + 2.0 * math.sin(0.05 * s), 5.0 + math.sin(0.1 * s)), 0.2 * math.sin(0.03 * s))
        if s == 120:
            w.joint_set_motor(ids["wheel0"], True, 8.0, 15.0)
        if s == 150:
            w.set_flags(allow_sleep=True, warm_starting=False, continuous=False)

    run(a, b, 260, "wheel/rope/friction/motor", between=between)
    a.close(); b.close()


def test_many_jointed_islands(libs, monkeypatch):
    """240 cars (chassis + two wheels on wheel joints, one driven) on three static strips: 240 separate islands that all
    carry joints (per-island joint lists, joint-aware pair filtering between every chassis and its wheels), a revolute
    trailer hitched to every fourth car. Exact-order mode, bitwise against the oracle."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs)
    for w in (a, b):
        for r in range(3):
            g = w.create_body(b2hip.STATIC, (480.0, 10.0 * r - 0.25))
            w.create_fixture(g, b2hip.box_shape(500.0, 0.25))
        for r in range(3):
            for c in range(80):
                x, y = 12.0 * c + 2.0, 10.0 * r + 1.0
                ch = w.create_body(b2hip.DYNAMIC, (x, y))
                w.create_fixture(ch, b2hip.box_shape(1.5, 0.4), density=1.0)
                for k, dx in enumerate((-1.0, 1.0)):
                    wh = w.create_body(b2hip.DYNAMIC, (x + dx, y - 0.6))
                    w.create_fixture(wh, b2hip.circle_shape(0.4), density=1.0, friction=0.9)
                    w.create_wheel_joint(ch, wh, anchor_a=(dx, -0.6), axis=(0.0, 1.0), frequency_hz=4.0, damping_ratio=0.7,
                                         enable_motor=(k == 0), motor_speed=-2.0 - 0.05 * c, max_motor_torque=20.0)
                if c % 4 == 0:
                    tr = w.create_body(b2hip.DYNAMIC, (x + 3.2, y - 0.3))
                    w.create_fixture(tr, b2hip.box_shape(1.0, 0.2), density=0.5, friction=0.1)
                    w.create_revolute_joint(ch, tr, anchor_a=(1.8, -0.3), anchor_b=(-1.4, 0.0), collide_connected=(c % 8 == 0))
    run(a, b, 120, "many jointed islands")
    assert a.counters()["islands"] >= 240
    a.close(); b.close()


def test_destroy_joint(libs, monkeypatch):
    """b2hip_destroy_joint (b2World::DestroyJoint): a pendulum chain cut in the middle, a welded pair whose members may collide
    only once the weld (collide_connected = false) is gone, a joint destroyed while its bodies sleep (they wake), the second of
    two joints between the same bodies, and creating new joints afterwards (ids are not reused)."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs)
    ids = {}
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 0.0))
        w.create_fixture(g, b2hip.box_shape(40.0, 0.5))
        prev = g
        for i in range(6):
            d = w.create_body(b2hip.DYNAMIC, (0.5 + i, 12.0))
            w.create_fixture(d, b2hip.box_shape(0.5, 0.1), density=2.0)
            ids["link%d" % i] = w.create_revolute_joint(prev, d, anchor_a=(0.0, 12.0) if i == 0 else (0.5, 0.0), anchor_b=(-0.5, 0.0))
            prev = d
        p = w.create_body(b2hip.DYNAMIC, (-10.0, 3.0))
        w.create_fixture(p, b2hip.box_shape(1.0, 0.5), density=1.0)
        q = w.create_body(b2hip.DYNAMIC, (-10.0, 3.6))
        w.create_fixture(q, b2hip.box_shape(0.5, 0.5), density=1.0)
        ids["weld"] = w.create_weld_joint(p, q, anchor_a=(0.0, 0.5), anchor_b=(0.0, -0.1))   # overlapping boxes, no contact while welded
        ids["rod"] = w.create_distance_joint(p, q, anchor_a=(0.9, 0.0), anchor_b=(0.4, 0.0), length=0.8, collide_connected=True)
        s_ = w.create_body(b2hip.DYNAMIC, (15.0, 1.0))
        w.create_fixture(s_, b2hip.box_shape(0.5, 0.5), density=1.0)
        ids["tether"] = w.create_rope_joint(g, s_, anchor_a=(15.0, 6.0), anchor_b=(0.0, 0.5), max_length=6.0)
        ids["bodies"] = (p, q, s_, prev)

    def between(s, w):
        if s == 90:
            w.destroy_joint(ids["link3"])
        if s == 140:
            w.destroy_joint(ids["weld"])
        if s == 200:
            w.destroy_joint(ids["rod"])
        if s == 260:
            w.destroy_joint(ids["tether"])          # its body has been asleep on the ground for a while
            n = w.create_revolute_joint(ids["bodies"][0], ids["bodies"][1], anchor_a=(1.0, 0.5), anchor_b=(0.5, -0.5))
            assert n > ids["tether"]

    run(a, b, 330, "destroy joint", between=between)
    with pytest.raises(b2hip.B2HipError):
        a.destroy_joint(ids["weld"])
    a.close(); b.close()


def test_joint_setters_between_steps(libs, monkeypatch):
    """b2hip_joint_set_motor / b2hip_joint_set_limits: reverse a slider's motor, switch motors off and on, move and drop the
    limits of a revolute arm while everything has gone to sleep (the setters wake both bodies, as the reference's do), and
    calls that change nothing (no wake-up)."""
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = both(libs)
    ids = {}
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, 0.0))
        w.create_fixture(g, b2hip.box_shape(40.0, 0.5))
        cart = w.create_body(b2hip.DYNAMIC, (0.0, 1.5))
        w.create_fixture(cart, b2hip.box_shape(1.0, 0.5), density=2.0, friction=0.2)
        ids["slider"] = w.create_prismatic_joint(g, cart, anchor_a=(0.0, 1.5), axis=(1.0, 0.0), enable_limit=True, lower=-3.0, upper=3.0,
                                                 enable_motor=True, motor_speed=3.0, max_motor_force=300.0)
        arm = w.create_body(b2hip.DYNAMIC, (10.0, 5.0))
        w.create_fixture(arm, b2hip.box_shape(2.0, 0.2), density=1.0)
        ids["hinge"] = w.create_revolute_joint(g, arm, anchor_a=(8.0, 5.0), anchor_b=(-2.0, 0.0), enable_limit=True, lower=-0.3, upper=0.3)
        for i in range(6):
            d = w.create_body(b2hip.DYNAMIC, (-1.0 + 0.4 * i, 2.5 + 0.5 * i))
            w.create_fixture(d, b2hip.box_shape(0.2, 0.2), density=1.0)

    def between(s, w):
        if s == 60:
            w.joint_set_motor(ids["slider"], True, -3.0, 300.0)
        if s == 100:
            w.joint_set_motor(ids["slider"], False, -3.0, 300.0)
        if s in (250, 251):
            w.joint_set_limits(ids["hinge"], True, -1.2, -0.6)     # asleep by now: wakes the arm (251: no change, no-op)
        if s == 330:
            w.joint_set_limits(ids["hinge"], False, -1.2, -0.6)
            w.joint_set_motor(ids["hinge"], True, 2.0, 80.0)
            w.joint_set_motor(ids["slider"], True, 1.0, 50.0)
            w.joint_set_limits(ids["slider"], True, -6.0, 0.5)

    run(a, b, 420, "joint setters", between=between)
    a.close(); b.close()


def test_edits_on_a_body_with_more_contacts_than_one_edit_pass(libs):
    """The edit kernel lists the contacts of an edited body / fixture in an LDS list of 8 192 entries and works a longer
    list off in passes (b2d_kernels_edit.h; the reference walks the body's contact list without a limit, b2World.cpp:617-625,
    b2Fixture.cpp:187-210). A ground body under 9 000 boxes: its fixture is re-filtered and turned into a sensor and back,
    then the body is destroyed with all 9 000 contacts at once; a second ground 0.2 below catches the boxes."""
    a, b = both(libs, continuous=True)
    ids = {}
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, -0.5))
        ids["gfix"] = w.create_fixture(g, b2hip.box_shape(3000.0, 0.5))
        g2 = w.create_body(b2hip.STATIC, (0.0, -1.2))
        w.create_fixture(g2, b2hip.box_shape(3000.0, 0.5))
        ids["ground"] = g
        for i in range(9000):
            body = w.create_body(b2hip.DYNAMIC, (-2700.0 + 0.6 * i, 0.26))
            w.create_fixture(body, b2hip.box_shape(0.25, 0.25), density=1.0)

    def between(s, w):
        if s == 4:
            w.fixture_refilter(ids["gfix"])
        if s == 6:
            w.fixture_set_sensor(ids["gfix"], True)
        if s == 7:
            w.fixture_set_sensor(ids["gfix"], False)
        if s == 12:
            w.destroy_body(ids["ground"])

    run(a, b, 6, "wide ground", between=between)
    assert a.contact_count >= 9000
    for s in range(6, 30):
        between(s, a)
        between(s, b)
        a.step()
        b.step()
        same(a, b, "wide ground step %d" % s, skip=(ids["ground"],) if s >= 12 else ())
    assert a.contact_count >= 9000  # (on the lower ground now)
    a.close(); b.close()


def test_dense_start_grows_the_pair_buffer(monkeypatch):
    """1 400 bodies and 450 bullets crammed into a 70 x 70 arena: the first pair update finds several times more candidate
    pairs than the buffer was sized for (21 000 contacts on step one). The buffers grow and the search runs again - no
    capacity error - and the result is still the oracle's, bit for bit (exact-order mode: the islands are huge).
    Step 1 destroys 5 600 contacts, more TOI candidates among them than the LDS tables of toiOrderDestroy hold (2 048): the
    removals are replayed in chunks (b2d_kernels_collide.h), and k_toi_first - which walks the manager's slot table since
    round 5 - finds every candidate in steps 2 - 4 (until round 5 the surplus left the table inconsistent, unnoticed)."""
    import b2harness as bh
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    amd, orc = bh.Harness(bh.AMD_LIB), bh.Harness(bh.ORACLE_LIB)
    kw = dict(p0=1406, p1=456, f0=35.0, f1=2.0, seed=2623, flags=bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM)
    a, o = amd.world(bh.FIELD, **kw), orc.world(bh.FIELD, **kw)
    for s in range(5):
        a.step(1)
        o.step(1)
        assert a.contact_count == o.contact_count and a.contact_count > 15000, "step %d" % s
        assert np.array_equal(a.bodies().view(np.uint32), o.bodies().view(np.uint32)), "step %d" % s
    a.close()
    o.close()


def test_lazy_readback_gives_the_same_states(libs):
    """b2hip_set_lazy_readback: the rows of a step stay on the device until somebody asks. A stack that is pushed, woken and
    grown between steps (edits pull single bodies; new bodies make the host's mirror grow) is read every 7th step only - and
    must be what the oracle has, bit for bit; so must a snapshot taken while rows are outstanding, and the world after the
    mode is switched off again."""
    a, b = both(libs, continuous=True)
    a.set_lazy_readback(True)
    b.set_lazy_readback(True)  # (the shim accepts and ignores it)
    ids = {}
    for w in (a, b):
        g = w.create_body(b2hip.STATIC, (0.0, -1.0))
        w.create_fixture(g, b2hip.box_shape(40.0, 1.0))
        ids[w] = []
        for i in range(40):
            d = w.create_body(b2hip.DYNAMIC, (-15.0 + 0.8 * (i % 20), 0.5 + 1.01 * (i // 20)))
            w.create_fixture(d, b2hip.box_shape(0.4, 0.5) if i % 3 else b2hip.circle_shape(0.4), density=1.0)
            ids[w].append(d)

    def between(s, w):
        if s % 5 == 2:
            w.apply_force(ids[w][s % 40], (20.0, 5.0), torque=1.0)          # pulls one row (the fetch), edits it
        if s % 11 == 4:
            w.set_velocity(ids[w][(3 * s) % 40], (0.0, 3.0), omega=-2.0)
        if s in (30, 31, 90):                                                # the mirror grows with rows outstanding
            for k in range(30 if s == 90 else 3):
                d = w.create_body(b2hip.DYNAMIC, (-10.0 + 0.7 * k, 6.0 + 0.01 * s))
                w.create_fixture(d, b2hip.box_shape(0.3, 0.3), density=2.0)
                ids[w].append(d)
        if s == 60:
            w.destroy_body(ids[w][5])

    for s in range(140):
        between(s, a)
        between(s, b)
        a.step()
        b.step()
        if s % 7 == 6 or s in (30, 31, 32, 90, 91):
            same(a, b, "lazy read-back, step %d" % s, skip=(ids[a][5],) if s >= 60 else ())
        if s == 100:
            blob = a.save_snapshot()  # (rows outstanding: the snapshot fetches them)
            c = b2hip.World.from_snapshot(blob, library=libs[0])
            same(c, b, "snapshot of a lazy world", skip=(ids[a][5],))
            c.close()
    a.set_lazy_readback(False)
    for s in range(140, 150):
        a.step()
        b.step()
    same(a, b, "lazy read-back switched off", skip=(ids[a][5],))
    a.close(); b.close()
