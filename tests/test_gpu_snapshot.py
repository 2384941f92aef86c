"""World snapshot (second half): save after N steps, load into a NEW world, continue - the loaded world
must go on bit for bit like the one it was taken from (bodies, contact array, warm-start impulses, cached impacts, colours)."""
import numpy as np
import pytest

import b2hip

pytestmark = pytest.mark.gpu


def build(continuous):
    w = b2hip.World(gravity=(0.0, -10.0), continuous=continuous)
    ground = w.create_body(b2hip.STATIC)
    w.create_fixture(ground, b2hip.edge_shape((-40.0, 0.0), (40.0, 0.0)))
    w.create_fixture(ground, b2hip.box_shape(0.5, 6.0), friction=0.4)
    rng = np.random.default_rng(3)
    for i in range(260):
        b = w.create_body(b2hip.DYNAMIC, position=(float(rng.uniform(-12, 12)), 1.0 + 0.55 * (i // 8) + float(rng.uniform(0, 0.2))),
                          angle=float(rng.uniform(0, 6.28)), velocity=(float(rng.uniform(-2, 2)), 0.0), bullet=(i % 37 == 0))
        shape = b2hip.circle_shape(float(rng.uniform(0.15, 0.35))) if i % 3 == 0 else b2hip.box_shape(float(rng.uniform(0.15, 0.4)), float(rng.uniform(0.15, 0.3)))
        w.create_fixture(b, shape, density=1.0, friction=0.3, restitution=0.1 if i % 5 == 0 else 0.0)
    arm = w.create_body(b2hip.DYNAMIC, position=(20.0, 6.0))
    w.create_fixture(arm, b2hip.box_shape(2.0, 0.2), density=2.0)
    w.create_revolute_joint(ground, arm, anchor_a=(18.0, 6.0), anchor_b=(-2.0, 0.0), enable_motor=True, motor_speed=1.5, max_motor_torque=500.0)
    # one of each remaining stateful joint family: spring (wheel), clamped rows (motor), soft target (mouse), unilateral (rope)
    cart = w.create_body(b2hip.DYNAMIC, position=(28.0, 1.5))
    w.create_fixture(cart, b2hip.box_shape(1.5, 0.3), density=1.0)
    for x in (-1.0, 1.0):
        wheel = w.create_body(b2hip.DYNAMIC, position=(28.0 + x, 0.9))
        w.create_fixture(wheel, b2hip.circle_shape(0.4), density=1.0, friction=0.9)
        w.create_wheel_joint(cart, wheel, anchor_a=(x, -0.6), axis=(0.0, 1.0), frequency_hz=4.0, damping_ratio=0.7, enable_motor=x < 0,
                             motor_speed=-3.0, max_motor_torque=10.0)
    plat = w.create_body(b2hip.DYNAMIC, position=(-25.0, 6.0))
    w.create_fixture(plat, b2hip.box_shape(1.5, 0.2), density=1.0)
    w.create_motor_joint(ground, plat, linear_offset=(-24.0, 7.0), angular_offset=0.3, max_force=300.0, max_torque=100.0)
    crate = w.create_body(b2hip.DYNAMIC, position=(-32.0, 3.0))
    w.create_fixture(crate, b2hip.box_shape(0.6, 0.6), density=1.0)
    w.mouse = w.create_mouse_joint(ground, crate, target=(-31.7, 3.2), max_force=1500.0)
    w.joint_set_target(w.mouse, (-30.0, 6.0))
    bob = w.create_body(b2hip.DYNAMIC, position=(-36.0, 5.0), velocity=(3.0, 0.0))
    w.create_fixture(bob, b2hip.circle_shape(0.3), density=1.0)
    w.create_rope_joint(ground, bob, anchor_a=(-36.0, 8.0), anchor_b=(0.0, 0.0), max_length=3.5)
    # a gear between two hinged discs (its record lives in its own device array: trailing snapshot section)
    hinges = []
    for x, r in ((34.0, 0.8), (36.5, 1.6)):
        disc = w.create_body(b2hip.DYNAMIC, position=(x, 9.0))
        w.create_fixture(disc, b2hip.circle_shape(r), density=2.0)
        hinges.append(w.create_revolute_joint(ground, disc, anchor_a=(x, 9.0), enable_motor=(r < 1.0), motor_speed=2.0, max_motor_torque=60.0))
    w.create_gear_joint(hinges[0], hinges[1], ratio=2.0)
    return w


def state(w):
    s = w.body_states()
    c = w.contacts()
    return s.tobytes(), w.contact_count, c.tobytes()


@pytest.mark.parametrize("continuous", [False, True])
def test_snapshot_resume_is_bit_exact(continuous):
    a = build(continuous)
    for _ in range(90):
        a.step()
    blob = a.save_snapshot()
    assert len(blob) > 10000
    b = b2hip.World.from_snapshot(blob)
    assert b.body_count == a.body_count
    assert state(a) == state(b), "the loaded world does not show the state it was saved from"
    for s in range(80):
        a.step()
        b.step()
        assert state(a) == state(b), "the loaded world parts from the original at step %d after the snapshot" % s
    # edits after the load go through the same mirrors
    a.apply_force(5, (30.0, 10.0), 1.0)
    b.apply_force(5, (30.0, 10.0), 1.0)
    a.joint_set_target(a.mouse, (-33.0, 4.0))
    b.joint_set_target(a.mouse, (-33.0, 4.0))
    na = a.create_body(b2hip.DYNAMIC, position=(0.0, 30.0))
    nb = b.create_body(b2hip.DYNAMIC, position=(0.0, 30.0))
    assert na == nb
    a.create_fixture(na, b2hip.box_shape(0.3, 0.3), density=1.0)
    b.create_fixture(nb, b2hip.box_shape(0.3, 0.3), density=1.0)
    for s in range(40):
        a.step()
        b.step()
        assert state(a) == state(b), "after edits: step %d" % s
    a.close()
    b.close()


def test_snapshot_rejects_garbage():
    with pytest.raises(b2hip.B2HipError):
        b2hip.World.from_snapshot(b"not a snapshot at all" * 10)


def test_snapshot_rejects_truncated_and_bit_flipped_blobs():
    """A damaged snapshot is refused (B2HIP_ERR_INVALID) before anything is copied: every count is checked against the blob
    and against the others, every index against its range. Truncations at many lengths, then single-byte damage at offsets
    spread over the header, the host tables and the device sections: each load either fails cleanly or (damage that only
    touched float payload) yields a world that can be stepped and destroyed."""
    a = build(False)
    for _ in range(30):
        a.step()
    blob = a.save_snapshot()
    a.close()
    n = len(blob)
    for cut in [0, 7, 8, 40, 96, 97, 200, n // 7, n // 3, n // 2, n - 4097, n - 1]:
        with pytest.raises(b2hip.B2HipError):
            b2hip.World.from_snapshot(blob[:cut])
    rng = np.random.default_rng(11)
    refused = 0
    offsets = list(range(8, 112, 4)) + [int(x) for x in rng.integers(112, n, 120)]
    for off in offsets:
        bad = bytearray(blob)
        bad[off] ^= 0xFF
        bad[min(off + 3, n - 1)] ^= 0x7F  # the high byte too: makes counts and indices wild, not off by one
        try:
            w = b2hip.World.from_snapshot(bytes(bad))
        except b2hip.B2HipError:
            refused += 1
            continue
        try:
            w.step()  # survived validation: payload damage only (or a counter whose damage the step itself reports)
        except b2hip.B2HipError:
            pass
        w.close()
    assert refused >= 20, "header damage must be refused (%d refusals)" % refused


def test_snapshot_right_after_joint_edits_keeps_the_refilter_flags():
    """a snapshot taken between CreateJoint / DestroyJoint and the next step must carry the pending contact
    re-filter (b2World.cpp:716-732, 833-845), and the contact-event switch."""
    a = build(False)
    for _ in range(60):
        a.step()
    # weld two bodies that are touching right now (collide_connected = false): their contact must go at the next collide
    c = a.contacts()
    touching = c[(c["flags"] & 1) != 0]
    pair = next((int(r["body_a"]), int(r["body_b"])) for r in touching if r["body_a"] > 1 and r["body_b"] > 1)
    a.create_weld_joint(pair[0], pair[1])
    a.enable_contact_events(True)
    b = b2hip.World.from_snapshot(a.save_snapshot())
    for s in range(30):
        a.step()
        b.step()
        assert state(a) == state(b), "step %d after the snapshot" % s
        assert np.array_equal(a.contact_events(), b.contact_events()), "contact events differ at step %d" % s
    a.close()
    b.close()
