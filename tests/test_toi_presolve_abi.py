"""PreSolve through the C ABI, called from INSIDE a TOI sub-step (include/b2hip.h: b2hip_toi_callback; reference:
b2World.cpp:866-881 - Update(listener) on the TOI contact, `if (!IsEnabled() || !IsTouching()) { restore the sweeps; continue; }`).

A small box falls fast onto a thin static platform with continuous physics on: the impact is found by the TOI phase (the
box would tunnel otherwise), the contact begins to touch inside the sub-step, and that sub-step's PreSolve is the FIRST
call the listener gets for it. A listener that answers 0 - the one-sided platform of Testbed/Tests/OneSidedPlatform.h - must
let the box through; one that answers 1 must stop it. Asked after the fact (rounds 2 and 3 until this test) the answer came
too late: the sub-step had already put the box on the platform.

  CPU : the oracle behind the same ABI (oracle/b2o_abi_shim.c; the oracle equals the reference build on this:
        tests/test_listener.py)
  GPU : the product; also: a callback that edits the world AND changes its contact is refused loudly
"""
import ctypes as C

import numpy as np
import pytest

import b2harness as bh
import b2hip

PRE_SOLVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float * 3))


def drop_on_platform(L, answer, edit=None, steps=40, force=None, rows=None):
    """-> ((lowest height the box reached, its slowest downward speed), number of PreSolve calls, rc of the first failing step or 0)
    force: applied to the box before every step (the host touches the body's row every frame); rows: list that receives the
    box's state after every step."""
    w = b2hip.World(gravity=(0.0, -10.0), continuous=True, library=L)
    ground = w.create_body(b2hip.STATIC, position=(0.0, 0.0))
    w.create_fixture(ground, b2hip.box_shape(20.0, 0.05))          # the platform: 0.1 thick
    box = w.create_body(b2hip.DYNAMIC, position=(0.0, 3.0), velocity=(0.0, -60.0))  # 1 m per step: tunnels without TOI
    w.create_fixture(box, b2hip.box_shape(0.1, 0.1), density=1.0)
    calls = []

    def pre_solve(user, contact, fa, fb, old_manifold, manifold, material):
        calls.append((fa, fb))
        if edit is not None:
            edit(w, box)
        return answer

    fn = PRE_SOLVE_FN(pre_solve)
    L.b2hip_set_pre_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.b2hip_set_pre_solve(w.p, C.cast(fn, C.c_void_p), None) == 0
    lowest, slowest, rc = 1e9, -1e9, 0
    for _ in range(steps):
        if force is not None:
            w.apply_force(box, force)
        rc = L.b2hip_step(w.p, C.c_float(1.0 / 60.0), 8, 3)
        if rc != 0:
            break
        row = w.bodies8()[box]
        if rows is not None:
            rows.append(row.copy())
        lowest, slowest = min(lowest, float(row[1])), max(slowest, float(row[4]))
    L.b2hip_set_pre_solve(w.p, None, None)
    w.close()
    return (lowest, slowest), len(calls), rc


def check(L):
    stopped, calls_on, rc = drop_on_platform(L, 1)
    assert rc == 0 and calls_on > 0
    assert stopped[0] > 0.1 and stopped[1] > -1.0, "continuous physics must stop the box on the platform (lowest y %g)" % stopped[0]
    through, calls_off, rc = drop_on_platform(L, 0)
    assert rc == 0 and calls_off > 0
    assert through[0] < -1.0, "a contact switched off from the sub-step's own PreSolve must let the box through (lowest y %g)" % through[0]
    # ... at full speed: an answer that came after the sub-step would find the box stopped on the platform, to fall from rest
    assert through[1] < -55.0, "the sub-step solved the contact its PreSolve had switched off (slowest vy %g)" % through[1]
    return stopped + through


def test_oracle_abi_one_sided_platform_through_a_toi_event():
    check(b2hip.load(bh.ORACLE_LIB, optional_ok=True))


@pytest.mark.gpu
def test_device_abi_one_sided_platform_through_a_toi_event():
    a = check(b2hip.lib())
    b = check(b2hip.load(bh.ORACLE_LIB, optional_ok=True))
    assert np.array_equal(np.float32(a).view(np.uint32), np.float32(b).view(np.uint32)), "device %s, oracle %s" % (a, b)


@pytest.mark.gpu
def test_device_refuses_a_toi_presolve_that_edits_the_world_and_changes_its_contact():
    """World edits from such a call are taken like edits between steps; together with an answer that sends the phase back to
    its snapshot the host mirror of the edited body would be stale: B2HIP_ERR_INVALID, never a silent wrong state."""
    L = b2hip.lib()
    (lowest, _), calls, rc = drop_on_platform(L, 1, edit=lambda w, box: w.set_bullet(box, True))
    assert rc == 0 and calls > 0 and lowest > 0.1  # (an edit alone is fine)
    _, calls, rc = drop_on_platform(L, 0, edit=lambda w, box: w.set_bullet(box, True))
    assert rc != 0 and b"PreSolve" in L.b2hip_last_error()


def forced_box_edited_from_the_sub_step(L):
    rows = []
    # (the edit: b2Body::SetAwake(true) on the awake, moving box - it changes nothing in either backend whenever it is applied,
    #  so the two can only differ through the ROW the edit lands on. An edit with an effect of its own - SetBullet - acts inside
    #  the sub-step on the reference and between the steps here: include/b2hip.h, b2hip_toi_callback)
    (lowest, _), calls, rc = drop_on_platform(L, 1, edit=lambda w, box: w.set_awake(box, True), force=(0.25, 0.0), rows=rows)
    assert rc == 0 and calls > 0 and lowest > 0.1
    return np.array(rows)


def test_oracle_abi_forced_box_edited_from_the_sub_step():
    rows = forced_box_edited_from_the_sub_step(b2hip.load(bh.ORACLE_LIB, optional_ok=True))
    assert (np.diff(rows[:, 0]) > 0).all(), "the sideways force must move the box steadily"


@pytest.mark.gpu
def test_device_edit_from_a_toi_presolve_lands_on_the_fresh_row():
    """a body the host touches every frame (apply_force: its host row is 'current' for the epoch of the last
    read-back) and that a PreSolve called from a TOI sub-step edits: the edit must land on the row of THIS step's read-back,
    not on the pre-step row - which the next upload would have written back, the box jumping back one step."""
    a = forced_box_edited_from_the_sub_step(b2hip.lib())
    b = forced_box_edited_from_the_sub_step(b2hip.load(bh.ORACLE_LIB, optional_ok=True))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "first difference at step %d" % int(np.nonzero((a != b).any(axis=1))[0][0])


# ---- b2World::ShiftOrigin through the bare ABI: the world-space anchors joints keep (b2MouseJoint / b2PulleyJoint::ShiftOrigin) ----
def shifted_joints(L, shift_at, steps=90):
    w = b2hip.World(gravity=(0.0, -10.0), library=L)
    ground = w.create_body(b2hip.STATIC, position=(0.0, 0.0))
    w.create_fixture(ground, b2hip.box_shape(30.0, 0.5))
    a = w.create_body(b2hip.DYNAMIC, position=(-3.0, 4.0))
    w.create_fixture(a, b2hip.box_shape(0.5, 0.5), density=1.0)
    b = w.create_body(b2hip.DYNAMIC, position=(3.0, 5.0))
    w.create_fixture(b, b2hip.box_shape(0.5, 0.5), density=3.0)
    w.create_pulley_joint(a, b, (-3.0, 9.0), (3.0, 9.0), anchor_a=(0.0, 0.5), anchor_b=(0.0, 0.5), length_a=4.5, length_b=3.5, ratio=1.5)
    c = w.create_body(b2hip.DYNAMIC, position=(8.0, 3.0))
    w.create_fixture(c, b2hip.circle_shape(0.4), density=1.0)
    w.create_mouse_joint(ground, c, (8.0, 3.0), 500.0)
    total = np.zeros(2, np.float32)
    rows = []
    for s in range(steps):
        if s in shift_at:
            o = np.float32(shift_at[s])
            w.shift_origin(float(o[0]), float(o[1]))
            total += o
        w.step()
        rows.append(w.bodies8().copy())
    w.close()
    return np.array(rows), total


def check_shift(L):
    plain, _ = shifted_joints(L, {})
    moved, total = shifted_joints(L, {30: (10.0, -4.0), 60: (-2.5, 1.25)})
    assert np.array_equal(plain[:30], moved[:30])
    # velocities do not notice the shift (the joints' anchors moved with the bodies), positions differ by it to rounding
    assert np.allclose(moved[-1, :, 3:6], plain[-1, :, 3:6], atol=2e-3), "the joints' world anchors did not move with the origin"
    assert np.allclose(moved[-1, :, :2] + total, plain[-1, :, :2], atol=2e-3)
    return moved


def test_oracle_abi_shift_origin_moves_joint_anchors():
    check_shift(b2hip.load(bh.ORACLE_LIB, optional_ok=True))


@pytest.mark.gpu
def test_device_abi_shift_origin_matches_the_oracle():
    a = check_shift(b2hip.lib())
    b = check_shift(b2hip.load(bh.ORACLE_LIB, optional_ok=True))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
