"""The host-side API of the drop-in layer that user code calls directly (SURVEY.md section 8b symbol list): b2CollidePolygons &
co., b2Distance, b2TimeOfImpact, b2ShapeCast, b2AABB::RayCast, b2DynamicTree, the tree statistics of b2World, and
b2Joint::GetReactionForce / GetReactionTorque. All CPU tests: the drop-in layer (linked against the oracle's ABI shim; the
host API itself is the same object code the product links) against the REFERENCE BUILD on the same inputs, bit for bit
wherever both compute the same thing, and against brute force for the tree (the shadow tree is an own implementation).

Reference: b2Collision.h:229-256, b2Collision.cpp:133-198, b2Distance.h, b2Distance.cpp:444-745, b2TimeOfImpact.cpp:253-486,
b2DynamicTree.h:52-130, b2World.h:199-206, b2Joint.h:129-133.
"""
import ctypes as C
import os

import numpy as np
import pytest

import b2harness as bh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
_fp = C.POINTER(C.c_float)


def fptr(a):
    return a.ctypes.data_as(_fp)


def rpoly(rng):
    n = rng.integers(3, 9)
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = rng.uniform(0.3, 1.0)
    return [(r * np.cos(a) * rng.uniform(0.7, 1), r * np.sin(a) * rng.uniform(0.7, 1)) for a in ang]


def hull(h, pts):
    o = h.polygon(pts)
    n = int(o[0])
    return o[1:1 + 2 * n].reshape(n, 2).copy()


def test_host_narrow_phase_functions_match_the_reference_build(oracle, ref):
    """b2CollidePolygons / PolygonAndCircle / Circles / EdgeAndPolygon / EdgeAndCircle of the drop-in API (the CPU build of the
    collide kernel's manifold code) on random pairs: manifold type, points and feature ids equal the reference's bitwise."""
    rng = np.random.default_rng(77)
    touching = 0
    for i in range(600):
        xa = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-7, 7)]
        xb = [xa[0] + rng.uniform(-1.6, 1.6), xa[1] + rng.uniform(-1.6, 1.6), rng.uniform(-7, 7)]
        k = i % 5
        if k == 0:
            args = (("verts", rpoly(rng)), xa, ("verts", rpoly(rng)), xb)
            a, b = oracle.collide_polygons(*args), ref.collide_polygons(*args)
        elif k == 1:
            args = (("verts", rpoly(rng)), xa, [rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)], xb)
            a, b = oracle.collide_polygon_circle(*args), ref.collide_polygon_circle(*args)
        elif k == 2:
            args = ([rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)], xa, [rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)], xb)
            a, b = oracle.collide_circles(*args), ref.collide_circles(*args)
        else:
            e = [-1, rng.uniform(-.2, .2), 1, rng.uniform(-.2, .2), rng.integers(0, 2), -2, rng.uniform(-1, 1), rng.integers(0, 2), 2, rng.uniform(-1, 1)]
            if k == 3:
                args = (e, xa, ("verts", rpoly(rng)), xb)
                a, b = oracle.collide_edge_polygon(*args), ref.collide_edge_polygon(*args)
            else:
                args = (e, xa, [rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.1, .8)], xb)
                a, b = oracle.collide_edge_circle(*args), ref.collide_edge_circle(*args)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "pair %d (kind %d)" % (i, k)
        touching += b[1] > 0
    assert touching > 100


def proxies(rng, h):
    kind = rng.integers(0, 3)
    if kind == 0:
        return np.array([[rng.uniform(-.2, .2), rng.uniform(-.2, .2)]], np.float32), float(np.float32(rng.uniform(.1, .8)))
    if kind == 1:
        return np.array([[-rng.uniform(.5, 3), rng.uniform(-.2, .2)], [rng.uniform(.5, 3), rng.uniform(-.2, .2)]], np.float32), 0.01
    return hull(h, rpoly(rng)), 0.01


def test_host_distance_toi_and_shape_cast_match_the_reference_build(oracle, ref):
    """b2Distance, b2TimeOfImpact (the GJK / conservative advancement of the TOI kernels, CPU build) and b2ShapeCast of the
    drop-in API against the reference build, bitwise, on random proxies (circles, segments, polygons)."""
    rng = np.random.default_rng(31)
    hits = 0
    for L in (oracle.lib, ref.lib):
        L.b2h_probe_shape_cast.argtypes = [C.c_int, _fp, C.c_float, _fp, C.c_int, _fp, C.c_float, _fp, C.c_float, C.c_float, _fp]
    for i in range(500):
        va, ra = proxies(rng, ref)
        vb, rb = proxies(rng, ref)
        xa = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-3, 3)], np.float32)
        xb = np.array([xa[0] + rng.uniform(-4, 4), xa[1] + rng.uniform(-4, 4), rng.uniform(-3, 3)], np.float32)
        for use_radii in (False, True):
            a = oracle.distance(va, ra, xa, vb, rb, xb, use_radii)
            b = ref.distance(va, ra, xa, vb, rb, xb, use_radii)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "b2Distance case %d" % i
        sa = [0, 0, xa[0], xa[1], xa[0] + rng.uniform(-2, 2), xa[1] + rng.uniform(-2, 2), xa[2], xa[2] + rng.uniform(-1, 1), 0.0]
        sb = [0, 0, xb[0], xb[1], xb[0] + rng.uniform(-6, 6), xb[1] + rng.uniform(-6, 6), xb[2], xb[2] + rng.uniform(-1, 1), 0.0]
        a, b = oracle.toi(va, ra, sa, vb, rb, sb), ref.toi(va, ra, sa, vb, rb, sb)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "b2TimeOfImpact case %d" % i
        t = rng.uniform(-6, 6, 2)
        outs = []
        for h in (oracle, ref):
            o = np.zeros(7, np.float32)
            fa, fb = np.ascontiguousarray(va).reshape(-1), np.ascontiguousarray(vb).reshape(-1)
            h.lib.b2h_probe_shape_cast(fa.size // 2, fptr(fa), ra, fptr(xa), fb.size // 2, fptr(fb), rb, fptr(xb), float(t[0]), float(t[1]), fptr(o))
            outs.append(o)
        assert outs[0][0] == outs[1][0], "b2ShapeCast hit / miss, case %d" % i
        if outs[1][0]:
            hits += 1
            assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), "b2ShapeCast case %d: %s vs %s" % (i, outs[0], outs[1])
    assert hits > 60


def test_aabb_raycast_matches_the_reference_build(oracle, ref):
    rng = np.random.default_rng(9)
    hits = 0
    for i in range(2000):
        lo = rng.uniform(-3, 3, 2)
        box = np.array([lo[0], lo[1], lo[0] + rng.uniform(0.1, 4), lo[1] + rng.uniform(0.1, 4)], np.float32)
        ray = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(0.2, 1.0)], np.float32)
        if i % 7 == 0:
            ray[2] = ray[0]  # parallel to an axis
        oa, ob = np.zeros(3, np.float32), np.zeros(3, np.float32)
        ha = oracle.lib.b2h_probe_aabb_raycast(fptr(box), fptr(ray), fptr(oa))
        hb = ref.lib.b2h_probe_aabb_raycast(fptr(box), fptr(ray), fptr(ob))
        assert ha == hb
        if hb:
            hits += 1
            assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32))
    assert hits > 100


def test_dynamic_tree_against_brute_force(oracle):
    """b2DynamicTree of the drop-in API (own implementation): random create / move / destroy, queries and ray casts against
    brute force over the stored boxes, structural validation after every batch; the tree stays balanced."""
    out = np.zeros(3, np.float32)
    oracle.lib.b2h_probe_dynamic_tree.argtypes = [C.c_uint, C.c_int, C.c_int, _fp]
    for seed, count, ops in ((1, 40, 4000), (2, 400, 12000), (3, 2000, 20000)):
        bad = oracle.lib.b2h_probe_dynamic_tree(seed, count, ops, fptr(out))
        assert bad == 0
        height, balance, ratio = out
        assert height <= 2.0 * np.log2(count) + 4, "height %d for up to %d leaves" % (height, count)
        assert balance <= 2 + np.log2(count) and np.isfinite(ratio) and ratio > 1.0


def test_world_tree_statistics_are_of_the_reference_kind(oracle, ref):
    """GetTreeHeight / Balance / Quality and GetProxyCount: the proxy count equals the reference's; height, balance and
    quality describe this repo's tree over the same fat AABBs - same kind, stated band (b2World.h:199-206)."""
    for h in (oracle, ref):
        h.lib.b2h_tree_stats.argtypes = [C.c_void_p, _fp]
    for scene, p0, p1 in ((bh.RAIN, 300, 0), (bh.PYRAMID, 20, 1), (bh.FIELD, 800, 0)):
        a, r = oracle.world(scene, p0, p1, seed=5), ref.world(scene, p0, p1, seed=5)
        for s in range(60):
            a.step(1)
            r.step(1)
            if s % 20 == 19:
                sa, sr = np.zeros(4, np.float32), np.zeros(4, np.float32)
                a.L.b2h_tree_stats(a.ptr, fptr(sa))
                r.L.b2h_tree_stats(r.ptr, fptr(sr))
                assert sa[3] == sr[3], "proxy count"
                n = max(sr[3], 2)
                assert np.ceil(np.log2(n)) <= sa[0] <= sr[0] + 4, "tree height %d (reference %d, %d proxies)" % (sa[0], sr[0], n)
                assert sa[1] <= sr[1] + 3
                assert sr[2] / 3.0 <= sa[2] <= 3.0 * sr[2], "quality %g (reference %g)" % (sa[2], sr[2])
        a.close()
        r.close()


@pytest.mark.parametrize("scene,p0,p1,seed", [(bh.MACHINES, 40, 4, 3), (bh.VEHICLES, 40, 3, 3), (bh.ROPES, 30, 8, 9), (bh.TUMBLER, 6, 0, 1)])
def test_joint_reaction_forces_match_the_reference_build(oracle, ref, scene, p0, p1, seed):
    """b2Joint::GetReactionForce / GetReactionTorque(inv_dt) of every joint after every step (all eleven joint types over the
    three joint scenes), bitwise against the reference build."""
    for h in (oracle, ref):
        h.lib.b2h_joint_reactions.argtypes = [C.c_void_p, C.c_float, C.c_int, _fp]
    a, r = oracle.world(scene, p0, p1, seed=seed), ref.world(scene, p0, p1, seed=seed)
    oa, orr = np.zeros((256, 3), np.float32), np.zeros((256, 3), np.float32)
    busy = 0
    for s in range(120):
        a.step(1)
        r.step(1)
        na = a.L.b2h_joint_reactions(a.ptr, 60.0, 256, fptr(oa))
        nr = r.L.b2h_joint_reactions(r.ptr, 60.0, 256, fptr(orr))
        assert na == nr and na > 0
        assert np.array_equal(oa[:na].view(np.uint32), orr[:nr].view(np.uint32)), "step %d: joint %s" % (s, np.nonzero((oa[:na] != orr[:nr]).any(axis=1))[0][:5])
        busy += int(np.count_nonzero(orr[:nr]))
    assert busy > 0
    a.close()
    r.close()


@pytest.mark.parametrize("scene,p0,p1,seed", [(bh.MACHINES, 40, 4, 3), (bh.VEHICLES, 40, 3, 3), (bh.ROPES, 30, 8, 9), (bh.LIFECYCLE, 36, 0, 2)])
def test_joint_anchors_match_the_reference_build(oracle, ref, scene, p0, p1, seed):
    """b2Joint::GetAnchorA / GetAnchorB (pure virtual in the reference, one pair per joint type - the mouse joint's target, the
    motor joint's body origins and the gear joint's borrowed anchors among them) of every joint every tenth step, bitwise."""
    for h in (oracle, ref):
        h.lib.b2h_joint_anchors.argtypes = [C.c_void_p, C.c_int, _fp]
    a, r = oracle.world(scene, p0, p1, seed=seed), ref.world(scene, p0, p1, seed=seed)
    oa, orr = np.zeros((256, 4), np.float32), np.zeros((256, 4), np.float32)
    for s in range(100):
        a.step(1)
        r.step(1)
        if s % 10 != 9:
            continue
        na = a.L.b2h_joint_anchors(a.ptr, 256, fptr(oa))
        nr = r.L.b2h_joint_anchors(r.ptr, 256, fptr(orr))
        assert na == nr and na > 0
        assert np.array_equal(oa[:na].view(np.uint32), orr[:nr].view(np.uint32)), "step %d: joint %s" % (s, np.nonzero((oa[:na] != orr[:nr]).any(axis=1))[0][:5])
    a.close()
    r.close()


@pytest.mark.parametrize("scene,p0,p1,seed", [(bh.MACHINES, 40, 4, 3), (bh.VEHICLES, 40, 3, 3), (bh.ROPES, 30, 8, 9), (bh.LIFECYCLE, 36, 0, 2)])
def test_body_joint_lists_match_the_reference_build(oracle, ref, scene, p0, p1, seed):
    """b2Body::GetJointList (b2Body.h:426-429): every body's joint edges, newest first, with the body on the other side - after
    the scene is built and again after joints and bodies have been destroyed (the life-cycle scene destroys bodies with joints
    on them, the vehicles scene lets its mouse joint go)."""
    ip = C.POINTER(C.c_int)
    for h in (oracle, ref):
        h.lib.b2h_body_joint_lists.argtypes = [C.c_void_p, C.c_int, ip]
    a, r = oracle.world(scene, p0, p1, seed=seed), ref.world(scene, p0, p1, seed=seed)
    oa, orr = np.zeros((512, 5), np.int32), np.zeros((512, 5), np.int32)
    for s in range(200):
        if s % 50 == 0:
            na = a.L.b2h_body_joint_lists(a.ptr, 512, oa.ctypes.data_as(ip))
            nr = r.L.b2h_body_joint_lists(r.ptr, 512, orr.ctypes.data_as(ip))
            assert na == nr and na > 0
            assert np.array_equal(oa[:na], orr[:nr]), "step %d: body %s" % (s, np.nonzero((oa[:na] != orr[:nr]).any(axis=1))[0][:5])
            assert oa[:na, 0].sum() > 0
        a.step(1)
        r.step(1)
    a.close()
    r.close()


RETUNE_CASES = [(bh.MACHINES, 40, 4, 3), (bh.VEHICLES, 40, 3, 3), (bh.ROPES, 30, 8, 9)]


def run_retuned(a, b, what):
    for h in (a, b):
        h.L.b2h_retune_joints.argtypes = [C.c_void_p, C.c_int]
        h.L.b2h_wheel_states.argtypes = [C.c_void_p, C.c_int, _fp]
    wa, wb = np.zeros((64, 4), np.float32), np.zeros((64, 4), np.float32)
    ip = C.POINTER(C.c_int)
    for h in (a, b):
        h.L.b2h_rope_states.argtypes = [C.c_void_p, C.c_int, ip]
    ra, rb = np.zeros(64, np.int32), np.zeros(64, np.int32)
    retuned = 0
    for s in range(160):
        if s in (20, 60, 100):
            na, nb = a.L.b2h_retune_joints(a.ptr, s // 20), b.L.b2h_retune_joints(b.ptr, s // 20)
            assert na == nb
            retuned += na
        a.step(1)
        b.step(1)
        assert np.array_equal(a.bodies().view(np.uint32), b.bodies().view(np.uint32)), "%s: body states differ at step %d" % (what, s)
        if s % 10 == 9:
            ka, kb = a.L.b2h_wheel_states(a.ptr, 64, fptr(wa)), b.L.b2h_wheel_states(b.ptr, 64, fptr(wb))
            assert ka == kb and np.array_equal(wa[:ka].view(np.uint32), wb[:kb].view(np.uint32)), "%s: wheel joint getters at step %d" % (what, s)
            ka, kb = a.L.b2h_rope_states(a.ptr, 64, ra.ctypes.data_as(ip)), b.L.b2h_rope_states(b.ptr, 64, rb.ctypes.data_as(ip))
            assert ka == kb and np.array_equal(ra[:ka], rb[:kb]), "%s: rope joint limit states at step %d" % (what, s)
    assert retuned > 0
    return a.bodies()


@pytest.mark.parametrize("scene,p0,p1,seed", RETUNE_CASES)
def test_joint_scalar_setters_match_the_reference_build(oracle, ref, scene, p0, p1, seed):
    """b2DistanceJoint::SetLength / SetFrequency / SetDampingRatio, b2FrictionJoint::SetMaxForce / SetMaxTorque, b2GearJoint::SetRatio,
    b2MotorJoint::SetMaxForce / SetMaxTorque / SetCorrectionFactor, b2MouseJoint::SetMaxForce / SetFrequency / SetDampingRatio,
    b2RopeJoint::SetMaxLength, b2WeldJoint::SetFrequency / SetDampingRatio applied three times to every joint of the three joint
    scenes (box2d-mt_amd/harness/harness.cpp: b2h_retune_joints), and the wheel joints' GetJointTranslation / LinearSpeed /
    Angle / AngularSpeed: states and getters bitwise against the reference build; the retuning changes the motion."""
    a, r = oracle.world(scene, p0, p1, seed=seed), ref.world(scene, p0, p1, seed=seed)
    tuned = run_retuned(a, r, "oracle vs reference")
    plain = ref.world(scene, p0, p1, seed=seed)
    plain.step(160)
    assert not np.array_equal(tuned, plain.bodies())
    for w in (a, r, plain):
        w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene,p0,p1,seed", RETUNE_CASES)
def test_device_joint_scalar_setters_match_the_oracle(amd, oracle, scene, p0, p1, seed):
    a, b = amd.world(scene, p0, p1, seed=seed), oracle.world(scene, p0, p1, seed=seed)
    run_retuned(a, b, "device vs oracle")
    a.close()
    b.close()


# ---- b2World::SetDebugDraw / DrawDebugData (b2World.h:77, 120; b2World.cpp:1797-2045) -------------------------------------------
DRAW_CASES = [(bh.PYRAMID, 8, 1), (bh.VEHICLES, 10, 2), (bh.CHAINS, 30, 0), (bh.MACHINES, 10, 2), (bh.FIELD, 200, 20), (bh.ROPES, 10, 4)]


@pytest.mark.parametrize("scene,p0,p1", DRAW_CASES)
def test_debug_draw_matches_the_reference_build(oracle, ref, scene, p0, p1):
    """The drop-in layer's DrawDebugData against the reference's, through a b2Draw that counts: the same number of calls per
    primitive kind and the same checksum of every coordinate, radius and colour handed over, for every b2Draw flag - shapes
    (circles, polygons, edges, chains with ghost vertices) in their state colours, joints (distance, pulley, mouse, the
    generic three segments), fat AABBs (the device's table), centres of mass. Sleeping bodies included (step 200)."""
    a, b = ref.world(scene, p0, p1, seed=3), oracle.world(scene, p0, p1, seed=3)
    for steps in (40, 160):
        a.step(steps)
        b.step(steps)
        for flags in (0x01, 0x02, 0x04, 0x08, 0x10, 0x1f):
            assert a.debug_draw(flags) == b.debug_draw(flags), "flags 0x%x after %d steps" % (flags, steps)
    assert sum(a.debug_draw(0x1f)[0]) > 0
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene,p0,p1", DRAW_CASES[:4])
def test_device_debug_draw_matches_the_oracle_backed_layer(amd, oracle, monkeypatch, scene, p0, p1):
    monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    a, b = amd.world(scene, p0, p1, seed=3), oracle.world(scene, p0, p1, seed=3)
    a.step(60)
    b.step(60)
    for flags in (0x01, 0x02, 0x04, 0x10, 0x1f):
        assert a.debug_draw(flags) == b.debug_draw(flags), "flags 0x%x" % flags
    a.close()
    b.close()
