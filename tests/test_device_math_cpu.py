"""CPU tests of the product's device math headers (box2d-mt_amd/csrc/b2d_*.h compiled for the host by
tests/probe/host_probe.cpp): manifolds, shape AABBs and sin/cos, bitwise against golden vectors."""
import ctypes as C
import os

import numpy as np

import probe_util as pu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
fp = C.POINTER(C.c_float)


def test_device_collide_headers_match_golden():
    P = pu.build_probe()
    v = np.load(os.path.join(GOLD, "collide_vectors.npz"))
    bad = 0
    touching = 0
    for sa, sb, xa, xb, want in zip(v["shapeA"], v["shapeB"], v["xfA"], v["xfB"], v["manifold"]):
        out = np.zeros(16, np.float32)
        sa, sb = np.ascontiguousarray(sa), np.ascontiguousarray(sb)
        xa, xb = np.ascontiguousarray(xa), np.ascontiguousarray(xb)
        P.probe_collide(sa.ctypes.data_as(C.c_void_p), xa.ctypes.data_as(fp), sb.ctypes.data_as(C.c_void_p),
                        xb.ctypes.data_as(fp), out.ctypes.data_as(fp))
        bad += not np.array_equal(out.view(np.uint32), want.view(np.uint32))
        touching += want[1] > 0
    assert touching > 300
    assert bad == 0


def test_device_sincos_matches_golden():
    P = pu.build_probe()
    v = np.load(os.path.join(GOLD, "sincos_vectors.npz"))
    a = np.ascontiguousarray(v["angle"])
    s = np.empty_like(a)
    c = np.empty_like(a)
    P.probe_sincos(a.size, a.ctypes.data_as(fp), s.ctypes.data_as(fp), c.ctypes.data_as(fp))
    assert np.array_equal(s.view(np.uint32), v["sin"].view(np.uint32))
    assert np.array_equal(c.view(np.uint32), v["cos"].view(np.uint32))


def test_device_sincos_matches_libm_dense_sample():
    """b2Rot::Set calls libm: the restated glibc algorithm must agree with this machine's libm bit for bit
    (exhaustive over all 2^32 inputs when written; a 1/4099 strided sweep of the whole range here)."""
    P = pu.build_probe()
    P.probe_sincos_vs_libm.restype = C.c_long
    bad = P.probe_sincos_vs_libm(C.c_uint(0), C.c_uint(0xFFFFFFFF), C.c_uint(4099))
    assert bad == 0


def _toi_vectors():
    return np.load(os.path.join(GOLD, "toi_vectors.npz"))


def test_device_distance_header_matches_golden():
    """b2d_toi.h b2dDistance (GJK) against the reference's b2Distance outputs."""
    P = pu.build_probe()
    v = _toi_vectors()
    bad = 0
    for i in range(len(v["d_out"])):
        out = np.zeros(6, np.float32)
        a = np.ascontiguousarray(v["d_vertsA"][i]); b = np.ascontiguousarray(v["d_vertsB"][i])
        xa = np.ascontiguousarray(v["d_xfA"][i]); xb = np.ascontiguousarray(v["d_xfB"][i])
        P.probe_distance(int(v["d_countA"][i]), a.ctypes.data_as(fp), C.c_float(v["d_radiusA"][i]), xa.ctypes.data_as(fp),
                         int(v["d_countB"][i]), b.ctypes.data_as(fp), C.c_float(v["d_radiusB"][i]), xb.ctypes.data_as(fp),
                         int(v["d_useRadii"][i]), out.ctypes.data_as(fp))
        bad += not np.array_equal(out.view(np.uint32), v["d_out"][i].view(np.uint32))
    assert bad == 0


def test_device_toi_header_matches_golden():
    """b2d_toi.h b2dTimeOfImpact against the reference's b2TimeOfImpact outputs (state and t, bitwise)."""
    P = pu.build_probe()
    v = _toi_vectors()
    bad = 0
    for i in range(len(v["t_out"])):
        out = np.zeros(2, np.float32)
        a = np.ascontiguousarray(v["t_vertsA"][i]); b = np.ascontiguousarray(v["t_vertsB"][i])
        sa = np.ascontiguousarray(v["t_sweepA"][i]); sb = np.ascontiguousarray(v["t_sweepB"][i])
        P.probe_toi(int(v["t_countA"][i]), a.ctypes.data_as(fp), C.c_float(v["t_radiusA"][i]), sa.ctypes.data_as(fp),
                    int(v["t_countB"][i]), b.ctypes.data_as(fp), C.c_float(v["t_radiusB"][i]), sb.ctypes.data_as(fp),
                    C.c_float(1.0), out.ctypes.data_as(fp))
        bad += not np.array_equal(out.view(np.uint32), v["t_out"][i].view(np.uint32))
    assert bad == 0


def test_own_id_block_is_the_exact_remainder():
    """ownIdBlock (b2d_math.h): hash(body) mod blocks + 1 by way of a float quotient with both corrections - the compiler's own
    24-bit expansion of `%` on the GPU returned 0xffffff for 0xc1f9f3 % 11 inside one kernel (round 5: a body of a jointed
    pile was home in no block). Checked against the integer remainder for every block count up to 1 024 over a stride of body
    ids and every id of the first 2 M whose hash lies in the top sixteenth of its 24 bits."""
    P = pu.build_probe()
    P.probe_own_id_block_check.restype = C.c_long
    assert P.probe_own_id_block_check(C.c_int(1024), C.c_uint(2000000), C.c_uint(257)) == 0

