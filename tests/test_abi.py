"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/b2hip.h
declares, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so")
HDR = os.path.join(ROOT, "include", "b2hip.h")


def declared_symbols():
    text = open(HDR).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(b2hip_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_step_path():
    syms = declared_symbols()
    for must in ("b2hip_world_create", "b2hip_step", "b2hip_collide", "b2hip_solve", "b2hip_sync_fixtures",
                 "b2hip_find_new_contacts", "b2hip_get_body_states", "b2hip_get_contacts", "b2hip_get_profile"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail("libb2hip.so missing: run __graft_entry__.build()")
    lib = C.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert missing == []


def test_no_cpu_fallback_without_device():
    """Without a GPU, world creation must fail loudly with B2HIP_ERR_NO_DEVICE (never silently step on the CPU)."""
    import b2hip
    L = b2hip.lib()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    d = b2hip.WorldDef(0.0, -10.0, 1, 1, 0, 0, 1, -1)
    p = C.c_void_p()
    rc = L.b2hip_world_create(C.byref(d), C.byref(p))
    if has_gpu:
        assert rc == 0
        L.b2hip_world_destroy(p)
    else:
        assert rc == -3, "expected B2HIP_ERR_NO_DEVICE, got %d" % rc
        assert b"no CPU fallback" in L.b2hip_last_error()


def test_product_does_not_link_the_oracle():
    """The shipped libraries must not depend on anything under oracle/."""
    import subprocess
    for so in ("libb2hip.so", "libb2amd_harness.so"):
        path = os.path.join(ROOT, "box2d-mt_amd", so)
        if not os.path.exists(path):
            pytest.fail(so + " missing")
        out = subprocess.check_output(["ldd", path]).decode()
        assert "oracle" not in out
        syms = subprocess.check_output(["nm", "-D", path]).decode()
        assert "b2o_" not in syms


@pytest.mark.gpu
def test_sub_stepping_solves_one_toi_event_per_call():
    """b2World::SetSubStepping (b2World.h:183; b2World.cpp:1082-1086, 1668): with the flag on, a step call that finds a TOI
    event solves that one event and leaves the step open; the next call runs no island solve - a body in free fall far away
    does not move during it - and a call that finds no event left closes the step, after which bodies are integrated again.
    (Parity with the oracle and the reference build: tests/test_events.py.)"""
    import b2hip
    w = b2hip.World(continuous=True)
    g = w.create_body(b2hip.STATIC, (0.0, 0.0))
    w.create_fixture(g, b2hip.box_shape(20.0, 0.05))
    fast = w.create_body(b2hip.DYNAMIC, (0.0, 3.0), velocity=(0.0, -60.0))      # reaches the platform through a TOI event
    w.create_fixture(fast, b2hip.box_shape(0.1, 0.1), density=1.0)
    free = w.create_body(b2hip.DYNAMIC, (50.0, 100.0))                           # far from everything: only Solve moves it
    w.create_fixture(free, b2hip.box_shape(0.1, 0.1), density=1.0)
    w.set_flags(continuous=True, sub_stepping=True)
    heights, continued = [], 0
    for _ in range(12):
        w.step()
        heights.append(float(w.body_states()["py"][free]))
        if len(heights) > 1 and heights[-1] == heights[-2]:
            continued += 1
    assert continued > 0, "no call ever continued an open step (the free body fell in every call): %s" % heights
    assert heights[-1] < heights[0], "the step never completed again"
    assert float(w.body_states()["py"][fast]) > 0.1, "the fast box went through the platform"
    w.close()


@pytest.mark.gpu
def test_an_open_sub_stepped_step_survives_a_snapshot():
    """m_stepComplete travels with the world (b2hip_save_snapshot): a world saved between two calls of one sub-stepped step
    continues that step after loading, call for call like the world it was taken from."""
    import b2hip
    w = b2hip.World(continuous=True)
    g = w.create_body(b2hip.STATIC, (0.0, 0.0))
    w.create_fixture(g, b2hip.box_shape(20.0, 0.05))
    for k in range(6):
        b = w.create_body(b2hip.DYNAMIC, (-5.0 + 2.0 * k, 3.0 + 0.1 * k), velocity=(1.0 * k, -60.0))
        w.create_fixture(b, b2hip.box_shape(0.1, 0.1), density=1.0)
    w.set_flags(continuous=True, sub_stepping=True)
    w.step()
    w.step()  # (six impacts pending after the first call: this one continues the step)
    w2 = b2hip.World.from_snapshot(w.save_snapshot())
    for k in range(14):
        w.step()
        w2.step()
        assert w.body_states().tobytes() == w2.body_states().tobytes(), "call %d after the snapshot" % k
    w.close()
    w2.close()


def test_no_wide_store_has_its_data_registers_overwritten_too_soon():
    """The gfx950 code of the built library: no dwordx3/x4 store with a VALU write to its data registers inside the next
    two issue slots. The compiler keeps that distance for its own stores, not for the sc1 hand-over stores written as
    asm statements (they carry an s_nop 1) - one such pair in k_solve_blocks made the bench scene's runs differ, round 5
    (tools/asm_store_hazard.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_store_hazard as hz
    if not os.path.exists(hz.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not present")
    if not os.path.exists(LIB):
        pytest.fail("libb2hip.so missing: run __graft_entry__.build()")
    stores, found = hz.scan(hz.disassemble(LIB))
    assert stores > 0
    assert found == []
