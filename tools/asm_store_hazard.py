"""Wide stores whose data registers are overwritten too soon, in the gfx950 code of libb2hip.so.

A VMEM store of more than 8 bytes reads its data VGPRs late: on gfx940 and later a VALU instruction that writes one of
them must be at least two instructions behind the store. The compiler keeps that distance for the stores it emits; a
store inside an `asm volatile` statement (the sc1 hand-over stores of b2d_handover.h, b2d_scan.h, b2d_world.h) is invisible
to its hazard pass, so those carry their own `s_nop 1`. This walks the disassembly of the built library and reports every
dwordx3 / dwordx4 store followed within two issue slots by a VALU write to its data registers (fall-through only; a taken
branch costs more than two slots). Round 5: k_solve_blocks had one such pair and the bench scene stopped repeating bit
for bit, four runs in ten.

usage: python tools/asm_store_hazard.py [path/to/libb2hip.so]      (exit code 1 if anything was found)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WAIT_STATES = 2

STORE = re.compile(r"^(global|flat|buffer|scratch)_store_dwordx[34]\b")


def vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return int(m.group(1)), int(m.group(1))
    return None


def disassemble(lib):
    """llvm-objdump drops the bundle's members next to its input: work on a copy in a scratch directory."""
    tmp = tempfile.mkdtemp(prefix="b2hip_dis_")
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        members = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not members:
            raise RuntimeError("no gfx950 code object in " + lib)
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, members[0])], check=True, stdout=subprocess.PIPE, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def data_operand(mnemonic, ops):
    """The registers a store sends: global/flat/scratch `addr, data, ...`, buffer `data, vaddr, ...`."""
    if mnemonic.startswith("buffer"):
        return vregs(ops[0])
    return vregs(ops[1]) if len(ops) > 1 else None


def scan(text):
    found, stores = [], 0
    lines = text.split("\n")
    fn = "?"
    for i, raw in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", raw)
        if m:
            fn = m.group(1)
            continue
        t = raw.split("//")[0].strip()
        if not STORE.match(t):
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")]
        data = data_operand(parts[0], ops)
        if data is None:
            continue
        stores += 1
        slots, j = 0, i + 1
        while slots < WAIT_STATES and j < len(lines):
            u = lines[j].split("//")[0].strip()
            j += 1
            if not u or u.endswith(":") or re.match(r"^[0-9a-f]+ <", u):
                continue
            mn = u.split()[0]
            if mn in ("s_branch", "s_endpgm", "s_setpc_b64"):
                break
            if mn == "s_nop":
                slots += int(u.split()[1], 0) + 1
                continue
            if mn.startswith("v_") and not mn.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
                dst = vregs(u.split(None, 1)[1].split(",")[0].strip()) if len(u.split(None, 1)) > 1 else None
                if dst and not (dst[1] < data[0] or dst[0] > data[1]):
                    found.append((fn, t, u, slots))
            slots += 1
    return stores, found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so")
    stores, found = scan(disassemble(lib))
    print("%d wide stores, %d with a VALU write to their data registers less than %d slots behind" % (stores, len(found), WAIT_STATES))
    for fn, st, wr, slots in found:
        print("  %s: `%s` then, %d slot(s) later, `%s`" % (fn, st, slots, wr))
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
