"""Registers, LDS, scratch and spills of every kernel in the gfx950 code object of libb2hip.so (from its metadata note), and
the waves per SIMD those registers allow (512 VGPRs per SIMD lane on gfx950, unified with the AGPRs; at most 8 waves).
CPU only.   usage: python tools/kernel_resources.py [path/to/libb2hip.so] [name filter]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def notes(lib):
    tmp = tempfile.mkdtemp(prefix="b2hip_res_")
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        member = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        return subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, member)], check=True, stdout=subprocess.PIPE, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", re.sub(r"^void ", "", o)) for o in out]


def kernels(text):
    ks, cur = [], None
    for line in text.split("\n"):
        if re.match(r"^\s+- \.agpr_count:", line):
            cur = {}
            ks.append(cur)
        m = re.match(r"^\s+(?:- )?\.(agpr_count|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|max_flat_workgroup_size|kernarg_segment_size|name):\s+(\S+)", line)
        if m and cur is not None and m.group(1) not in cur:
            cur[m.group(1)] = m.group(2)
    return [k for k in ks if "name" in k and "vgpr_count" in k]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "box2d-mt_amd", "libb2hip.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    ks = kernels(notes(lib))
    names = demangle([k["name"] for k in ks])
    rows = []
    for k, n in zip(ks, names):
        if flt and flt not in n:
            continue
        v = int(k["vgpr_count"]) + int(k.get("agpr_count", 0))
        alloc = max(8, (v + 7) // 8 * 8)
        waves = min(8, 512 // alloc)
        rows.append((n, int(k["vgpr_count"]), int(k.get("agpr_count", 0)), int(k["sgpr_count"]), int(k["group_segment_fixed_size"]), int(k["private_segment_fixed_size"]),
                     int(k.get("vgpr_spill_count", 0)), int(k["max_flat_workgroup_size"]), int(k["kernarg_segment_size"]), waves))
    rows.sort(key=lambda r: (-r[5], -r[1], r[0]))
    print("%d kernels in %s" % (len(rows), os.path.relpath(lib, ROOT)))
    print("%-58s %5s %5s %5s %8s %8s %6s %6s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "LDS B", "scratch", "spills", "max wg", "kernarg", "waves"))
    for r in rows:
        print("%-58s %5d %5d %5d %8d %8d %6d %6d %8d %6d" % ((r[0][:58],) + r[1:]))
    spilled = [r[0] for r in rows if r[6] > 0]
    print("kernels with VGPR spills: %d%s" % (len(spilled), (" (" + ", ".join(spilled[:12]) + ")") if spilled else ""))


if __name__ == "__main__":
    main()
