"""What the settled Tumbler's island looks like to the large-island solver: colour census, body degrees, the hub list (rows,
partners that occur twice, other hubs), fixed-point rounds and serial chunks per step, ms per step.
usage: python tools/gpu_tumbler_probe.py [n = 316] [settle = 400] [steps = 20]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as bh, b2hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 316
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
amd = bh.Harness(bh.AMD_LIB); L = b2hip.lib()
L.b2hip_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
w = amd.world(bh.TUMBLER, n, 0, flags=bh.F_SLEEP | bh.F_WARM)
dev = C.c_void_p(w.device_world())
t0 = time.time(); w.step(settle); print("settled %d steps in %.1f s" % (settle, time.time() - t0), flush=True)
per = []
for s in range(steps):
    t0 = time.perf_counter(); w.step(1); per.append(1e3 * (time.perf_counter() - t0))
    c = b2hip.Counters(); L.b2hip_get_counters(dev, C.byref(c))
    print("step %d: %.2f ms, L rows %d bodies %d, colours %d, hub rows %d, rounds %d, serial chunks %d" % (settle + s, per[-1], c.large_island_contacts, c.large_island_bodies, c.colors, c.hub_constraints, c.hub_fixpoint_rounds, c.hub_serial_chunks), flush=True)
print("mean %.3f ms p50 %.3f" % (np.mean(per), np.median(per)))
c = b2hip.Counters(); L.b2hip_get_counters(dev, C.byref(c))
cc = np.zeros(65, np.int32); L.b2hip_debug_read(dev, 12, 0, 65, cc.ctypes.data)
print("colour census:", [int(x) for x in cc[:64] if x], "hub group", int(cc[63]))
nb = w.body_count
deg = np.zeros(nb, np.int32); L.b2hip_debug_read(dev, 14, 0, nb, deg.ctypes.data)
print("degree histogram (bodies by solid touching contacts):", np.bincount(np.minimum(deg, 40)).tolist())
nh = c.hub_constraints
if nh > 0:
    hl = np.zeros(nh, np.int32); L.b2hip_debug_read(dev, 15, 0, nh, hl.ctypes.data)
    nrow = c.large_island_contacts
    ref = np.zeros((nrow, 4), np.int32); L.b2hip_debug_read(dev, 18, 0, nrow, ref.ctypes.data)
    r = ref[hl]
    a, b = r[:, 1], r[:, 2]
    ia = np.where(a >= 0, a, -(a + 1)); ib = np.where(b >= 0, b, -(b + 1))
    hubs = np.where(deg > 30)[0]
    print("hub bodies:", hubs.tolist(), "degrees", deg[hubs].tolist())
    hubA = np.isin(ia, hubs) & (a >= 0); hubB = np.isin(ib, hubs) & (b >= 0)
    print("hub rows %d: hub is A %d, hub is B %d, both %d, neither (serial orphans) %d" % (nh, int(hubA.sum()), int(hubB.sum()), int((hubA & hubB).sum()), int((~hubA & ~hubB).sum())))
    partner = np.where(hubA, ib, ia)
    pdyn = np.where(hubA, b >= 0, a >= 0)
    u, cnt = np.unique(partner[pdyn], return_counts=True)
    print("dynamic partners %d distinct %d; partners occurring twice or more: %d (rows %d)" % (int(pdyn.sum()), len(u), int((cnt > 1).sum()), int(cnt[cnt > 1].sum())))
    # per chunk of 64: would the chunk be "simple" (one hub, no duplicate partner)?
    bad = 0
    for k in range(0, nh, 64):
        p = partner[k:k + 64][pdyn[k:k + 64]]
        if len(np.unique(p)) != len(p): bad += 1
    print("chunks %d, with a duplicate partner inside %d" % ((nh + 63) // 64, bad))
if os.environ.get("B2HIP_SWEEP_STAMPS"):
    st = (C.c_int * 6)()
    L.b2hip_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    acc = np.zeros(5)
    for _ in range(10):
        w.step(1); L.b2hip_debug_stamps(dev, st); acc += np.array(list(st)[:5])
    acc *= 0.01 / 10  # 10 ns ticks -> us
    print("k_sweep_end<1> / the hub workgroup of k_rest_hub<1> (last velocity sweep of a step, mean of 10 steps), us since its start: tail colours done %.1f, hub fixed point done %.1f, leftover rows done %.1f, joints done %.1f; bodies handed over by the rest rows settled %.1f" % tuple(acc))
    bodyRest = np.zeros(nb, np.uint64)
    if nh > 0 and L.b2hip_debug_read(dev, 21, 0, nb, bodyRest.ctypes.data) == 0:
        bit = np.uint64(1) << np.uint64(63)
        serial = (bodyRest & bit) != 0
        restdeg = np.array([bin(int(x) & ((1 << 63) - 1)).count("1") for x in bodyRest])
        print("bodies with the serial bit %d, of them with rest rows %d (histogram of their rest rows: %s)" % (int(serial.sum()), int((serial & (restdeg > 0)).sum()), np.bincount(restdeg[serial]).tolist()))
print({k: round(v, 3) for k, v in w.profile().items() if v and k != "steps"})
w.close()
