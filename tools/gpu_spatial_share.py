"""What ONE rank of a spatially sharded world pays per step, measured on one GPU without the other ranks in its way (VERDICT r03
item 2: "a one-GPU proxy run showing a rank's step for config 4 <= 1.15 x a one-pyramid world").

  1. record: the world (N pyramids) over N ranks in this process (tests/spatial_util.py), every all-gather's result kept;
  2. replay: rank 0 alone on the GPU, stepping the same world again with the recorded results handed to it where the
     collective would deliver them (its own contribution is checked against the record: the run is deterministic) - the wall
     clock of its timed steps is a rank's step, the exchange being a host round trip (D2H of its slab, H2D of all slabs) in
     place of RCCL over xGMI;
  3. the same steps of a world that holds ONE pyramid, unsharded.
usage: gpu_spatial_share.py <rows> <pyramids = ranks> <settle> <timed>   (CCD on, default mode)
       gpu_spatial_share.py field <bodies> <bullets> <ranks> <settle> <timed>   (config 5: the field over `ranks` strips; part 3 is
       then the whole field on one GPU, unsharded)"""
import os, sys, time, ctypes as C, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
import spatial_util as SU
FIELD = sys.argv[1] == "field"
if FIELD:
    bodies, bullets, ranks, settle, timed = (int(a) for a in sys.argv[2:7])
    rows = bodies
else:
    rows, ranks, settle, timed = (int(a) for a in sys.argv[1:5])
amd = H.Harness(H.AMD_LIB); L = b2hip.lib()
flags = H.F_SLEEP | H.F_WARM | H.F_CONTINUOUS


def make(count):
    """the world: `count` pyramids, or the field (count is ignored: the ranks share ONE field)"""
    return amd.world(H.FIELD, bodies, bullets, seed=3, flags=flags) if FIELD else amd.world(H.PYRAMID, rows, count, seed=3, flags=flags)


WHAT = ("the field of %d bodies / %d bullets" % (bodies, bullets)) if FIELD else "%d pyramids of %d rows" % (ranks, rows)
L.b2hip_shard_spatial.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.b2hip_set_shard_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]

# ---- 1. record ------------------------------------------------------------------------------------------------------------
L.b2hip_shard_tape.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
ws = [make(ranks) for _ in range(ranks)]
sr = SU.SpatialRanks(L, [(w, w.device_world()) for w in ws])
assert L.b2hip_shard_tape(C.c_void_p(ws[0].device_world()), 1, None) == 0  # rank 0 keeps what its collectives delivered (device memory)
t0 = time.time()
marks = []
for s in range(settle + timed):
    marks.append((sr.gather.calls, sr.gather.bytes))
    sr.step()
marks.append((sr.gather.calls, sr.gather.bytes))
st = [sr.stats(r) for r in range(ranks)]
nb = ws[0].body_count
own_end = sr.owners(0, nb)
hash_end = H.fnv1a64(ws[0].bodies()[own_end == 0])
print("recorded %d steps of %s over %d ranks in %.1f s: %d collectives, %.1f MB; rank 0 owns %d bodies, %d contacts with content, %d rows of %d contacts in the world" % (
    settle + timed, WHAT, ranks, time.time() - t0, sr.gather.calls, sr.gather.bytes / 1e6, st[0].owned_bodies, st[0].owned_contacts, st[0].constraint_rows, ws[0].contact_count), flush=True)
for w in ws[1:]: w.close()

# ---- 2. replay: rank 0 alone ------------------------------------------------------------------------------------------------
w0 = make(ranks)
assert L.b2hip_shard_spatial(C.c_void_p(w0.device_world()), 0, ranks, None) == 0
assert L.b2hip_shard_tape(C.c_void_p(w0.device_world()), 2, C.c_void_p(ws[0].device_world())) == 0
w0.step(settle)
w0.reset_profile()
stamps = [time.perf_counter()]
for s in range(timed):
    w0.step(1)
    stamps.append(time.perf_counter())
per = 1000.0 * np.diff(stamps)
prof = w0.profile()
bytes_timed = (marks[settle + timed][1] - marks[settle][1]) / max(timed, 1)
gathers_timed = (marks[settle + timed][0] - marks[settle][0]) / max(timed, 1)
own0 = np.zeros(nb, np.uint8); L.b2hip_get_body_owners.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
L.b2hip_get_body_owners(C.c_void_p(w0.device_world()), nb, own0.ctypes.data_as(C.c_void_p))
same = np.array_equal(own0, own_end) and H.fnv1a64(w0.bodies()[own0 == 0]) == hash_end
print("rank 0 alone, steps %d..%d: %.3f ms per step (p50 %.3f, max %.3f); %.1f collectives delivering %.3f MB per step (replayed device to device); its bodies equal the recorded run's: %s" % (
    settle, settle + timed - 1, per.mean(), np.percentile(per, 50), per.max(), gathers_timed, bytes_timed / 1e6, same), flush=True)
print("   device profile of rank 0 (ms per step):", {k: round(v, 4) for k, v in prof.items() if k != "steps"})
w0.close()
ws[0].close()

if os.environ.get("B2_SHARE_REPLAY_ONLY") == "1":
    sys.exit(0)  # (under rocprofv3: the replayed rank's steps are the last ones in the trace -> tools/trace_steady.py)

# ---- 3. one rank's share as a world of its own (one pyramid), unsharded --------------------------------------------------------------
if not FIELD:
    w1 = make(1)
    w1.step(settle)
    w1.reset_profile()
    stamps = [time.perf_counter()]
    for s in range(timed):
        w1.step(1)
        stamps.append(time.perf_counter())
    one = 1000.0 * np.diff(stamps)
    print("one pyramid of %d rows, unsharded, same steps: %.3f ms per step (p50 %.3f)  ->  a rank of the %d-pyramid world pays %.2f x (p50 %.2f x)" % (
        rows, one.mean(), np.percentile(one, 50), ranks, per.mean() / one.mean(), np.percentile(per, 50) / np.percentile(one, 50)), flush=True)
    print("   device profile (ms per step):", {k: round(v, 4) for k, v in w1.profile().items() if k != "steps"})
    w1.close()

# ---- 4. the whole world on one GPU, unsharded (what sharding has to beat) ----------------------------------------------------------
wN = make(ranks)
wN.step(settle)
wN.reset_profile()
stamps = [time.perf_counter()]
for s in range(timed):
    wN.step(1)
    stamps.append(time.perf_counter())
allN = 1000.0 * np.diff(stamps)
print("%s in one unsharded world on one GPU: %.3f ms per step  ->  %d ranks at %.3f ms each: %.2f x" % (WHAT, allN.mean(), ranks, per.mean(), allN.mean() / per.mean()))
print("   device profile (ms per step):", {k: round(v, 4) for k, v in wN.profile().items() if k != "steps"})
wN.close()
