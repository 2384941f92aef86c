"""Spatially sharded worlds (b2hip_shard_spatial) against the unsharded world, N ranks in one process on one GPU.
usage: gpu_spatial_check.py <scene> <p0> <p1> <ranks> <steps> [exact] [ccd]"""
import os, sys, ctypes as C, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))
import b2harness as H, b2hip
from spatial_util import SpatialRanks
scene = getattr(H, sys.argv[1].upper()); p0 = int(sys.argv[2]); p1 = int(sys.argv[3]); ranks = int(sys.argv[4]); steps = int(sys.argv[5])
exact = "exact" in sys.argv[6:]; ccd = "ccd" in sys.argv[6:]; full = "full" in sys.argv[6:]
if full: os.environ["B2HIP_SHARD_FULL_ROWS"] = "1"  # every rank holds every body's row (else: a rank answers for ITS bodies)
if exact: os.environ["B2HIP_FORCE_LARGE"] = "2"
amd = H.Harness(H.AMD_LIB); L = b2hip.lib()
flags = H.F_SLEEP | H.F_WARM | (H.F_CONTINUOUS if ccd else 0)
ref = amd.world(scene, p0, p1, seed=3, flags=flags)
ws = [amd.world(scene, p0, p1, seed=3, flags=flags) for _ in range(ranks)]
sr = SpatialRanks(L, [(w, w.device_world()) for w in ws])
nb = ref.body_count
own0 = sr.owners(0, nb).copy()
bad = None
t0 = time.time()
for s in range(steps):
    ref.step(1)
    sr.step()
    rb = ref.bodies().view(np.uint32)
    for r, w in enumerate(ws):
        wb = w.bodies().view(np.uint32)
        own = sr.owners(r, nb)
        mine = np.ones(nb, bool) if full else ((own == r) | (ref.bodies()[:, 7] == 0))
        if w.contact_count != ref.contact_count or not np.array_equal(wb[mine], rb[mine]):
            d = np.nonzero((wb != rb).any(axis=1) & mine)[0]
            for b in d[:2]:
                for nm, ww in (("ref", ref), ("rank", w)):
                    ids, fl, man = ww.contacts()
                    sel = (ids[:, 0] == b) | (ids[:, 2] == b)
                    print("   %s contacts of body %d:" % (nm, b), [(ids[k].tolist(), int(fl[k]), man[k][1], man[k][8:10].tolist()) for k in np.nonzero(sel)[0]])
            print("   owners now differ from the strips for bodies", np.nonzero(own != own0)[0][:20].tolist())
            for b in d[:3]: print("   body %d ref %s\n            got %s" % (b, ref.bodies()[b].tolist(), w.bodies()[b].tolist()))
            print("step %d rank %d: contacts %d / %d, %d bodies differ, first %s (owners %s)" % (s, r, w.contact_count, ref.contact_count, len(d), d[:8].tolist(), own[d[:8]].tolist()))
            bad = s
    if bad is not None: break
print("%s %d %d over %d ranks, %d steps (%s%s): %s in %.1f s; gathers %d, %.2f MB" % (sys.argv[1], p0, p1, ranks, steps, "exact-order" if exact else "default", ", ccd" if ccd else "",
      ("BITWISE EQUAL to the unsharded world on every rank" if full else "every rank's OWN bodies BITWISE EQUAL to the unsharded world") if bad is None else "MISMATCH at step %d" % bad, time.time() - t0, sr.gather.calls, sr.gather.bytes / 1e6))
for r in range(ranks):
    st = sr.stats(r)
    print("  rank %d: owns %d bodies %d proxies %d contacts; islands %d rows %d; migrated %d in %d resolutions; pairs sent %d; TOI phases redone %d; %.1f KB received last step" % (
        r, st.owned_bodies, st.owned_proxies, st.owned_contacts, st.islands_solved, st.constraint_rows, st.migrated_bodies, st.resolutions, st.pairs_sent, st.toi_redos, st.bytes_received_last_step / 1e3))
