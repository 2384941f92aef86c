"""One world over N ranks by SPATIAL OWNERSHIP (include/b2hip.h: b2hip_shard_spatial; box2d-mt_amd/csrc/b2d_kernels_spatial.h)
against the unsharded world, on the GPU: N ranks in ONE process (one world per rank on the same device, one thread per
rank, an all-gather over host memory: tests/spatial_util.py). The multi-process form of the same protocol runs over gloo
on the CPU (tests/test_spatial_gloo.py, on the oracle's shim).

What is pinned:
  * every rank's copy of EVERY body equals the unsharded world bit for bit after every step, contact counts included -
    fields of free bodies with bullets (continuous physics on: TOI events, contacts created inside sub-steps and merged over
    the ranks, components migrating over strip boundaries), piles, rain, jointed vehicles, four pyramids (one per rank) in
    exact-order mode;
  * what a rank owns and maintains is its share: bodies, contacts with content, constraint rows ~ 1 / N;
  * an event that reaches over an ownership boundary is caught and the phase redone (the dense bullets scene).
Reference: the reference guarantees results independent of the worker count (README.md:161-175, TestMT.cpp:91-110); its own
partition of these phases: b2World.cpp:100-160, 1236-1241.
"""
import os

import numpy as np
import pytest

import b2harness as bh
import b2hip
from spatial_util import SpatialRanks

pytestmark = pytest.mark.gpu

CCD = bh.F_CONTINUOUS | bh.F_SLEEP | bh.F_WARM
PLAIN = bh.F_SLEEP | bh.F_WARM


def run_sharded(amd, scene, p0, p1, ranks, steps, flags, exact, monkeypatch, seed=3, compare=True, full=True, f0=0.0, f1=0.0):
    """full: every rank holds the current row of every body (B2HIP_SHARD_FULL_ROWS=1) and is compared whole; else the lean
    exchange of the default - a rank answers for the bodies it owns, the world is the union of the owners' rows."""
    if exact:
        monkeypatch.setenv("B2HIP_FORCE_LARGE", "2")
    else:
        monkeypatch.delenv("B2HIP_FORCE_LARGE", raising=False)
    if full:
        monkeypatch.setenv("B2HIP_SHARD_FULL_ROWS", "1")
    else:
        monkeypatch.delenv("B2HIP_SHARD_FULL_ROWS", raising=False)
    L = b2hip.lib()
    ref = amd.world(scene, p0, p1, f0, f1, seed=seed, flags=flags)
    ws = [amd.world(scene, p0, p1, f0, f1, seed=seed, flags=flags) for _ in range(ranks)]
    sr = SpatialRanks(L, [(w, w.device_world()) for w in ws])
    for s in range(steps):
        ref.step(1)
        sr.step()
        rbf = ref.bodies()
        rb = rbf.view(np.uint32)
        packed = None if full else [sr.own_rows(r, len(rb)) for r in range(ranks)]
        first = ws[0].bodies().view(np.uint32)
        claimed = np.zeros(len(rb), np.int32)
        for r, w in enumerate(ws):
            wb = w.bodies().view(np.uint32)
            assert w.contact_count == ref.contact_count, "step %d rank %d: %d contacts, the unsharded world has %d" % (s + 1, r, w.contact_count, ref.contact_count)
            if full:
                assert np.array_equal(wb, first), "step %d: rank %d and rank 0 hold different worlds" % (s + 1, r)
                mine = np.ones(len(rb), bool)
            else:
                own = sr.owners(r, len(rb))
                mine = (own == r) & (rbf[:, 7] != 0)
                claimed += mine
                # (what the step itself brought to the host: the packed rows of the rank's own bodies - read BEFORE the table
                # of all rows is asked for; position, angle and velocities are columns 0-5 of both)
                ids, rows = packed[r]
                assert np.array_equal(np.sort(ids), np.nonzero(mine)[0]), "step %d rank %d: the packed rows are not the rank's bodies" % (s + 1, r)
                assert np.array_equal(rows[:, :6], rb[ids, :6]), "step %d rank %d: a packed row differs from the unsharded world" % (s + 1, r)
                mine |= rbf[:, 7] == 0
            if compare:
                bad = np.nonzero((wb != rb).any(axis=1) & mine)[0]
                assert bad.size == 0, "step %d rank %d: %d bodies differ from the unsharded world (first %s)" % (s + 1, r, bad.size, bad[:8].tolist())
        if not full:
            assert (claimed[rbf[:, 7] != 0] == 1).all(), "step %d: a body without an owner, or with two" % (s + 1)
    stats = [sr.stats(r) for r in range(ranks)]
    nonstatic = int((ref.bodies()[:, 7] != 0).sum())
    contacts = ref.contact_count
    for w in ws:
        w.close()
    ref.close()
    return stats, nonstatic, contacts, sr


@pytest.mark.parametrize("name,scene,p0,p1,ranks,steps,flags,exact", [
    ("field with bullets, 2 ranks", bh.FIELD, 3000, 300, 2, 80, CCD, False),
    ("field with bullets, 4 ranks", bh.FIELD, 3000, 300, 4, 80, CCD, False),
    ("field 20 000 / 2 000 bullets, 4 ranks", bh.FIELD, 20000, 2000, 4, 30, CCD, False),
    ("4 pyramids, 2 ranks, exact order", bh.PYRAMID, 20, 4, 2, 100, CCD, True),
    ("4 pyramids, 4 ranks, exact order", bh.PYRAMID, 20, 4, 4, 100, CCD, True),
    ("rain, 3 ranks", bh.RAIN, 800, 0, 3, 200, CCD, False),
    ("piles, 4 ranks", bh.PILES, 100, 8, 4, 150, CCD, False),
    ("vehicles (wheel / rope / motor / mouse joints), 2 ranks", bh.VEHICLES, 30, 2, 2, 150, CCD, False),
    ("machines (prismatic / weld / gear / pulley joints), 2 ranks", bh.MACHINES, 30, 3, 2, 120, CCD, False),
    ("tumbler (hub body + motor joint: one component), 2 ranks", bh.TUMBLER, 20, 0, 2, 80, PLAIN, False),
])
def test_sharded_world_is_the_unsharded_world_bit_for_bit(amd, monkeypatch, name, scene, p0, p1, ranks, steps, flags, exact):
    stats, nonstatic, contacts, _ = run_sharded(amd, scene, p0, p1, ranks, steps, flags, exact, monkeypatch)
    assert sum(st.owned_bodies for st in stats) == nonstatic, "every non-static body has exactly one owner"
    assert sum(st.owned_contacts for st in stats) <= contacts  # (a contact between two static-only ... has none; the rest one)


@pytest.mark.parametrize("name,scene,p0,p1,ranks,steps,flags,exact", [
    ("field with bullets, 4 ranks", bh.FIELD, 3000, 300, 4, 80, CCD, False),
    ("4 pyramids, 4 ranks, exact order", bh.PYRAMID, 20, 4, 4, 100, CCD, True),
    ("rain, 3 ranks", bh.RAIN, 800, 0, 3, 200, CCD, False),
    ("bullets (events over ownership boundaries), 2 ranks", bh.BULLETS, 60, 6, 2, 120, CCD, False),
    ("vehicles, 2 ranks", bh.VEHICLES, 30, 2, 2, 150, CCD, False),
])
def test_lean_exchange_every_rank_answers_for_its_own_bodies(amd, monkeypatch, name, scene, p0, p1, ranks, steps, flags, exact):
    """The default exchange: only what the other ranks' WORK needs travels every step - the fat AABBs of moved proxies, awake
    bits that changed, new pairs - and a body's row travels when the body changes owner. A rank then answers for the bodies it
    owns: the union of the owners' rows is the unsharded world, bit for bit, every step; every body has exactly one owner."""
    run_sharded(amd, scene, p0, p1, ranks, steps, flags, exact, monkeypatch, full=False)


def test_an_event_over_an_ownership_boundary_is_caught_and_the_phase_redone(amd, monkeypatch):
    """A dense box of bullets: TOI sub-steps create contacts with bodies of the other rank. Every rank takes its phase back,
    the components merge as if the contact existed, the phase runs again (b2hip.hip: spExchangeState) - same bits."""
    stats, _, _, _ = run_sharded(amd, bh.BULLETS, 60, 6, 2, 120, CCD, False, monkeypatch)
    assert stats[0].toi_redos >= 1, "the scene no longer reaches over a boundary: the test is vacuous"


def test_a_rank_holds_and_solves_its_share(amd, monkeypatch):
    """Work and content per rank ~ 1 / N (b2hip_get_shard_stats): a field of 20 000 free bodies over 4 ranks, and 4 pyramids
    over 4 ranks. (The id tables - body rows, fat AABBs, the structure of the contact array - are replicated, as SURVEY
    section 8e prescribes: every order the results depend on stays the unsharded world's.)"""
    stats, nonstatic, contacts, _ = run_sharded(amd, bh.FIELD, 20000, 2000, 4, 30, CCD, False, monkeypatch)
    for st in stats:
        assert 0.8 * nonstatic / 4 <= st.owned_bodies <= 1.2 * nonstatic / 4, "rank %d owns %d of %d bodies" % (st.rank, st.owned_bodies, nonstatic)
        assert st.owned_contacts <= 0.4 * contacts, "rank %d maintains %d of %d contacts" % (st.rank, st.owned_contacts, contacts)
        assert st.islands_solved <= 0.3 * nonstatic
    stats, nonstatic, contacts, _ = run_sharded(amd, bh.PYRAMID, 20, 4, 4, 60, CCD, True, monkeypatch)
    for st in stats:
        assert st.owned_bodies == 210 and 1 <= st.islands_solved <= 3, "one pyramid per rank (rank %d: %d bodies, %d islands)" % (st.rank, st.owned_bodies, st.islands_solved)
        assert st.owned_contacts <= 0.26 * contacts
        assert st.migrated_bodies == 0


def test_default_mode_large_islands_stay_in_their_parity_class(amd, monkeypatch):
    """Four pyramids of 40 rows over 4 ranks in the DEFAULT mode: each pile is a large island, solved in the coloured order,
    whose block partition is the owning rank's own - the same parity class as the unsharded world (tests/test_gpu_onestep.py),
    not the same bits. Pinned: all ranks hold the same world, contact counts follow the unsharded world's, the piles stand."""
    monkeypatch.delenv("B2HIP_FORCE_LARGE", raising=False)
    monkeypatch.setenv("B2HIP_SHARD_FULL_ROWS", "1")  # (every rank holds every body's row: the ranks are compared whole)
    L = b2hip.lib()
    ref = amd.world(bh.PYRAMID, 40, 4, seed=3, flags=CCD)
    ws = [amd.world(bh.PYRAMID, 40, 4, seed=3, flags=CCD) for _ in range(4)]
    sr = SpatialRanks(L, [(w, w.device_world()) for w in ws])
    for s in range(150):
        ref.step(1)
        sr.step()
        first = ws[0].bodies().view(np.uint32)
        for w in ws[1:]:
            assert np.array_equal(w.bodies().view(np.uint32), first), "step %d: the ranks hold different worlds" % (s + 1)
        assert abs(ws[0].contact_count - ref.contact_count) <= 0.01 * ref.contact_count + 4
    a, b = ws[0].bodies(), ref.bodies()
    scale = np.abs(b[:, :2]).max()
    assert np.isfinite(a).all() and np.abs(a[:, :2] - b[:, :2]).max() / scale < 2e-2
    for w in ws:
        w.close()
    ref.close()


# ---- BASELINE configs 4 and 5 in their multi-GPU form, at FULL size, against traces of the reference build ------------------------
GOLD_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_scale.npz")


def _assemble(ws, sr, nb, static_mask):
    """the world as its owners hold it: every body's row from the rank that owns it (static bodies from rank 0)"""
    rows = ws[0].bodies().copy()
    own = sr.owners(0, nb)
    claimed = static_mask.copy()
    for r, w in enumerate(ws):
        mine = (own == r) & ~static_mask
        if r > 0:
            rows[mine] = w.bodies()[mine]
        claimed |= mine
    assert claimed.all()
    return rows


def test_config_5_over_8_ranks_at_full_size_is_the_reference_bit_for_bit(amd, monkeypatch):
    """BASELINE config 5 as stated - 1 000 000 bodies + 10 000 bullets, continuous physics on, over EIGHT ranks - in the
    default mode with the default (lean) exchange: the world assembled from its owners' rows reproduces the trace of the
    reference build (tests/golden/config_scale.npz: contact count, awake count, the hash of all 8 000 008 state words) after
    every step. Eight ranks in one process on one GPU (tests/spatial_util.py); the collective is a barrier."""
    monkeypatch.delenv("B2HIP_FORCE_LARGE", raising=False)
    monkeypatch.delenv("B2HIP_SHARD_FULL_ROWS", raising=False)
    g = np.load(GOLD_PATH)
    sc, p0, p1, seed, steps, flags = (int(x) for x in g["config5_field1m/params"])
    counts, awake, hashes = g["config5_field1m/contact_counts"], g["config5_field1m/awake"], g["config5_field1m/hashes"]
    ranks = 8
    ws = [amd.world(sc, p0, p1, seed=seed, flags=flags) for _ in range(ranks)]
    sr = SpatialRanks(b2hip.lib(), [(w, w.device_world()) for w in ws])
    nb = ws[0].body_count
    static_mask = ws[0].bodies()[:, 7] == 0
    for s in range(min(steps, 8)):
        sr.step()
        rows = _assemble(ws, sr, nb, static_mask)
        for w in ws:
            assert w.contact_count == int(counts[s]), "step %d: %d contacts, the reference has %d" % (s + 1, w.contact_count, counts[s])
        assert int((rows[:, 6] != 0).sum()) == int(awake[s]), "step %d: awake bodies" % (s + 1)
        assert bh.fnv1a64(rows) == hashes[s], "step %d: the state the 8 ranks hold differs from the reference build's" % (s + 1)
    stats = [sr.stats(r) for r in range(ranks)]
    nonstatic = int((~static_mask).sum())
    assert sum(st.owned_bodies for st in stats) == nonstatic
    for st in stats:
        assert 0.8 * nonstatic / ranks <= st.owned_bodies <= 1.2 * nonstatic / ranks, "rank %d owns %d of %d bodies" % (st.rank, st.owned_bodies, nonstatic)
        assert st.owned_contacts <= 0.25 * int(counts[0]) + 1000
    assert sum(st.migrated_bodies for st in stats) > 0, "no body ever crossed a strip boundary"
    for w in ws:
        w.close()


def test_config_4_over_4_ranks_at_full_size_against_the_reference(amd, monkeypatch):
    """BASELINE config 4 as stated - four disjoint 316-row pyramids (200 344 boxes) in one world over FOUR ranks, one pyramid
    each - in the default mode: bit for bit the reference build's trace while the rows fall (nothing is solved before they
    meet at step 13), the reference's contact counts for all 40 steps, one pyramid and a quarter of the contact content per rank."""
    monkeypatch.delenv("B2HIP_FORCE_LARGE", raising=False)
    monkeypatch.delenv("B2HIP_SHARD_FULL_ROWS", raising=False)
    g = np.load(GOLD_PATH)
    sc, p0, p1, seed, steps, flags = (int(x) for x in g["config4_4pyramids316/params"])
    counts, awake, hashes = g["config4_4pyramids316/contact_counts"], g["config4_4pyramids316/awake"], g["config4_4pyramids316/hashes"]
    ranks = 4
    ws = [amd.world(sc, p0, p1, seed=seed, flags=flags) for _ in range(ranks)]
    sr = SpatialRanks(b2hip.lib(), [(w, w.device_world()) for w in ws])
    nb = ws[0].body_count
    static_mask = ws[0].bodies()[:, 7] == 0
    for s in range(steps):
        sr.step()
        for w in ws:
            assert abs(w.contact_count - int(counts[s])) <= 1e-4 * counts[s], "step %d: %d contacts, the reference has %d" % (s + 1, w.contact_count, counts[s])
        if s < 12:
            rows = _assemble(ws, sr, nb, static_mask)
            assert bh.fnv1a64(rows) == hashes[s], "step %d (free fall): the state the 4 ranks hold differs from the reference build's" % (s + 1)
    stats = [sr.stats(r) for r in range(ranks)]
    for st in stats:
        assert st.owned_bodies == 316 * 317 // 2, "one pyramid per rank (rank %d owns %d bodies)" % (st.rank, st.owned_bodies)
        assert st.owned_contacts <= 0.26 * int(counts[-1])
        assert st.migrated_bodies == 0
    for w in ws:
        w.close()


def test_a_dense_start_grows_the_pair_buffer_of_a_sharded_world_too(amd, monkeypatch):
    """1 400 bodies and 450 bullets crammed into a 70 x 70 arena - the first pair update of every rank finds
    several times more candidate pairs than its buffer was sized for. The unsharded world grows the buffer and searches again
    (tests/test_gpu_edge_cases.py::test_dense_start_grows_the_pair_buffer); a spatially sharded world used to fail its step with
    a capacity error. Now every rank reads the overflow in the slab headers, all grow alike and all search again: the same
    bits as the unsharded world, exact-order mode (the islands are huge)."""
    # (continuous physics off: with it the first TOI phase of this arena creates more contacts inside sub-steps than the
    #  4 096 a sharded world merges per phase - SP_TAIL_MAX, a stated limit that is reported as a capacity error)
    stats, nonstatic, contacts, _ = run_sharded(amd, bh.FIELD, 1406, 456, 2, 3, PLAIN, True, monkeypatch, seed=2623, f0=35.0, f1=2.0)
    assert contacts > 15000


def test_two_revolving_containers_that_come_to_touch_merge_under_one_owner(amd, monkeypatch):
    """Two Tumbler containers 2 S + 4 apart (B2H_TUMBLER_CLOSE=1; bench.py's N > 1 world keeps them 3 S + 4 apart): after ~55
    steps their fat AABBs overlap, a contact between the two hub bodies exists, and the two components - each a container on a
    motor joint with its boxes - merge under one owner with everything they hold. Exact-order mode: every rank's world equals
    the unsharded one bit for bit through the merge, and one rank ends up owning every body."""
    monkeypatch.setenv("B2H_TUMBLER_CLOSE", "1")
    stats, nonstatic, contacts, _ = run_sharded(amd, bh.TUMBLER, 12, 2, 2, 150, PLAIN, True, monkeypatch)
    assert sorted(st.owned_bodies for st in stats) == [0, nonstatic], [st.owned_bodies for st in stats]
    assert stats[0].migrated_bodies > 0
