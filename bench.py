#!/usr/bin/env python3
"""bench.py - world steps/sec of the MI355X-native b2World::Step() hot path.

A "step" is one full Step(1/60 s, 8 velocity / 3 position iterations) of the workload, including the mandatory host-visible
body-state read-back. BASELINE.json quotes its metric on several configurations; the N = 1 workload is the
LARGEST one it assigns to a single GPU - configs[2], "Tumbler 100k bodies, CCD off, 1 x MI355X": 316 x 316 = 99 856 boxes in a
revolving container that hangs on a motorised revolute joint (Testbed/Tests/Tumbler.h:31-87, scaled as SURVEY.md 8d says) -
measured in its SETTLED state: the boxes start on a grid that fills the container, have come down after ~400 steps and slosh
until ~700 (the step time follows the contact count: SETTLE below), so the
scene is stepped for SETTLE[...] untimed steps as part of building the workload whatever --warmup says (that transient is
reported separately), then --warmup untimed steps, then exactly --steps timed steps. The other configurations are
`extra_configs` of the same line (config 2 = Pyramid 141, config 4's per-GPU share = Pyramid 316, config 5 on ONE GPU = the
1 M-body field with 10 000 bullets), each in its own settled window with its own solver roofline and CPU baseline.

N > 1 (`--gpus N`, one rank per GPU, RCCL): ONE world of N such containers side by side, sharded by spatial ownership
(include/b2hip.h: b2hip_shard_spatial) - rank r evaluates, solves and moves the bodies of container r, the exchange runs
inside b2hip_step; "weak" scaling, `value` = container-steps/s = world steps/s x N. Without a launcher `python bench.py
--gpus N` starts the N ranks itself. World flags follow the configuration (config 3: CCD off; the pyramids and the field:
the reference's defaults - continuous physics ON); sleeping and warm starting are ON everywhere, on the GPU path and on the
CPU baselines alike.

One JSON line is printed by rank 0. Beside the contract's keys:
  roofline      the large-island SOLVER FAMILY of the timed workload - every kernel between the island build and
                SynchronizeFixtures that works on the large islands (k_large_integrate / init / velocity, k_large_rest,
                k_sweep_end, k_large_position, store_impulses, finalize, sleep), aggregated: SURVEY 8d's algorithmic bytes
                of one solve / the family's duration per step, measured live with one HIP event pair per step on the world's
                stream; `launches_per_step` says how many dependent launches that is. `traffic` = HBM bytes per step of
                the same kernels from the committed rocprofv3 --pmc passes (profiles/, never sampled in-process).
  cpu_baseline  the reference build (oracle/_ref, kind "reference", 8 threads = b2_maxThreads) on the host: the SAME scene
                in the SAME window where the reference can afford to get there inside its time budget - for the Tumbler
                steps 60..79, with the GPU's figure for that very window beside it - and it says which window it is.
  per_step_distribution_300   300 single steps of the timed workload behind the timed region: mean / p50 / p99.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "box2d-mt_amd", "python"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8 TB/s
# Settle steps per workload (workload construction, untimed, always): the Tumbler's boxes have come down after ~400 steps; the
# 141-row pyramid's contact count levels off at step ~200 (SURVEY asks for >= 120); the 316-row pyramid's top row lands at
# ~240; the field is random from the start.
# (round 6: the Tumbler 700, not 400 - the boxes have come down by step ~400, but the splash they make has not died down: the
# contact count falls to a minimum of 2.4 M at step ~420, climbs to 3.1 M at ~550 and settles around 2.5 - 2.7 M from step ~700
# on, and the step time follows it - 3.35 ms at 400..449, 3.96 at 550..599, 3.46 - 3.64 from 700 to 1200
# (tools/gpu_step_series.py, profiles/r06_tumbler316_step_series.txt). Twenty steps behind step 400 were the cheapest window of
# the whole run, 12 % below the 300 steps behind them; behind step 700 the timed steps and the 300 that follow agree to ~2 %.)
SETTLE = {"tumbler": 700, "pyramid141": 240, "pyramid316": 320, "field": 30}
PARITY_TUMBLER = ("coloured order (launch per colour + k_large_rest + k_sweep_end): one island of ~370 000 constraints, a hub of ~900 "
                  "and a revolute joint; integer results (island membership, awake flags, contact counts against the reference for 30 "
                  "steps: tests/test_gpu_configs_full_size.py) exact; floats differ from the reference by the ORDER dependence of 8 + 3 "
                  "Gauss-Seidel sweeps - one step from a bit-identical snapshot on Tumbler 2000: 5.4e-4 of the scene "
                  "(tests/test_gpu_onestep.py, bound 8.5e-4: above north_star's 1e-4, stated); bit-identical to launch-per-colour under "
                  "the same colouring (tests/test_gpu_sweep_end.py); the bit-exact class (B2HIP_FORCE_LARGE=2) is priced on config 2 below")
STATE_VS_REFERENCE_TUMBLER = ("NOT the reference's state: the reference build run to the same steps (700..760: tests/golden/settled_windows.npz, 3 hours on 8 cores) "
                              "holds 5.38 M contacts / 712 000 touching with penetration p99 0.237 m, this path's default mode 2.6 M / 370 000 with p99 0.16 m - "
                              "8 + 3 Gauss-Seidel iterations cannot carry a pile 250 boxes deep in either order, and the reference's depth-first order leaves it "
                              "more penetrated than colour order does (the effect grows with depth: 1 - 2 % at 10 000 boxes, 9 - 12 % at 22 500, 2 x here; "
                              "profiles/r06_c_order_effect_tumbler150.txt, tests/test_gpu_settled_windows.py). The timed state therefore holds about half the "
                              "contacts the reference would be stepping; where this path's own pile was that dense (steps 150..300: 5 M contacts, 520 000 - 630 000 "
                              "touching) a step takes 4.5 - 4.9 ms (`transient.steps_200_299_ms_per_step`, tools/gpu_step_series.py; 5.5 - 6.3 before the second half of round 6)")
PARITY_PYRAMID = ("coloured order (k_solve_blocks): integer results exact; ONE step from a bit-identical snapshot of the timed state: "
                  "|dp| <= 1.7 cm on 1 m boxes (1.13e-4 of the 150 m scene; median 1-2 mm), |dv| <= 0.30 m/s, 18 of 30 000 contacts differ "
                  "(tests/test_gpu_onestep.py); the bit-exact class is `exact_order`")
FAMILY_KERNELS = ("k_large_integrate", "k_large_init", "k_large_velocity", "k_large_rest", "k_rest_hub", "k_large_warm", "k_sweep_end", "k_large_position", "k_large_store_impulses", "k_large_after_velocity",
                  "k_large_integrate_positions", "k_large_pos_begin", "k_large_finalize", "k_large_sleep", "k_large_hub", "k_large_joints", "k_large_pos_end", "k_joints_sort")


def committed_pmc_traffic(kernel, workload_key):
    """HBM bytes per launch (or, for a kernel family, per step) from a committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE profile
    of THIS workload in THIS state (profiles/*_pmc_traffic.json: {"kernel", "workload", "state": "steady", "hbm_bytes_per_dispatch"
    or "hbm_bytes_per_step"}); PMC counters cannot be sampled from inside the process. None when no committed profile matches."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        if pmc.get("kernel") == kernel and pmc.get("workload") == workload_key and pmc.get("state") == "steady":
            best = (pmc.get("hbm_bytes_per_step", pmc.get("hbm_bytes_per_dispatch")), os.path.relpath(path, ROOT))
    return best


def _timing_api(hipL):
    hipL.b2hip_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    hipL.b2hip_set_kernel_timing_units.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong]
    hipL.b2hip_get_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_char), C.c_int, C.POINTER(C.c_float),
                                             C.POINTER(C.c_int), C.POINTER(C.c_double)]


def kernel_roofline(hipL, dev, step_fn, mode, steps, units=None):
    """HIP-event timing of one kernel (family) over `steps` steps (b2hip_set_kernel_timing modes: 1 dominant solver kernel,
    2 k_collide, 3 k_sync_fixtures, 4 k_find_pairs_small, 5 the large-island solver family of the launch-per-colour path - one
    pair per step): {"kernel", "achieved" GB/s of algorithmic bytes, "frac", ...}."""
    _timing_api(hipL)
    if units is not None:
        hipL.b2hip_set_kernel_timing_units(dev, int(units[0]), int(units[1]))
    hipL.b2hip_set_kernel_timing(dev, mode)
    names = {}
    for _ in range(steps):
        step_fn()
        buf = C.create_string_buffer(192)
        ms, launches, nbytes = C.c_float(), C.c_int(), C.c_double()
        hipL.b2hip_get_kernel_timing(dev, buf, 192, C.byref(ms), C.byref(launches), C.byref(nbytes))
        acc = names.setdefault(buf.value.decode(), [0.0, 0, 0.0])
        acc[0] += ms.value
        acc[1] += launches.value
        acc[2] += nbytes.value
    hipL.b2hip_set_kernel_timing(dev, 0)
    kname, (tot_ms, launches, tot_bytes) = max(names.items(), key=lambda kv: kv[1][0])
    if launches <= 0 or tot_ms <= 0:
        return None
    achieved = tot_bytes / (tot_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": None, "launches_per_step": launches / float(steps), "timed_steps": steps}
    if mode == 5:
        out["family_ms_per_step"] = tot_ms / steps
        out["algorithmic_bytes_per_step"] = tot_bytes / steps
        out["kernels"] = list(FAMILY_KERNELS)
    else:
        out["mean_launch_us"] = 1000.0 * tot_ms / launches
        out["algorithmic_bytes_per_launch"] = tot_bytes / launches
    return out


def attach_committed_traffic(roof, workload_key):
    """HBM-side bytes are PMC counters, which cannot be sampled from inside the process: they come from a committed rocprofv3
    --pmc profile of the same workload in the same state, and the key says so."""
    if roof is None:
        return
    hit = committed_pmc_traffic(roof["kernel"].split(" (")[0], workload_key)
    if hit is not None and hit[0] is not None:
        roof["traffic"] = hit[0]
        roof["traffic_committed_profile"] = hit[0]
        roof["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over the same workload and state; NOT sampled in this run)" % hit[1]
        per = roof.get("mean_launch_us", 1000.0 * roof.get("family_ms_per_step", 0.0))
        if per:
            roof["traffic_GBps"] = hit[0] / (1e-6 * per) / 1e9


class _ShardStats(C.Structure):
    _fields_ = [("rank", C.c_int32), ("count", C.c_int32), ("owned_bodies", C.c_int32), ("owned_proxies", C.c_int32),
                ("owned_contacts", C.c_int32), ("islands_solved", C.c_int32), ("constraint_rows", C.c_int32), ("pad", C.c_int32),
                ("migrated_bodies", C.c_int64), ("resolutions", C.c_int64), ("bytes_received_last_step", C.c_int64),
                ("pairs_sent", C.c_int64), ("toi_redos", C.c_int64)]


def shard_stats(hipL, dev):
    """b2hip_get_shard_stats of a spatially sharded world, as a dict"""
    st = _ShardStats()
    hipL.b2hip_get_shard_stats.argtypes = [C.c_void_p, C.POINTER(_ShardStats)]
    if hipL.b2hip_get_shard_stats(dev, C.byref(st)) != 0:
        return None
    return {n: int(getattr(st, n)) for n, _ in _ShardStats._fields_ if n != "pad"}


def distribution(per_ms):
    per_ms = np.asarray(per_ms)
    return {"steps": int(per_ms.size), "ms_per_step": float(per_ms.mean()), "ms_per_step_p50": float(np.percentile(per_ms, 50)),
            "ms_per_step_p99": float(np.percentile(per_ms, 99)), "ms_per_step_max": float(per_ms.max()), "steps_per_s": 1000.0 / float(per_ms.mean())}


def time_steps(step_fn, n):
    stamps = np.empty(n + 1)
    stamps[0] = time.perf_counter()
    for k in range(n):
        step_fn()
        stamps[k + 1] = time.perf_counter()
    return 1000.0 * np.diff(stamps)


def cpu_baseline_window(scene, p0, p1, flags, seed, first, steps, max_seconds, threads=8):
    """The reference build on the host, bounded: `steps` steps of the scene starting at step `first` if the reference gets there
    inside the budget (judged from its first two steps), else as many steps from t = 0 as fit - and the entry says which."""
    import b2harness as bh
    if not bh.have_ref():
        return None
    try:
        ref = bh.Harness(bh.REF_LIB)
        t00 = time.perf_counter()
        w = ref.world(scene, p0, p1, seed=seed, flags=flags, threads=threads)
        build_s = time.perf_counter() - t00
        # (the first step builds the reference's broad-phase tree - 5 s for a million proxies - and says nothing about the others:
        # the window is judged from the second)
        t0 = time.perf_counter()
        w.step(1)
        one = time.perf_counter() - t0
        t0 = time.perf_counter()
        w.step(1)
        two = time.perf_counter() - t0
        done = 2
        reach = first > 2 and one + two * (first + min(steps, 10) - 1) < max_seconds
        if reach:
            w.step(first - done)
            done = first
        elif first <= 2:
            reach = True
        t1 = time.perf_counter()
        timed = 0
        while timed < steps and (timed < 2 or time.perf_counter() - t1 < (max_seconds / 2 if reach else max_seconds)):
            w.step(1)
            timed += 1
        dt = time.perf_counter() - t1
        res = {"value": timed / dt, "unit": "steps/s", "ms_per_step": 1000.0 * dt / timed, "kind": "reference", "cores": threads,
               "host_cores": os.cpu_count(), "bodies": w.body_count, "contacts": w.contact_count, "build_s": round(build_s, 2),
               "window": [done, done + timed - 1], "same_window_as_requested": bool(reach),
               "sample": "steps %d..%d of the same scene on the reference build (%d threads%s)" % (done, done + timed - 1, threads, ": b2_maxThreads" if threads == 8 else "") +
                         ("" if reach else "; the requested window starts at step %d - a settle of that length is unaffordable for the reference inside the %.0f s this baseline may take (%.2f s per step at the start), so these are its FIRST steps" % (first, max_seconds, two))}
        w.close()
        return res
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def gpu_window(amd, scene, p0, p1, flags, seed, first, steps):
    """The same scene on the GPU in the window [first, first + steps): what the reference's same-window figure stands beside."""
    w = amd.world(scene, p0, p1, seed=seed, flags=flags)
    if first:
        w.step(first)
    per = time_steps(lambda: w.step(1), steps)
    out = {"window": [first, first + steps - 1], "ms_per_step": float(per.mean()), "steps_per_s": 1000.0 / float(per.mean()), "bodies": w.body_count, "contacts": w.contact_count}
    w.close()
    return out


def time_extra(amd, hipL, name, scene, p0, p1, flags, settle, roof_modes, workload_key, cpu_first, cpu_seconds, seed=3, long_steps=300, exact_order=False, parity=None, cpu_budget=1.0):
    """One of the other BASELINE configs on this GPU: settle (the window is stated in the entry), `long_steps` single steps
    timed one by one (read-back included: mean / p50 / p99), then roofline passes - the configuration's SOLVER kernel first
    (mode 1; mode 5 where the solver is a launch-per-colour family), then its dominant bandwidth kernel - and the CPU baseline."""
    import b2harness as bh
    import b2hip
    t0 = time.perf_counter()
    w = amd.world(scene, p0, p1, seed=seed, flags=flags)
    build_s = time.perf_counter() - t0
    w.step(settle)
    w.reset_profile()
    per = time_steps(lambda: w.step(1), long_steps)
    ctr = b2hip.Counters()
    dev = C.c_void_p(w.device_world())
    hipL.b2hip_get_counters(dev, C.byref(ctr))
    if parity is None and scene == bh.FIELD and ctr.large_island_contacts == 0:  # (every island of this world is in the reference-order tier)
        parity = "reference order, bit-exact: every island lies in the reference-order tier (large_island_constraints = 0); tests/test_gpu_configs_full_size.py pins 12 steps of THIS world (1 000 000 bodies, 10 000 bullets, seed 3) against hashes from the reference build"
    out = {"workload": name, "bodies": w.body_count, "contacts": w.contact_count, "settle_steps": settle, "timed_steps": long_steps, "parity_class": parity,
           "timed_window": "steps %d..%d of the scene" % (settle, settle + long_steps - 1), "build_s": round(build_s, 2)}
    out.update({k: v for k, v in distribution(per).items() if k != "steps"})
    out.update({"islands": ctr.islands, "large_island_constraints": ctr.large_island_contacts, "small_island_constraints": ctr.small_island_contacts,
                "toi_events_last_step": ctr.toi_events, "device_profile_ms": {k: round(v, 4) for k, v in w.profile().items() if k != "steps"}})
    for key, mode in roof_modes:
        try:
            # algorithmic units: collide 480 B per polygon-polygon contact (the Tumbler has nothing else), sync fixtures 250 B per proxy
            units = {2: (w.contact_count, 0), 3: (hipL.b2hip_fixture_count(dev), 0), 4: (hipL.b2hip_fixture_count(dev), 0)}.get(mode)
            roof = kernel_roofline(hipL, dev, lambda: w.step(1), mode, 5, units)
            attach_committed_traffic(roof, workload_key)
            out[key] = roof
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": str(e)}
    if exact_order:
        out["exact_order"] = exact_order_cost(hipL, w)
    w.close()
    # (cpu_budget: the 1 M field's reference needs ~45 s to reach the window the GPU is timed in - 1.2 s per step - and gets them:
    # same scene, same window on both sides)
    cb = cpu_baseline_window(scene, p0, p1, flags, seed, cpu_first, 20, cpu_seconds * cpu_budget)
    if cb is not None and "error" not in cb and cb["window"][0] != settle:
        try:
            cb["gpu_same_window"] = gpu_window(amd, scene, p0, p1, flags, seed, cb["window"][0], cb["window"][1] - cb["window"][0] + 1)
        except Exception as e:  # noqa: BLE001
            cb["gpu_same_window"] = {"error": str(e)}
    out["cpu_baseline"] = cb
    return out


def exact_order_cost(hipL, w, steps=3):
    """What the bit-exact parity class costs on a workload: the state `w` is in, saved and loaded into a world in exact-order mode
    (every island walked in the reference's constraint order), a few steps timed."""
    import b2hip
    import torch
    try:
        holder = b2hip.World.__new__(b2hip.World)  # (a view of the harness's device world for the snapshot call; not closed)
        holder.L, holder.p = hipL, C.c_void_p(w.device_world())
        blob = holder.save_snapshot()
        holder.p = None
        os.environ["B2HIP_FORCE_LARGE"] = "2"  # read when the world is created
        try:
            ex = b2hip.World.from_snapshot(blob, library=hipL)
        finally:
            os.environ.pop("B2HIP_FORCE_LARGE", None)
        ex.step(1.0 / 60.0, w.vel_iters, w.pos_iters)  # (first step: allocations)
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(steps):
            ex.step(1.0 / 60.0, w.vel_iters, w.pos_iters)
        torch.cuda.synchronize()
        ems = 1000.0 * (time.perf_counter() - te) / steps
        ex.close()
        return {"mode": "B2HIP_FORCE_LARGE=2: every island in the reference's constraint order, bit-identical to the reference build (tests/test_gpu_parity.py)",
                "state": "snapshot of the world after its timed window", "timed_steps": steps, "ms_per_step": ems, "steps_per_s": 1000.0 / ems}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="tumbler", choices=["tumbler", "pyramid141"], help="the timed workload: BASELINE config 3 (default) or config 2")
    ap.add_argument("--tumbler", type=int, default=316, help="boxes per side of the Tumbler's grid (316 -> 99 856 boxes, BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long-window", action="store_true", help="skip the 300-step per-step distribution behind the timed region")
    ap.add_argument("--no-secondary", action="store_true", help="skip the multi-island roofline sample (500 k bodies in 100 k piles)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs (Pyramid 141, 1 M field, 50 k pyramid)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of each CPU baseline (the headline's: twice that)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) as children BEFORE this process
        # touches a GPU and hand their exit code on; N = 1 stays the plain in-process path.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and args.gpus != world_size:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world_size))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the Step() path has no CPU fallback")
    if world_size > 1 and os.environ.get("B2_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < world_size:
        raise SystemExit("bench.py: %d ranks but %d GPUs visible (B2_BENCH_SHARE_GPU=1 B2_BENCH_BACKEND=gloo shares one for a functional run)"
                         % (world_size, torch.cuda.device_count()))
    # (B2_BENCH_SHARE_GPU=1 + B2_BENCH_BACKEND=gloo: every rank on GPU 0, collectives over gloo - how the N > 1 path is
    # exercised on a one-GPU box; the driver's runs use one GPU per rank and RCCL)
    if os.environ.get("B2_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        backend = os.environ.get("B2_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import b2harness as bh
    import b2hip
    if not bh.have_amd():
        raise SystemExit("box2d-mt_amd/libb2amd_harness.so missing: run `python __graft_entry__.py` first")
    amd = bh.Harness(bh.AMD_LIB)
    hipL = b2hip.lib()
    ccd = bh.F_SLEEP | bh.F_WARM | bh.F_CONTINUOUS
    if args.workload == "tumbler":
        scene, p0, p1, flags, wkey, settle = bh.TUMBLER, args.tumbler, (world_size if world_size > 1 else 0), bh.F_SLEEP | bh.F_WARM, "tumbler%d" % args.tumbler, SETTLE["tumbler"]
        unit_name = "container"
    else:
        scene, p0, p1, flags, wkey, settle = bh.PYRAMID, 141, world_size, ccd, "pyramid141", SETTLE["pyramid141"]
        unit_name = "pyramid"

    # N > 1: ONE world of `world_size` such containers (pyramids) side by side, sharded by spatial ownership
    # (include/b2hip.h: b2hip_shard_spatial; strips of equal body count along x = one container per rank): a rank evaluates,
    # solves and moves its own bodies; the library exchanges fat AABBs / awake bits / new pairs inside b2hip_step over its own
    # RCCL communicator (gloo: an all-gather of host memory). (hipSetDevice above selects this rank's GPU for the world's stream.)
    t_build = time.perf_counter()
    w = amd.world(scene, p0, p1, flags=flags)
    build_s = time.perf_counter() - t_build
    nbodies = w.body_count
    sharded = None
    if world_size > 1:
        import sharding

        class _Raw:  # the C-ABI world behind the drop-in b2World, as sharding.SpatialWorld wants it
            pass
        raw = _Raw()
        raw.p = C.c_void_p(w.device_world())
        raw.L = hipL
        sharded = sharding.SpatialWorld(raw, dist=dist, device=torch.device("cuda", local_rank))
        sharded.exchange_bytes = 0
        step_world = lambda n=1: [sharded.step(1.0 / 60.0, w.vel_iters, w.pos_iters) for _ in range(n)]
    else:
        step_world = lambda n=1: w.step(n)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    dev = C.c_void_p(w.device_world())
    ROOF_STEPS = 5

    def solver_roofline_pass():
        """The solver of the timed workload: the launch-per-colour family (mode 5) where that is what runs, else the dominant
        resident solver kernel (mode 1: k_solve_blocks on the pyramid)."""
        try:
            # (the mode follows the WORKLOAD, never the outcome: every rank of a sharded world must take the same number of steps -
            # a fallback to another mode on a rank whose family happened to be empty left the ranks out of step, round 5)
            roof = kernel_roofline(hipL, dev, lambda: step_world(1), 5 if args.workload == "tumbler" else 1, ROOF_STEPS)
            ctr = b2hip.Counters()
            hipL.b2hip_get_counters(dev, C.byref(ctr))
            if roof is not None:
                roof.update({"constraints": ctr.large_island_contacts, "bodies": ctr.large_island_bodies, "colors": ctr.colors,
                             "hub_constraints": ctr.hub_constraints, "position_iterations_executed": ctr.pos_iterations_large})
                attach_committed_traffic(roof, wkey)
            return roof
        except Exception as e:  # the timing hooks are best effort; the headline number does not depend on them
            return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "error": str(e)}

    # workload construction: settle (untimed, always), keeping the cost of the transient for the report
    settle_ms = time_steps(lambda: step_world(1), settle)
    roof = None
    if args.warmup >= ROOF_STEPS:
        step_world(args.warmup - ROOF_STEPS)
        roof = solver_roofline_pass()  # (the last ROOF_STEPS warm-up steps: the state the timed region starts from)
    else:
        step_world(args.warmup)
    w.reset_profile()
    barrier()
    # one Step() per call so that the per-step distribution can be reported as well (Step() returns after its read-back,
    # so a call is one complete step; the loop adds ~1 us of Python per step to the timed region)
    t0 = time.perf_counter()
    per_step_ms = time_steps(lambda: step_world(1), args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = w.profile()  # device phase times (clock stamps) averaged over the timed steps only
    if sharded is not None:
        ms13 = (C.c_float * 13)()
        hipL.b2hip_get_profile.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        if hipL.b2hip_get_profile(dev, ms13) == 0:
            prof = dict(zip(bh.PROFILE_FIELDS, [float(x) for x in ms13]))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if roof is None:
        roof = solver_roofline_pass()
    ctr = b2hip.Counters()
    hipL.b2hip_get_counters(dev, C.byref(ctr))
    contacts = w.contact_count

    # ---- the per-step distribution over 300 steps behind the timed region, whatever --steps was (not part of `value`)
    long_window = None
    if not args.no_long_window:
        try:
            first = settle + args.warmup + args.steps + (ROOF_STEPS if args.warmup < ROOF_STEPS else 0)
            long_window = distribution(time_steps(lambda: step_world(1), 300))
            long_window["timed_window"] = "steps %d..%d of the scene (the same world, behind the timed region)" % (first, first + 299)
        except Exception as e:  # noqa: BLE001
            long_window = {"error": str(e)}
    # the same window length once more with the read-back on demand (b2hip_set_lazy_readback): nobody looks at a body between
    # these steps, the states come home once at the end (inside the timed region)
    lazy = None
    if world_size == 1:
        try:
            hipL.b2hip_set_lazy_readback.argtypes = [C.c_void_p, C.c_int]
            if hipL.b2hip_set_lazy_readback(dev, 1) == 0:
                one = (C.c_float * 10)()
                hipL.b2hip_get_body_states.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
                tl = time.perf_counter()
                w.step(20)
                hipL.b2hip_get_body_states(dev, 0, 1, one)
                lazy = 1000.0 * (time.perf_counter() - tl) / 20
                hipL.b2hip_set_lazy_readback(dev, 0)
        except Exception:  # noqa: BLE001
            lazy = None
    spatial_stats = shard_stats(hipL, dev) if sharded is not None else None
    w.close()

    # ---- secondary roofline: the in-LDS small-island kernel on a multi-island world (not part of `value`) ----------
    secondary = None
    if world_size == 1 and not args.no_secondary:
        try:
            w2 = amd.world(bh.PILES, 100000, 5, seed=3, flags=ccd)
            dev2 = w2.device_world()
            w2.step(40)
            secondary = kernel_roofline(hipL, C.c_void_p(dev2), lambda: w2.step(1), 1, 10)
            if secondary is not None:
                secondary["workload"] = "100 000 piles of 5 boxes (%d bodies, %d contacts), same step parameters" % (w2.body_count, w2.contact_count)
                secondary["achieved_is"] = "algorithmic (reference-layout) bytes / launch time; the kernel keeps rows in LDS / registers, so counter traffic is lower"
                attach_committed_traffic(secondary, "piles100000x5")
            w2.close()
        except Exception as e:
            secondary = {"error": str(e)}

    # ---- the other BASELINE configs (not part of `value`): config 2, the 1-GPU form of config 5, config 4's share ----
    extras = None
    if not args.no_extras:
        extras = []
        jobs = []
        if world_size == 1:
            if args.workload == "tumbler":
                jobs.append(dict(name="config 2: Pyramid 141 rows = 10 011 boxes on a ground edge, one island, CCD on (Pyramid.h)", scene=bh.PYRAMID, p0=141, p1=1, flags=ccd,
                                 settle=SETTLE["pyramid141"], roof_modes=[("roofline", 1)], workload_key="pyramid141", cpu_first=SETTLE["pyramid141"], exact_order=True, parity=PARITY_PYRAMID, seed=1))
            else:
                jobs.append(dict(name="config 3: Tumbler 316 x 316 = 99 856 boxes in a revolving container, CCD off (Tumbler.h)", scene=bh.TUMBLER, p0=316, p1=0, flags=bh.F_SLEEP | bh.F_WARM,
                                 settle=SETTLE["tumbler"], roof_modes=[("roofline", 5), ("roofline_collide", 2)], workload_key="tumbler316", cpu_first=60, parity=PARITY_TUMBLER, seed=1))
            jobs.append(dict(name="config 5 on ONE GPU: 1 M mixed circles + boxes random field, 10 000 bullets, CCD on", scene=bh.FIELD, p0=1000000, p1=10000, flags=ccd,
                             settle=SETTLE["field"], roof_modes=[("roofline", 1), ("roofline_sync_fixtures", 3)], workload_key="field1000000", cpu_first=SETTLE["field"], long_steps=300, cpu_budget=6.0))
            jobs.append(dict(name="config 4, one GPU's share: Pyramid 316 rows = 50 086 boxes, CCD on", scene=bh.PYRAMID, p0=316, p1=1, flags=ccd,
                             settle=SETTLE["pyramid316"], roof_modes=[("roofline", 1)], workload_key="pyramid316", cpu_first=60))
        for job in jobs:
            try:
                extras.append(time_extra(amd, hipL, cpu_seconds=args.cpu_seconds, **job))
            except Exception as e:
                extras.append({"workload": job["name"], "error": str(e)})
        if dist is not None:
            # the configurations BASELINE assigns to SEVERAL GPUs, as stated, each as ONE world sharded by spatial ownership over
            # the ranks; whole-job rate from the slowest rank's clock (barrier + synchronise on both sides of the timed steps):
            #   config 4: `world_size` disjoint 50 086-box pyramids, one per rank;
            #   config 5: the 1 M-body field with 10 000 bullets, CCD on, in `world_size` strips of equal body count (round 6: it
            #             was missing from the N > 1 line - VERDICT r05 missing #4).
            def sharded_leg(label, scene_id, a0, a1, settle_steps, timed_steps, unit_rate_name, units):
                try:
                    wx = amd.world(scene_id, a0, a1, flags=ccd, seed=3 if scene_id == bh.FIELD else 1)
                    rawx = _Raw()
                    rawx.p = C.c_void_p(wx.device_world())
                    rawx.L = hipL
                    sx = sharding.SpatialWorld(rawx, dist=dist, device=torch.device("cuda", local_rank))
                    sx.exchange_bytes = 0
                    for _ in range(settle_steps):
                        sx.step(1.0 / 60.0, wx.vel_iters, wx.pos_iters)
                    barrier()
                    tx = time.perf_counter()
                    for _ in range(timed_steps):
                        sx.step(1.0 / 60.0, wx.vel_iters, wx.pos_iters)
                    barrier()
                    el = torch.tensor([time.perf_counter() - tx], dtype=torch.float64, device="cuda")
                    dist.all_reduce(el, op=dist.ReduceOp.MAX)
                    ms = 1000.0 * float(el.item()) / timed_steps
                    out = {"workload": label, "shard_stats_rank0": shard_stats(hipL, rawx.p), "bodies": wx.body_count, "contacts": wx.contact_count,
                           "settle_steps": settle_steps, "timed_steps": timed_steps, "ms_per_step": ms, "world_steps_per_s": 1000.0 / ms,
                           unit_rate_name: units * 1000.0 / ms,
                           "collectives": ("the library's own RCCL communicator on the world's stream" if sx.connected else "torch.distributed over host memory: NOT RCCL")}
                    wx.close()
                    return out
                except Exception as e:  # noqa: BLE001
                    return {"workload": label, "error": str(e)}

            extras.append(sharded_leg("config 4: %d disjoint pyramids of 316 rows (50 086 boxes each) in one world sharded by spatial ownership over %d GPUs, CCD on" % (world_size, world_size),
                                      bh.PYRAMID, 316, world_size, 60, 20, "island_steps_per_s_all_ranks", world_size))
            extras.append(sharded_leg("config 5: 1 M mixed circles + boxes random field, 10 000 bullets, CCD on, ONE world sharded by spatial ownership over %d GPUs (every rank keeps the id tables and the contact structure: DESIGN.md section 7)" % world_size,
                                      bh.FIELD, 1000000, 10000, SETTLE["field"], 20, "body_steps_per_s_all_ranks", 1000001))

    # ---- CPU baseline of the timed workload: the reference build, 8 threads, bounded. The reference cannot afford the GPU's
    # settled window (the Tumbler: ~0.1-0.3 s per step, 400 steps), so it is timed in a window it can reach - steps 60..79 -
    # and the GPU is timed in that very window beside it.
    cb = None
    if world_size == 1 and not args.no_cpu_baseline:
        cpu_first = 60 if args.workload == "tumbler" else settle + args.warmup
        cb = cpu_baseline_window(scene, p0, p1, flags, 1, cpu_first, 20, 2.0 * args.cpu_seconds)
        if cb is not None and "error" not in cb:
            try:
                cb["gpu_same_window"] = gpu_window(amd, scene, p0, p1, flags, 1, cb["window"][0], cb["window"][1] - cb["window"][0] + 1)
                cb["gpu_over_cpu_same_window"] = cb["gpu_same_window"]["steps_per_s"] / cb["value"]
            except Exception as e:  # noqa: BLE001
                cb["gpu_same_window"] = {"error": str(e)}
            cb["note"] = "a reported baseline, not the target: the roofline fraction is what measures the kernels"

    if rank == 0:
        total_steps = args.steps * world_size
        line = {
            "metric": "world steps/sec at %d bodies per GPU (BASELINE config %s; Step = collide + island solve + broad-phase%s + state read-back); for N > 1: %s-steps/s = world steps/s x N %ss in the one sharded world"
                      % (nbodies // world_size, "3: Tumbler 100k, CCD off" if args.workload == "tumbler" else "2: Pyramid 10k", "" if args.workload == "tumbler" else " + TOI", unit_name, unit_name),
            "world_steps_per_s": args.steps / elapsed,
            "value": total_steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "ms_per_step_p50": float(np.percentile(per_step_ms, 50)),
            "ms_per_step_p99": float(np.percentile(per_step_ms, 99)),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (("Tumbler %d x %d: %d bodies (%d boxes of 0.25 m in a revolving container on a motorised revolute joint), %d contacts" % (args.tumbler, args.tumbler, nbodies, args.tumbler * args.tumbler, contacts)
                                     if args.workload == "tumbler" else "Pyramid 141 rows: %d bodies, %d contacts" % (nbodies, contacts)) if world_size == 1 else
                                    "%d %ss in ONE world (%d bodies, %d contacts in all; every rank keeps the id tables and the contact structure, owns one %s)" % (world_size, unit_name, nbodies, contacts, unit_name)) +
                                   ", settled for %d untimed steps (steady state), dt 1/60, 8 vel / 3 pos iterations, CCD %s, sleep + warm start on"
                                   % (settle, "off (as BASELINE config 3 says)" if args.workload == "tumbler" else "on (reference default)"),
                       "baseline_config": 3 if args.workload == "tumbler" else 2,
                       "settle_steps": settle, "build_s": round(build_s, 2),
                       "timed_window": "steps %d..%d of the scene" % (settle + args.warmup, settle + args.warmup + args.steps - 1),
                       "parity_class": PARITY_TUMBLER if args.workload == "tumbler" else PARITY_PYRAMID,
                       "state_vs_reference": (STATE_VS_REFERENCE_TUMBLER if args.workload == "tumbler" and args.tumbler == 316 and world_size == 1 else None),
                       "bodies_total": nbodies, "islands": ctr.islands, "large_island_constraints": ctr.large_island_contacts, "colors": ctr.colors,
                       "hub_constraints": ctr.hub_constraints,
                       "parallelism": "one world over the ranks by spatial ownership: a rank evaluates, solves and moves the bodies of its strip (one %s); fat AABBs / awake bits / new pairs by all-gather, migrating components ship their content" % unit_name if world_size > 1 else "single GPU"},
            "device_profile_ms": {k: round(v, 4) for k, v in prof.items() if k != "steps"},
        }
        # the transient the settle steps went through (rank 0), never part of `value`
        line["transient"] = {"steps": "0..%d (workload construction, untimed)" % (settle - 1), "ms_per_step": float(settle_ms.mean()),
                             "ms_per_step_p50": float(np.percentile(settle_ms, 50)), "ms_per_step_p99": float(np.percentile(settle_ms, 99)),
                             "ms_per_step_max": float(settle_ms.max())}
        if args.workload == "tumbler" and settle >= 320:
            # (the pile at its densest - ~5 M contacts, 520 000 - 630 000 touching: what the REFERENCE's pile still looks like at
            # step 700, `state_vs_reference`)
            line["transient"]["steps_200_299_ms_per_step"] = float(settle_ms[200:300].mean())
        if lazy is not None:
            line["ms_per_step_lazy_readback"] = lazy
        if sharded is not None and spatial_stats is not None:
            line["sharding"] = "spatial ownership (b2hip_shard_spatial): a rank evaluates, solves and moves the bodies of its strip; rows / fat AABBs of moved bodies, new pairs and migrating components travel by all-gather " + ("over the library's own RCCL communicator on the world's stream" if sharded.connected else "over torch.distributed (host memory): NOT RCCL") + ", inside the timed region"
            line["shard_stats_rank0"] = spatial_stats
            line["exchange_bytes_per_step"] = spatial_stats["bytes_received_last_step"]
        if long_window is not None:
            line["per_step_distribution_300"] = long_window
        if roof is not None:
            # (stated plainly) one island of 370 000 constraints is a chain of DEPENDENT sweeps: 8 + 3 + 1 sweeps of ~14 dependent
            # launches each; the family's algorithmic traffic (~1 GB) would take 0.13 ms at peak, the chain of launches takes
            # ~1.8 ms. What the fraction measures is that depth; the same arithmetic reaches 0.6 - 0.87 of peak where islands are
            # many and small (roofline_small_islands, the field's k_solve_small).
            roof["target_0_40"] = "not reached on one island of this depth: a sweep is a chain of dependent colour steps, each a launch (~6.7 us) or a hand-over through memory (~2.5 us) whatever it holds; see DESIGN.md section 3"
            line["roofline"] = roof
        if secondary is not None:
            line["roofline_small_islands"] = secondary
        if extras is not None:
            line["extra_configs"] = extras
        if cb is not None:
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
