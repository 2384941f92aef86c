#!/bin/bash
# End-of-round measurement on the GPU box: parity suite, bench line, rocprofv3 kernel stats, PMC traffic passes (separate),
# steady-state per-step kernel budgets of the headline scene and of the other configurations in THEIR settled windows.
# Summaries land in gpurun_out/prof_round/ (copy what is judged into profiles/rNN_*).
# usage: tools/gpu_profile_round.sh [notests]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
if [ "$1" != "notests" ]; then timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/pytest_gpu.txt; fi
timeout 1200 python3 bench.py --steps 200 --warmup 60 2>$OUT/bench.err | tail -1 > $OUT/bench.json
# kernel trace of the bench command in steady state (the 240 settle steps are part of workload construction)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-extras --no-exact-order > $OUT/stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/trace_steady.py /tmp/prof_stats 80 1 > $OUT/steady_state_per_step.txt
# HBM-side traffic, FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM): headline workload ...
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 bench.py --steps 10 --warmup 20 --no-cpu-baseline --no-secondary --no-extras --no-exact-order > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 bench.py --steps 10 --warmup 20 --no-cpu-baseline --no-secondary --no-extras --no-exact-order > $OUT/write.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_fetch last 10 > $OUT/pmc_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write last 10 > $OUT/pmc_write_size.csv
python3 tools/pmc_traffic_json.py $OUT k_solve_blocks pyramid141 $OUT/pmc_fetch_size.csv $OUT/pmc_write_size.csv > $OUT/pmc_traffic.json
# ... the small-island sample (100 000 piles of 5 boxes), steady state only: tools/gpu_piles_steady.py
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch2 -- python3 tools/gpu_piles_steady.py > $OUT/fetch2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write2 -- python3 tools/gpu_piles_steady.py > $OUT/write2.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_fetch2 last 10 > $OUT/pmc_piles_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write2 last 10 > $OUT/pmc_piles_write_size.csv
python3 tools/pmc_traffic_json.py $OUT k_solve_small piles100000x5 $OUT/pmc_piles_fetch_size.csv $OUT/pmc_piles_write_size.csv > $OUT/piles_pmc_traffic.json
# ... and the bandwidth kernels bench.py quotes for the other configurations, in the windows it times them in:
# config 3 (Tumbler 316 x 316, settled 400 steps): k_collide; config 5 on one GPU (1 M field, 10 000 bullets, from step 30): k_sync_fixtures
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch3 -- python3 tools/gpu_one_scene.py 2 316 0 410 > $OUT/fetch3.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write3 -- python3 tools/gpu_one_scene.py 2 316 0 410 > $OUT/write3.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_fetch3 last 10 > $OUT/pmc_tumbler_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write3 last 10 > $OUT/pmc_tumbler_write_size.csv
python3 tools/pmc_traffic_json.py $OUT k_collide tumbler316 $OUT/pmc_tumbler_fetch_size.csv $OUT/pmc_tumbler_write_size.csv > $OUT/tumbler_pmc_traffic.json
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch4 -- python3 tools/gpu_one_scene.py 3 1000000 10000 40 ccd > $OUT/fetch4.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write4 -- python3 tools/gpu_one_scene.py 3 1000000 10000 40 ccd > $OUT/write4.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_fetch4 last 10 > $OUT/pmc_field_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write4 last 10 > $OUT/pmc_field_write_size.csv
python3 tools/pmc_traffic_json.py $OUT k_sync_fixtures field1000000 $OUT/pmc_field_fetch_size.csv $OUT/pmc_field_write_size.csv > $OUT/field_pmc_traffic.json
# per-step budgets: HelloWorld (the floor), config 3 settled (steps 400-420), config 4's share settled (steps 340-360), config 5 on one GPU
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_hello -- python3 tools/gpu_one_scene.py 0 0 0 400 ccd > $OUT/hello.log 2>&1
python3 tools/trace_steady.py /tmp/prof_hello 100 > $OUT/helloworld_steady_state_per_step.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tumbler -- python3 tools/gpu_one_scene.py 2 316 0 420 > $OUT/tumbler.log 2>&1
python3 tools/trace_steady.py /tmp/prof_tumbler 20 > $OUT/tumbler316_settled_steady_state_per_step.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_p316 -- python3 tools/gpu_one_scene.py 1 316 1 360 ccd > $OUT/p316.log 2>&1
python3 tools/trace_steady.py /tmp/prof_p316 20 > $OUT/pyramid316_settled_steady_state_per_step.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_field -- python3 tools/gpu_one_scene.py 3 1000000 10000 50 ccd > $OUT/field.log 2>&1
python3 tools/trace_steady.py /tmp/prof_field 10 > $OUT/field1m_steady_state_per_step.txt
timeout 100 python3 tools/gpu_floor.py 3000 1 > $OUT/floor.txt 2>&1; timeout 100 python3 tools/gpu_floor.py 3000 0 >> $OUT/floor.txt 2>&1
timeout 300 python3 tools/gpu_lazy_field.py > $OUT/lazy_readback_field1m.txt 2>&1
# round 4: what ONE rank of a spatially sharded world pays (recorded over 4 in-process ranks, rank 0 replayed alone), and
# k_collide on the 1 M field (PMC traffic of the field run above covers it: pmc_field_*.csv list every kernel)
timeout 900 python3 tools/gpu_spatial_share.py 316 4 320 40 > $OUT/spatial_share.txt 2>&1
timeout 600 python3 tools/gpu_spatial_share.py 141 4 245 40 >> $OUT/spatial_share.txt 2>&1
python3 tools/pmc_traffic_json.py $OUT k_collide field1000000_collide $OUT/pmc_field_fetch_size.csv $OUT/pmc_field_write_size.csv > $OUT/field_collide_pmc_traffic.json 2>/dev/null
head -3 $OUT/pmc_fetch_size.csv; head -3 $OUT/pmc_write_size.csv; head -4 $OUT/steady_state_per_step.txt; cat $OUT/pmc_traffic.json $OUT/piles_pmc_traffic.json $OUT/tumbler_pmc_traffic.json $OUT/field_pmc_traffic.json $OUT/floor.txt
python3 tools/print_bench.py $OUT/bench.json
