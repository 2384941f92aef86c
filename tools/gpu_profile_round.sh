#!/bin/bash
# End-of-round measurement on the GPU box: parity suite, bench line, rocprofv3 kernel stats, PMC traffic passes (separate),
# steady-state per-step kernel budget. Summaries land in gpurun_out/prof_round/ (copy what is judged into profiles/).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
timeout 900 python3 bench.py --steps 200 --warmup 60 2>$OUT/bench.err | tail -1 > $OUT/bench.json
# kernel trace of the bench command in steady state (the 120 settle steps are part of workload construction)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-extras > $OUT/stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/trace_steady.py /tmp/prof_stats 80 1 > $OUT/steady_state_per_step.txt
# HBM-side traffic, FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM): headline workload ...
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 bench.py --steps 10 --warmup 20 --no-cpu-baseline --no-secondary --no-extras > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 bench.py --steps 10 --warmup 20 --no-cpu-baseline --no-secondary --no-extras > $OUT/write.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_fetch last 10 > $OUT/pmc_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write last 10 > $OUT/pmc_write_size.csv
# ... and the small-island sample (100 000 piles of 5 boxes), steady state only: tools/gpu_piles_steady.py
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch2 -- python3 tools/gpu_piles_steady.py > $OUT/fetch2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write2 -- python3 tools/gpu_piles_steady.py > $OUT/write2.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_fetch2 last 10 > $OUT/pmc_piles_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write2 last 10 > $OUT/pmc_piles_write_size.csv
python3 tools/pmc_traffic_json.py $OUT k_solve_blocks pyramid141 $OUT/pmc_fetch_size.csv $OUT/pmc_write_size.csv > $OUT/pmc_traffic.json
python3 tools/pmc_traffic_json.py $OUT k_solve_small piles100000x5 $OUT/pmc_piles_fetch_size.csv $OUT/pmc_piles_write_size.csv > $OUT/piles_pmc_traffic.json
# per-step budgets of HelloWorld (the floor) and of the Tumbler
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_hello -- python3 tools/gpu_one_scene.py 0 0 0 400 ccd > $OUT/hello.log 2>&1
python3 tools/trace_steady.py /tmp/prof_hello 100 > $OUT/helloworld_steady_state_per_step.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tumbler -- python3 tools/gpu_tumbler100k.py 316 80 > $OUT/tumbler.log 2>&1
python3 tools/trace_steady.py /tmp/prof_tumbler 20 > $OUT/tumbler100k_steady_state_per_step.txt
timeout 100 python3 tools/gpu_floor.py 3000 1 > $OUT/floor.txt 2>&1; timeout 100 python3 tools/gpu_floor.py 3000 0 >> $OUT/floor.txt 2>&1
head -3 $OUT/pmc_fetch_size.csv; head -3 $OUT/pmc_write_size.csv; head -4 $OUT/steady_state_per_step.txt; cat $OUT/pmc_traffic.json $OUT/piles_pmc_traffic.json $OUT/floor.txt
python3 tools/print_bench.py $OUT/bench.json
