#!/bin/bash
# End-of-round measurement on the GPU box: parity suite, bench line, rocprofv3 kernel stats, PMC traffic passes (separate),
# steady-state per-step kernel budget. Summaries land in gpurun_out/prof_round/ (copy what is judged into profiles/).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
timeout 900 python3 bench.py --steps 300 --warmup 120 2>$OUT/bench.err | tail -1 > $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 100 --warmup 120 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/trace_steady.py /tmp/prof_stats 80 1 > $OUT/steady_state_per_step.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 bench.py --steps 10 --warmup 120 --no-cpu-baseline --no-secondary > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 bench.py --steps 10 --warmup 120 --no-cpu-baseline --no-secondary > $OUT/write.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_fetch > $OUT/pmc_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write > $OUT/pmc_write_size.csv
head -3 $OUT/pmc_fetch_size.csv; head -3 $OUT/pmc_write_size.csv; head -4 $OUT/steady_state_per_step.txt
python3 -c "
import json; j=json.load(open('$OUT/bench.json')); print(j['value'], j['ms_per_step']); print(j['roofline']); print(j.get('roofline_small_islands')); print(j.get('cpu_baseline'))"
