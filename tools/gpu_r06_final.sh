#!/bin/bash
# Round 6, final set: GPU suite, bench line, the headline workload under rocprofv3 (kernel stats + PMC traffic of the solver
# family), settled-window budgets of configs 3 / 4-share / 5 / 2, the component-wise TOI loops' phase budget, the one-GPU proxy
# of a spatially sharded rank. usage: tools/gpu_r06_final.sh <tag> [notests]
set -u
: "${1:?usage: gpu_r06_final.sh <tag> [notests]}"
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
OUT="gpurun_out/$1"
rm -rf "$OUT"; mkdir -p "$OUT"
if [ "${2:-}" != "notests" ]; then
  timeout 2400 python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1
  grep -a "passed\|failed" $OUT/pytest_gpu.txt | tail -2
  timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
fi
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_tumbler316.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras --no-long-window > $OUT/stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_tumbler316_kernel_stats.csv
python3 tools/trace_steady.py /tmp/prof_stats 20 > $OUT/tumbler316_settled_steady_state_per_step.txt
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 tools/gpu_one_scene.py 2 316 0 725 > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 tools/gpu_one_scene.py 2 316 0 725 > $OUT/write.log 2>&1
python3 tools/pmc_family_json.py /tmp/prof_fetch /tmp/prof_write tumbler316 20 > $OUT/tumbler316_solver_family_pmc_traffic.json
python3 tools/pmc_summary.py /tmp/prof_fetch last 20 > $OUT/pmc_tumbler_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write last 20 > $OUT/pmc_tumbler_write_size.csv
for spec in "field1m 3 1000000 10000 50 ccd" "pyramid316 1 316 1 340 ccd" "pyramid141 1 141 1 260 ccd"; do
  set -- $spec
  rm -rf /tmp/prof_$1; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$1 -- python3 tools/gpu_one_scene.py $2 $3 $4 $5 $6 > $OUT/$1.log 2>&1
  python3 tools/trace_steady.py /tmp/prof_$1 10 1 > $OUT/$1_steady_state_per_step.txt
done
timeout 300 python3 tools/gpu_step_series.py 2 316 0 1200 50 > $OUT/tumbler316_step_series.txt 2>&1
cp gpurun_out/settled_windows_*.txt gpurun_out/solution_quality_pyramid141.txt $OUT/ 2>/dev/null
timeout 300 python3 tools/gpu_toi_domains_probe.py 1000000 10000 40 4 > $OUT/toi_domains_probe.txt 2>&1
(timeout 900 python3 tools/gpu_spatial_share.py 316 4 320 40; timeout 900 python3 tools/gpu_spatial_share.py field 1000000 10000 8 30 20) 2>&1 | grep -v "^E2026\|^W2026" > $OUT/spatial_share.txt
rm -f $OUT/*.log
head -8 $OUT/tumbler316_settled_steady_state_per_step.txt; head -12 $OUT/tumbler316_solver_family_pmc_traffic.json
tail -c 400 $OUT/bench.err
