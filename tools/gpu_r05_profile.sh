#!/bin/bash
# Round 5: kernel trace + PMC traffic of the headline workload (config 3, the settled Tumbler) and the N = 2 bench path on one GPU.
# usage: tools/gpu_r05_profile.sh <tag>
set -u
: "${1:?usage: gpu_r05_profile.sh <tag>}"
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
OUT="gpurun_out/$1"
rm -rf "$OUT"; mkdir -p "$OUT"
( export B2_BENCH_SHARE_GPU=1 B2_BENCH_BACKEND=gloo; timeout 900 python3 bench.py --gpus 2 --steps 10 --warmup 5 --tumbler 100 --no-extras --no-secondary > $OUT/bench_n2_shared_gpu_gloo.json 2> $OUT/bench_n2.err; echo "n2 rc=$?" )
tail -3 $OUT/bench_n2.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras --no-long-window > $OUT/stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_tumbler316_kernel_stats.csv
python3 tools/trace_steady.py /tmp/prof_stats 20 > $OUT/tumbler316_settled_steady_state_per_step.txt
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 tools/gpu_one_scene.py 2 316 0 425 > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 tools/gpu_one_scene.py 2 316 0 425 > $OUT/write.log 2>&1
python3 tools/pmc_family_json.py /tmp/prof_fetch /tmp/prof_write tumbler316 20 > $OUT/tumbler316_solver_family_pmc_traffic.json
python3 tools/pmc_summary.py /tmp/prof_fetch last 20 > $OUT/pmc_tumbler_fetch_size.csv
python3 tools/pmc_summary.py /tmp/prof_write last 20 > $OUT/pmc_tumbler_write_size.csv
head -30 $OUT/tumbler316_settled_steady_state_per_step.txt; head -12 $OUT/tumbler316_solver_family_pmc_traffic.json; head -c 600 $OUT/bench_n2_shared_gpu_gloo.json
