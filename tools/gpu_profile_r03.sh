#!/bin/bash
# Round-3 kernel traces (GPU box): steady-state per-step kernel budgets of config 2 (bench scene), config 4's share (Pyramid 316,
# settled 320 steps) and config 3 (Tumbler 316, settled 400 steps). Output: gpurun_out/r03prof/*.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r03prof
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-extras --no-exact-order > $OUT/stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_pyramid141_kernel_stats.csv
python3 tools/trace_steady.py /tmp/prof_stats 80 1 > $OUT/pyramid141_steady_state_per_step.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_p316 -- python3 tools/gpu_one_scene.py 1 316 1 360 ccd > $OUT/p316.log 2>&1
python3 tools/trace_steady.py /tmp/prof_p316 20 1 > $OUT/pyramid316_steady_state_per_step.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tumbler -- python3 tools/gpu_one_scene.py 2 316 0 430 > $OUT/tumbler.log 2>&1
python3 tools/trace_steady.py /tmp/prof_tumbler 20 1 > $OUT/tumbler316_steady_state_per_step.txt
head -45 $OUT/pyramid316_steady_state_per_step.txt
