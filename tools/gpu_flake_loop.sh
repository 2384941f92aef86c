#!/bin/bash
# How often one GPU test fails in N fresh processes: tools/gpu_flake_loop.sh <N> <pytest -k expression> [test file]
# (round 5: the bench scene's run-to-run determinism test, 12 runs per library variant, found the asm store hazard)
n=${1:-12}; expr=${2:-run_to_run_determinism}; file=${3:-tests/test_gpu_parity.py}
mkdir -p gpurun_out/flake
f=0
for i in $(seq 1 $n); do
  timeout 300 python3 -m pytest "$file" -x -q -m gpu -k "$expr" > gpurun_out/flake/run_$i.log 2>&1 || f=$((f+1))
done
echo "$expr: $f of $n failed" | tee gpurun_out/flake/summary.txt
