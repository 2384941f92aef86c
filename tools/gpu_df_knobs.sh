#!/bin/bash
# resident large-island solver knobs on the bench workload (pyramid 141 rows, CCD on):
# workgroup size x {pushed mailboxes across XCDs, single-XCD attempt, polled body rows}
for lanes in 256 128 64; do
  for mode in "B2HIP_SOLVER_DEFAULT=1" "B2HIP_SOLVER_SINGLE_XCD=1" "B2HIP_SOLVER_ROWS=1"; do
    echo -n "lanes $lanes $mode: "
    env B2HIP_DF_LANES=$lanes $mode python3 bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 120 2>&1 | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print(round(j['ms_per_step'], 4), 'ms/step; resident solver launch us', round(j['roofline']['mean_launch_us'], 1))"
  done
done
