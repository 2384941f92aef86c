#!/bin/bash
# Round 6: kernel budget of the settled Tumbler under an environment switch.
# usage: tools/gpu_r06_trace.sh <tag> [ENV=VAL ...]   -> gpurun_out/settled/<tag>/steady_state_per_step.txt
set -eu
: "${1:?tag}"
cd "${GRAFT_REPO_ROOT:?}"
tag="$1"; shift
for kv in "$@"; do export "$kv"; done
out="gpurun_out/settled/$tag"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 tools/gpu_one_scene.py 2 316 0 424 > "$out/run.log" 2>&1 || true
python3 tools/trace_steady.py "$out" 20 > "$out/steady_state_per_step.txt" 2>&1 || true
head -40 "$out/steady_state_per_step.txt"
