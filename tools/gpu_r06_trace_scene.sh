#!/bin/bash
# Round 6: kernel budget of the last 20 steps of one harness scene under environment switches.
# usage: tools/gpu_r06_trace_scene.sh <tag> <scene> <p0> <p1> <steps> [ENV=VAL ...]
set -eu
: "${5:?usage: tag scene p0 p1 steps [ENV=VAL ...]}"
cd "${GRAFT_REPO_ROOT:?}"
tag="$1"; scene="$2"; p0="$3"; p1="$4"; steps="$5"; shift 5
for kv in "$@"; do export "$kv"; done
out="gpurun_out/settled/$tag"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 tools/gpu_one_scene.py "$scene" "$p0" "$p1" "$steps" > "$out/run.log" 2>&1 || true
python3 tools/trace_steady.py "$out" 20 > "$out/steady_state_per_step.txt" 2>&1 || true
head -40 "$out/steady_state_per_step.txt"
