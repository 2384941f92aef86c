#!/bin/bash
# Kernel budget of one step of a settled large scene: rocprofv3 kernel trace over the last steps.
# usage: tools/gpu_settled_trace.sh <tag> <scene 0..> <p0> <p1> <steps> [ccd]   (arguments of tools/gpu_one_scene.py)
set -u
: "${1:?usage: gpu_settled_trace.sh <tag> <scene> <p0> <p1> <steps> [ccd]}"
export TMPDIR=/tmp
tag="$1"; shift
out="${GRAFT_REPO_ROOT:?}/gpurun_out/settled/$tag"
rm -rf "$out"; mkdir -p "$out"
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/gpu_one_scene.py "$@" > $out/run.log 2>&1
python3 tools/trace_steady.py $out 20 > $out/steady_state_per_step.txt 2>&1
head -45 $out/steady_state_per_step.txt
