#!/bin/bash
# Instruction-mix counters of the kernels of one harness scene's last steps (one rocprofv3 --pmc pass per counter group;
# no trace domains besides --kernel-trace): tools/gpu_pmc_sq.sh <tag> <scene> <p0> <p1> <steps> [ccd]
set -u
cd "${GRAFT_REPO_ROOT:?}" || exit 1; export TMPDIR=/tmp
tag="$1"; shift
OUT=gpurun_out/pmc_sq_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
g=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  g=$((g+1))
  rm -rf /tmp/pmcsq_$g
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmcsq_$g -- python3 tools/gpu_one_scene.py "$@" > $OUT/run_$g.log 2>&1
  python3 tools/pmc_summary.py /tmp/pmcsq_$g last 10 > $OUT/group_$g.csv 2>>$OUT/run_$g.log
done
head -70 $OUT/group_*.csv
