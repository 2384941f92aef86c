set -u
cd "${GRAFT_REPO_ROOT:?}" || exit 1; export TMPDIR=/tmp; OUT=gpurun_out/r05_u; rm -rf "$OUT"; mkdir -p "$OUT"
for spec in "pyramid316 1 316 1 340 ccd k_solve_blocks" "field1000000 3 1000000 10000 50 ccd k_solve_small"; do
  set -- $spec
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$1_$ctr
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$1_$ctr -- python3 tools/gpu_one_scene.py $2 $3 $4 $5 $6 > /dev/null 2>&1
    python3 tools/pmc_summary.py /tmp/pmc_$1_$ctr last 10 > $OUT/pmc_$1_$(echo $ctr | tr A-Z a-z).csv
  done
  python3 tools/pmc_traffic_json.py $OUT $7 $1 $OUT/pmc_$1_fetch_size.csv $OUT/pmc_$1_write_size.csv > $OUT/$1_solver_pmc_traffic.json
  head -8 $OUT/$1_solver_pmc_traffic.json
done
