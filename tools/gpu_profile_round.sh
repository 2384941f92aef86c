cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
rm -rf gpurun_out/prof_f; mkdir -p gpurun_out/prof_f
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f/stats -- python3 bench.py --steps 100 --warmup 120 --no-cpu-baseline --no-secondary > gpurun_out/prof_f/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_f/fetch -- python3 bench.py --steps 10 --warmup 120 --no-cpu-baseline --no-secondary > gpurun_out/prof_f/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_f/write -- python3 bench.py --steps 10 --warmup 120 --no-cpu-baseline --no-secondary > gpurun_out/prof_f/write.log 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_f/fetch > gpurun_out/prof_f/fetch_summary.csv
python3 tools/pmc_summary.py gpurun_out/prof_f/write > gpurun_out/prof_f/write_summary.csv
find gpurun_out/prof_f -name "*kernel_stats.csv" | head -3
# keep only the small files
find gpurun_out/prof_f -name "*kernel_trace.csv" -delete; find gpurun_out/prof_f -name "*counter_collection.csv" -delete; find gpurun_out/prof_f -name "*.db" -delete
head -5 gpurun_out/prof_f/fetch_summary.csv; head -5 gpurun_out/prof_f/write_summary.csv
