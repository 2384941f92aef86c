cd $GRAFT_REPO_ROOT
for i in 1 2; do python3 bench.py --no-cpu-baseline | tail -1 > gpurun_out/b.json; python3 -c "
import json; j=json.load(open('gpurun_out/b.json')); r=j['roofline_small_islands']; print('primary', round(j['ms_per_step'],4), 'small-island kernel us', round(r['mean_launch_us'],1), 'frac', round(r['frac'],4))"; done
