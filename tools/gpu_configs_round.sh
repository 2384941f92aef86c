#!/bin/bash
# BASELINE.json configurations 3, 4 (one GPU's share) and 5 on one MI355X, plus the multi-island scale cases.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/configs_round.txt
: > $OUT
echo "== config 4, one GPU's share: Pyramid 316 rows (50 086 bodies), CCD on" >> $OUT
timeout 600 python3 tools/gpu_pyr50k.py 316 >> $OUT 2>&1
echo "== config 3: Tumbler 316x316 = 99 856 boxes (CCD off as the reference scene sets it)" >> $OUT
timeout 900 python3 tools/gpu_tumbler100k.py 316 6 10 >> $OUT 2>&1
echo "== config 5: 1 M-body field, no bullets, CCD on" >> $OUT
timeout 900 python3 tools/gpu_field1m.py 1000000 0 12 2>&1 | tail -8 >> $OUT
echo "== config 5: 1 M-body field, 10 000 bullets, CCD on" >> $OUT
timeout 900 python3 tools/gpu_field1m.py 1000000 10000 12 2>&1 | tail -8 >> $OUT
echo "== scale cases (CCD on)" >> $OUT
CCD=1 timeout 900 python3 tools/gpu_scale.py >> $OUT 2>&1
cat $OUT
