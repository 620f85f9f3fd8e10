#!/bin/bash
# round 6, call 26: ONLY the columns-are-tokens fold on the latency kernel (unpaired, lat_mask 1), restricted by level through lat_min_kt / lat_max_kt / lat_tiles
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; mkdir -p $O
cd $GRAFT_REPO_ROOT
SG_LAT_MASK=1 SG_LAT_MAX_KT=12 timeout 600 python tools/exp_determinism.py reps=20 only=one-graph nopairs > $O/a.txt 2>&1; echo "only the C = 640 level (N = 768 tokens): $(grep -c bit-identical $O/a.txt) of 20"
SG_LAT_MASK=1 SG_LAT_MIN_KT=16 timeout 600 python tools/exp_determinism.py reps=20 only=one-graph nopairs > $O/b.txt 2>&1; echo "only the C = 1280 levels (N = 192 and 48): $(grep -c bit-identical $O/b.txt) of 20"
SG_LAT_MASK=1 SG_LAT_MIN_KT=16 SG_LAT_TILES=30 timeout 600 python tools/exp_determinism.py reps=20 only=one-graph nopairs > $O/c.txt 2>&1; echo "only N = 48 (the 4x4 level): $(grep -c bit-identical $O/c.txt) of 20"
