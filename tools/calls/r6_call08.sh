#!/bin/bash
# round 6, call 8: which launch kind on the latency kernel makes the one-graph loop differ run to run?  (lat_mask bisect, 30 repeats each)
O=$GRAFT_REPO_ROOT/gpurun_out/r6h; mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 63 62 61 59 55 47 31 1 2 4 8 16 32; do
SG_LAT_MASK=$m timeout 600 python tools/exp_determinism.py reps=30 only=one-graph > $O/det_mask$m.txt 2>&1
echo "lat_mask $m: $(grep -c bit-identical $O/det_mask$m.txt) of 30 bit-identical"
done
