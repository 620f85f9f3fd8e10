#!/bin/bash
# round 6, call 16: is it the PAIRED launch or the problem kinds that only occur in pairs (columns-are-tokens, swapped operands)?
# every kind on the latency kernel (lat_mask 63), with the pairs launched as pairs vs as two plain launches each
O=$GRAFT_REPO_ROOT/gpurun_out/r6p; mkdir -p $O
cd $GRAFT_REPO_ROOT
SG_LAT_MASK=63 timeout 600 python tools/exp_determinism.py reps=30 only=one-graph > $O/det_pairs.txt 2>&1; echo "pairs as pairs: $(grep -c bit-identical $O/det_pairs.txt) of 30"
SG_LAT_MASK=63 timeout 600 python tools/exp_determinism.py reps=30 only=one-graph nopairs > $O/det_nopairs.txt 2>&1; echo "pairs as two launches: $(grep -c bit-identical $O/det_nopairs.txt) of 30"
