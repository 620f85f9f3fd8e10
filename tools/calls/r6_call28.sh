#!/bin/bash
# round 6, call 28: is the hinted V^T launch itself wrong when a repeat differs?  (recomputed on the 64x64-per-wave kernel inside the graph, mismatches counted on the device)
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; mkdir -p $O
cd $GRAFT_REPO_ROOT
SG_LAT_MASK=0 timeout 900 python tools/exp_determinism.py reps=40 only=one-graph "vt=both:down_blocks.1.attentions.0" check > $O/vt_check.txt 2>&1; grep -E "differ|max" $O/vt_check.txt | tail -40
