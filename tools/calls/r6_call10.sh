#!/bin/bash
# round 6, call 10: which buffers differ when the one-graph loop differs (pairs on the latency kernel only)
O=$GRAFT_REPO_ROOT/gpurun_out/r6j; mkdir -p $O
cd $GRAFT_REPO_ROOT
SG_LAT_MASK=1 timeout 900 python tools/exp_determinism.py reps=25 only=one-graph buffers > $O/det_buffers.txt 2>&1; grep -v "bit-identical" $O/det_buffers.txt | cut -c1-1500 | tail -12
