#!/bin/bash
# round 6, call 27: the V^T projection of SELECTED transformers hinted onto the latency kernel, everything else off it
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; mkdir -p $O
cd $GRAFT_REPO_ROOT
for sel in "ref:blocks.1" "main:blocks.1" "both:down_blocks.1.attentions.0" "both:up_blocks.2.attentions.2" "ref:down_blocks.1.attentions.0" "main:down_blocks.1.attentions.0"; do
SG_LAT_MASK=0 timeout 600 python tools/exp_determinism.py reps=16 only=one-graph "vt=$sel" > $O/vt.txt 2>&1; echo "vt=$sel: $(grep -c bit-identical $O/vt.txt) of 16"
done
