#!/bin/bash
# round 6, call 29: the one V^T launch hinted onto other forms of the 32x32-per-wave kernel: 8-stage ring, 64x128 / 6-stage form
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; mkdir -p $O
cd $GRAFT_REPO_ROOT
SG_LAT_MASK=0 timeout 600 python tools/exp_determinism.py reps=16 only=one-graph "vt=both:down_blocks.1.attentions.0" > $O/v4.txt 2>&1; echo "64x64, 4 stages: $(grep -c bit-identical $O/v4.txt) of 16"
SG_LAT_MASK=0 SG_LAT_STAGES=8 timeout 600 python tools/exp_determinism.py reps=16 only=one-graph "vt=both:down_blocks.1.attentions.0" > $O/v8.txt 2>&1; echo "64x64, 8 stages: $(grep -c bit-identical $O/v8.txt) of 16"
SG_LAT_MASK=0 timeout 600 python tools/exp_determinism.py reps=16 only=one-graph "vt=both:down_blocks.1.attentions.0" wide > $O/vw.txt 2>&1; echo "64x128, 6 stages: $(grep -c bit-identical $O/vw.txt) of 16"
