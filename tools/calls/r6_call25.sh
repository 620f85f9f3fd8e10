#!/bin/bash
# round 6, call 25: unpaired launches free to take the latency kernel: which problem kind is it?  (32x32 latent: the 4x4 level has 48 tokens — ragged tiles)
O=$GRAFT_REPO_ROOT/gpurun_out/r6u; mkdir -p $O
cd $GRAFT_REPO_ROOT
SG_LAT_MASK=62 timeout 600 python tools/exp_determinism.py reps=30 only=one-graph nopairs > $O/a.txt 2>&1; echo "unpaired, columns-are-tokens fold OFF the latency kernel, plain swapped projections ON: $(grep -c bit-identical $O/a.txt) of 30"
SG_LAT_MASK=3 timeout 600 python tools/exp_determinism.py reps=30 only=one-graph nopairs > $O/b.txt 2>&1; echo "unpaired, LayerNorm-folded (rows- and columns-are-tokens) ON, everything else OFF: $(grep -c bit-identical $O/b.txt) of 30"
SG_LAT_MASK=1 timeout 600 python tools/exp_determinism.py reps=30 only=one-graph nopairs > $O/c.txt 2>&1; echo "unpaired, ONLY the columns-are-tokens fold ON: $(grep -c bit-identical $O/c.txt) of 30"
