#!/bin/bash
# round 6, call 44 (second run: with the V^T images): the columns-are-tokens fold on the 4-stage latency kernel under the two-branch graph: do the INPUTS of a differing launch differ between runs, or only its output?
O=$GRAFT_REPO_ROOT/gpurun_out/r6bm; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_determinism.py only=one-graph reps=40 vt=both:down_blocks.1.attentions.0 hash images > $O/hash_both.txt 2>&1; grep -v "amdgpu.ids" $O/hash_both.txt | grep -v "bit-identical" | cut -c1-1500

