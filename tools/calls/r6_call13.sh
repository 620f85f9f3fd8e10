#!/bin/bash
# round 6, call 13: fp16 residual stream in the REFERENCE engine only (batch 20): full-depth error vs the reference's own 50-step latents, and ms per step
O=$GRAFT_REPO_ROOT/gpurun_out/r6m; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/exp_fp16_stream.py ref > $O/ref_fp16_stream.json 2>$O/err.txt; tail -3 $O/err.txt; python -c "
import json
d=json.load(open('$O/ref_fp16_stream.json'))
for k,v in d.items(): print(k, v)
"
