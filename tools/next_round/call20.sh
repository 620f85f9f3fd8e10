#!/bin/bash
# experiment: fp16 residual stream inside the transformer blocks, full-depth parity + step time
O=$GRAFT_REPO_ROOT/gpurun_out/r2c20; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python tools/exp_fp16_stream.py 2>/dev/null | tail -n 1 | tee $O/fp16_stream.json
