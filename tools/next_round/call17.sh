#!/bin/bash
# first hardware run of the f3 (VAE / CLIP) and f4 (optimizer / training loop) pieces
O=$GRAFT_REPO_ROOT/gpurun_out/r2c17; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests/test_encoders_gpu.py tests/test_optim_gpu.py -q -m gpu --no-header -p no:cacheprovider --maxfail=40 -rP 2>&1 | tee $O/tests.log | grep -v "^$" | tail -n 150
