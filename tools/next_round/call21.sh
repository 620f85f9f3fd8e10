#!/bin/bash
# stage-1 training (attn1 gradients) first hardware run + the stage-2 trainer tests touched by the same change
O=$GRAFT_REPO_ROOT/gpurun_out/r2c21; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_backward_gpu.py tests/test_optim_gpu.py -q -m gpu --no-header -p no:cacheprovider --maxfail=20 -rP -k "training_step or stage" 2>&1 | tee $O/tests.log | grep -v "^$" | tail -n 40
