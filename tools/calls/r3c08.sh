#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3c08; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp_batch_scaling.py 2>&1 | grep -v amdgpu.ids | tee $O/exp_batch_scaling.txt
