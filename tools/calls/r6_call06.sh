#!/bin/bash
# round 6, call 6: hunt the rare run-to-run difference: kernel-level stress, then the 5-step loop repeated per configuration
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_kernel_determinism.py 300 > $O/kernel_det.txt 2>&1; cat $O/kernel_det.txt | tail -25
