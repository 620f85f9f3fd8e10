#!/bin/bash
# round 6, call 9: paired launches on the latency kernel under the kernel-level stress
O=$GRAFT_REPO_ROOT/gpurun_out/r6i; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_kernel_determinism.py 200 > $O/kernel_det.txt 2>&1; grep -E "pair|TOTAL" $O/kernel_det.txt | tail -25
