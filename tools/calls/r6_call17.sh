#!/bin/bash
# round 6, call 17: single launches of the pair-only problem kinds under real concurrency
O=$GRAFT_REPO_ROOT/gpurun_out/r6q; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp_concurrent_determinism.py 64 > $O/concurrent_det.txt 2>&1; grep -v " 0 of" $O/concurrent_det.txt | tail -40
