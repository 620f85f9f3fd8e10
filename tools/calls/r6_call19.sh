#!/bin/bash
# round 6, call 19: out-of-bounds write check of the GEMM plans (guard bands)
O=$GRAFT_REPO_ROOT/gpurun_out/r6s; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp_oob.py > $O/oob.txt 2>&1; tail -30 $O/oob.txt
