#!/bin/bash
# round 6, call 52: whole GPU suite + smoke on the final tree
O=$GRAFT_REPO_ROOT/gpurun_out/r6gz; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1800 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --maxfail=20 > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log
