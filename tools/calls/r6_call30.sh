#!/bin/bash
# round 6, call 30: whole GPU suite once more (new reproducibility test, development hooks in the engine), smoke, contract line
O=$GRAFT_REPO_ROOT/gpurun_out/r6zz; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tee $O/gpu_tests.log | tail -n 4
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/err.txt; cut -c1-260 $O/bench.json
