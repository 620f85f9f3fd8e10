#!/bin/bash
# round 6, call 23: A/B switches re-measured on the round-6 kernels (one box, alternating with the default)
O=$GRAFT_REPO_ROOT/gpurun_out/r6w; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-loop "$@" > $O/tmp.json 2>>$O/err.txt; python -c "
import json; d=json.load(open('$O/tmp.json')); print('$*'.ljust(28), d['ms_per_step'])" | tee -a $O/ab.txt; }
run
run --ref-ahead 10
run --attn-pair
run
run --no-gemm-pairs
run --no-ff-split
run --no-shared-head
run
run --ref-ahead 2
run --no-splitk-in-gn
