#!/bin/bash
# round 6, call 1: contract line of the round-5 sources on this round's box + in-graph cost of the main pass's small GEMMs (tools/bench_chain.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cut -c1-300 $O/bench.json
timeout 600 python tools/bench_chain.py default 128x128 128x64 64x128 64x64 > $O/chain.txt 2>&1; cat $O/chain.txt
