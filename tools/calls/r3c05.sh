#!/bin/bash
# round 3, call 5: schedule probe (main / ref alone vs together), tile re-tune for the new epilogue + LayerNorm-fold signatures, bench with the new table
O=$GRAFT_REPO_ROOT/gpurun_out/r3c05; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/probe_schedule.py 2>&1 | tail -n 5 | tee $O/probe_schedule.txt
timeout 900 python tools/tune_tiles.py > $O/tune.log 2>&1; tail -n 3 $O/tune.log
cp storygen_amd/tuning/mi355x_tiles.json $O/mi355x_tiles.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cut -c1-200 $O/bench.json
