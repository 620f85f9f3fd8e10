#!/bin/bash
# round 3, call 10: tile + explicit split-K tuning, then the step with the new table
O=$GRAFT_REPO_ROOT/gpurun_out/r3c10; mkdir -p $O
cd $GRAFT_REPO_ROOT
cp storygen_amd/tuning/mi355x_tiles.json $O/tiles_before.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_before.json 2>/dev/null; cut -c1-190 $O/bench_before.json
timeout 1500 python tools/tune_tiles.py > $O/tune.log 2>&1; tail -n 2 $O/tune.log
cp storygen_amd/tuning/mi355x_tiles.json $O/mi355x_tiles.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_after.json 2>/dev/null; cut -c1-190 $O/bench_after.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_after2.json 2>/dev/null; cut -c1-190 $O/bench_after2.json
