#!/bin/bash
# round 6, call 3: the latency kernel as the default for the small GEMMs (tile table's GEMM entries dropped, 4-stage ring): step A/B, per-shape table, convolutions under the hint
O=$GRAFT_REPO_ROOT/gpurun_out/r6c; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lat_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_lat_$i.json
SG_DEV_OPTIONS=1 SG_LAT_TILES=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_nolat_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_nolat_$i.json
done
timeout 600 python tools/bench_chain.py default lat > $O/chain.txt 2>&1; tail -9 $O/chain.txt
timeout 900 python tools/profile_step.py --ref-ahead 5 > $O/per_shape.txt 2>&1; head -60 $O/per_shape.txt
