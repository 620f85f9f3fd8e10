#!/bin/bash
# round 6, call 2: first hardware run of the 32x32-per-wave deep-ring GEMM kernel (mma_lat_kernel): parity tests, in-graph chain cost, contract line A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "latency_kernel or register_epilogue or gemm" > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 600 python tools/bench_chain.py default lat > $O/chain.txt 2>&1; cat $O/chain.txt
SG_LAT_STAGES=4 timeout 600 python tools/bench_chain.py lat > $O/chain_s4.txt 2>&1; tail -18 $O/chain_s4.txt
SG_LAT_STAGES=8 timeout 600 python tools/bench_chain.py lat > $O/chain_s8.txt 2>&1; tail -18 $O/chain_s8.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lat_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_lat_$i.json
SG_DEV_OPTIONS=1 SG_LAT_TILES=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_nolat_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_nolat_$i.json
done
