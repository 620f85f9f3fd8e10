#!/bin/bash
# round 6, call 5: bisect the split+priority difference; parity of the wide (64x128, 6-stage) latency form; chain cost of the 8x8 level under it
O=$GRAFT_REPO_ROOT/gpurun_out/r6e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp_determinism.py > $O/det_default.txt 2>&1; tail -12 $O/det_default.txt
timeout 600 python tools/exp_determinism.py nolat nowide > $O/det_nolat.txt 2>&1; tail -12 $O/det_nolat.txt
timeout 600 python tools/exp_determinism.py nomerge > $O/det_nomerge.txt 2>&1; tail -12 $O/det_nomerge.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "latency_kernel or register_epilogue or gemm or conv3x3" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python tools/bench_chain.py default lat latw lat/s1 lat/s2 latw/s2 latw/s4 latw/s8 > $O/chain.txt 2>&1; cat $O/chain.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/bench_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_$i.json
SG_DEV_OPTIONS=1 SG_LAT_WIDE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/bench_nowide_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_nowide_$i.json
done
