#!/bin/bash
# round 6: the two PMC traffic passes + algorithmic bytes alone (re-stamps profiles/traffic.json after a change of the kernel sources); also the kernel + UNet suites
O=$GRAFT_REPO_ROOT/gpurun_out/r6pmc; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --dump-algorithmic $O/algorithmic.json > $O/bench_first.json 2>$O/bench_contract.err; cut -c100-200 $O/bench_first.json
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-loop"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-loop (ref_ahead 5); algorithmic bytes: bench.py --dump-algorithmic of the same build; MI355X; round 6, FINAL sources and build flags (no SLP vectorisation); $(date -u +%F)" $O/algorithmic.json > $O/traffic.json; head -c 200 $O/traffic.json; echo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/f $O/w
cp $O/traffic.json profiles/traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/err.txt; cut -c1-200 $O/bench.json
