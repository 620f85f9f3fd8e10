#!/bin/bash
# round 3, call 15: reduced closing run of the final kernel sources (the whole suite ran on the previous commit in r3c13; the GPU budget
# left does not cover it twice): bench line, rocprofv3 kernel stats, PMC traffic, then the kernel tests and a UNet parity subset
O=$GRAFT_REPO_ROOT/gpurun_out/r3c15; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py > $O/bench.json 2>$O/bench.err; cut -c1-230 $O/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 3 final sources; $(date -u +%F)" > $O/traffic.json; head -c 400 $O/traffic.json
K=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $K $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
timeout 300 python -m pytest tests/test_kernels_gpu.py -q --no-header -p no:cacheprovider --maxfail=10 2>&1 | tee $O/kernel_tests.log | tail -n 6
timeout 400 python -m pytest tests/test_unet_gpu.py -q --no-header -p no:cacheprovider --maxfail=10 -k "single_pass or denoise_steps or paired" 2>&1 | tee $O/unet_tests.log | tail -n 6
