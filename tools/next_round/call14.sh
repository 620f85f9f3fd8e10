#!/bin/bash
# final validation of round 2: smoke, every GPU test, the contract bench line, rocprofv3 kernel stats, PMC traffic
O=$GRAFT_REPO_ROOT/gpurun_out/r2c14; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 2
timeout 1500 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider 2>&1 | tee $O/gpu_tests.log | tail -n 8
timeout 400 python bench.py > $O/bench.json 2>$O/bench.err; tail -c 300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $CMD > $O/stats.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/traffic_from_pmc.py $O/f/p_counter_collection.csv $O/w/p_counter_collection.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 2 final; $(date -u +%F)" > $O/traffic.json; head -14 $O/traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
