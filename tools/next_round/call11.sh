#!/bin/bash
# full validation of the round: every GPU test, the contract bench line, rocprofv3 kernel stats, PMC traffic
O=$GRAFT_REPO_ROOT/gpurun_out/r2c11; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider 2>&1 | tee $O/gpu_tests.log | tail -n 15
timeout 400 python bench.py > $O/bench.json 2>$O/bench.err; tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $CMD > $O/stats.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/traffic_from_pmc.py $O/f/p_counter_collection.csv $O/w/p_counter_collection.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 2; $(date -u +%F)" > $O/traffic.json; cat $O/traffic.json | head -40
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls -la $O $O/stats 2>/dev/null | head -30
