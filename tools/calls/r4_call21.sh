#!/bin/bash
# round 4, call 21: after the tile-table fix (a tuned split-K count only with a workspace that holds it) and the last comment edits of the
# kernel sources: whole GPU suite, the contract line, the training-step lines, PMC traffic on the final sources
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --maxfail=20 --durations=8 2>&1 | tee $O/gpu_tests.log | tail -n 14
timeout 600 python bench.py > $O/bench.json 2>$O/bench.err; cut -c1-230 $O/bench.json
for opt in none adamw8bit; do timeout 300 python bench.py --train-step --optimizer $opt --steps 8 --warmup 2 2>$O/train_$opt.err | tail -n 1 > $O/train_$opt.json; cut -c1-300 $O/train_$opt.json; done
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 4, final sources; $(date -u +%F)" > $O/traffic.json; head -c 400 $O/traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
