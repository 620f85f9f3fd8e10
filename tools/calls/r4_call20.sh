#!/bin/bash
# round 4, call 20: closing validation of the frozen kernel sources: tile table re-tuned on this build, smoke, whole GPU suite, every bench line,
# rocprofv3 kernel stats, PMC traffic
O=$GRAFT_REPO_ROOT/gpurun_out/r4s; mkdir -p $O
cd $GRAFT_REPO_ROOT
cp storygen_amd/tuning/mi355x_tiles.json $O/tiles_before.json
timeout 400 python tools/tune_tiles.py > $O/tune.log 2>&1; tail -n 1 $O/tune.log
cp storygen_amd/tuning/mi355x_tiles.json $O/tiles_after.json
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --maxfail=20 --durations=12 2>&1 | tee $O/gpu_tests.log | tail -n 25
timeout 600 python bench.py > $O/bench.json 2>$O/bench.err; cut -c1-230 $O/bench.json
timeout 300 python bench.py --no-cpu-baseline --stage auto-regressive > $O/bench_autoregressive.json 2>/dev/null; cut -c1-200 $O/bench_autoregressive.json
timeout 300 python bench.py --no-cpu-baseline --config5-shape > $O/bench_config5_fp16.json 2>/dev/null; cut -c1-200 $O/bench_config5_fp16.json
timeout 300 python bench.py --no-cpu-baseline --config5-shape --fp8-attention > $O/bench_config5_fp8.json 2>/dev/null; cut -c1-200 $O/bench_config5_fp8.json
for opt in none adamw8bit; do timeout 300 python bench.py --train-step --optimizer $opt --steps 8 --warmup 2 2>/dev/null | tail -n 1 > $O/train_$opt.json; cut -c1-200 $O/train_$opt.json; done
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; head -n 4 $O/per_shape.txt
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 4 closing (second); $(date -u +%F)" > $O/traffic.json; head -c 1200 $O/traffic.json
K=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $K $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
