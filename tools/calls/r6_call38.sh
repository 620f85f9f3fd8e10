#!/bin/bash
# round 6, call 38 (run three times: r6bz; r6cz after the epilogue was generalised for the 128x64-per-wave prototype; r6dz on the library built without SLP vectorisation): closing measurements on the final kernel sources (backward attention / batched transpose changed the source hash): smoke, whole GPU
# suite, contract line (cpu_baseline, loop_50_steps_ms, algorithmic bytes), kernel stats of the same command, the two PMC traffic passes, training lines
O=$GRAFT_REPO_ROOT/gpurun_out/r6dz; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1800 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --maxfail=20 > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --dump-algorithmic $O/algorithmic.json > $O/bench_first.json 2>$O/bench_contract.err; cut -c1-200 $O/bench_first.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/ks.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -6 $O/kernel_stats.csv | cut -c1-160
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-loop"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-loop (ref_ahead 5); algorithmic bytes: bench.py --dump-algorithmic of the same build; MI355X; round 6, FINAL sources and build flags (no SLP vectorisation); $(date -u +%F)" $O/algorithmic.json > $O/traffic.json; head -c 300 $O/traffic.json; echo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/ks $O/f $O/w
cp $O/traffic.json profiles/traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/err.txt; cut -c1-200 $O/bench.json
timeout 900 python bench.py --train-step --steps 20 --warmup 3 > $O/train_none.json 2>>$O/err.txt; cut -c100-260 $O/train_none.json
timeout 900 python bench.py --train-step --steps 20 --warmup 3 --optimizer adamw8bit > $O/train_adamw8bit.json 2>>$O/err.txt; cut -c100-260 $O/train_adamw8bit.json
timeout 600 python tools/profile_train_step.py > $O/train_per_shape.txt 2>&1; head -n 4 $O/train_per_shape.txt
cp profiles/traffic.json $O/traffic_used.json
