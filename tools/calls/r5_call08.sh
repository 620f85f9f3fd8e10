#!/bin/bash
# round 5, call 8: closing measurements on the FINAL round-5 kernel sources — whole GPU suite, smoke, the contract line, the non-contract lines
# (auto-regressive, config 5 fp16 / fp8, training step), rocprofv3 kernel stats of the contract command, PMC traffic + algorithmic bytes
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --maxfail=20 --durations=8 2>&1 | tee $O/gpu_tests.log | tail -n 14
timeout 600 python bench.py --steps 20 --warmup 5 --dump-algorithmic $O/algorithmic.json > $O/bench.json 2>$O/bench.err; cut -c1-260 $O/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --stage auto-regressive 2>>$O/bench.err | tail -n 1 > $O/bench_autoregressive.json; cut -c1-200 $O/bench_autoregressive.json
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --config5-shape 2>>$O/bench.err | tail -n 1 > $O/config5_fp16_bench.json; cut -c1-200 $O/config5_fp16_bench.json
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --config5-shape --fp8-attention 2>>$O/bench.err | tail -n 1 > $O/config5_fp8_bench.json; cut -c1-200 $O/config5_fp8_bench.json
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --config5-shape --fp8-attention --ref-ahead 1 2>>$O/bench.err | tail -n 1 > $O/config5_fp8_G1_bench.json; cut -c1-200 $O/config5_fp8_G1_bench.json
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --config5-shape --ref-ahead 1 2>>$O/bench.err | tail -n 1 > $O/config5_fp16_G1_bench.json; cut -c1-200 $O/config5_fp16_G1_bench.json
for opt in none adamw8bit; do timeout 300 python bench.py --train-step --optimizer $opt --steps 8 --warmup 2 2>$O/train_$opt.err | tail -n 1 > $O/train_$opt.json; cut -c1-300 $O/train_$opt.json; done
timeout 300 python tools/profile_step.py --ref-ahead 5 > $O/per_shape.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $O/ks.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 5 --warmup 5 --no-cpu-baseline (ref_ahead 5); algorithmic bytes: bench.py --dump-algorithmic of the same build; MI355X; round 5, FINAL sources (taps innermost, grouped tiles); $(date -u +%F)" $O/algorithmic.json > $O/traffic.json; head -c 600 $O/traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/ks/*/*.csv 2>/dev/null
