#!/bin/bash
# round 6, call 20: closing measurements on the FINAL kernel sources: smoke, whole GPU suite, contract line (with cpu_baseline, loop_50_steps_ms,
# algorithmic bytes), rocprofv3 kernel stats of the same command, the two PMC traffic passes, per-shape table, chain costs
O=$GRAFT_REPO_ROOT/gpurun_out/r6z; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 1800 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --maxfail=20 2>&1 | tee $O/gpu_tests.log | tail -n 4
timeout 900 python bench.py --steps 20 --warmup 5 --dump-algorithmic $O/algorithmic.json > $O/bench.json 2>$O/bench_contract.err; cut -c1-300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/ks.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-loop"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-loop (ref_ahead 5); algorithmic bytes: bench.py --dump-algorithmic of the same build; MI355X; round 6, FINAL sources (latency-form GEMM kernel, ff.net.2 + proj_out as one GEMM, reference engine on an fp16 stream); $(date -u +%F)" $O/algorithmic.json > $O/traffic.json; head -c 400 $O/traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/ks $O/f $O/w
timeout 900 python tools/profile_step.py --ref-ahead 5 > $O/per_shape.txt 2>&1; head -8 $O/per_shape.txt
timeout 600 python tools/bench_chain.py default > $O/chain.txt 2>&1; tail -8 $O/chain.txt
