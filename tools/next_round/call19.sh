#!/bin/bash
# round-2 closing validation after the f3 / f4 additions: the GPU tests touched by them, the contract bench line, PMC traffic of this
# build, rocprofv3 table of the encoders, training step with the optimizer in the timed region
O=$GRAFT_REPO_ROOT/gpurun_out/r2c19; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1
timeout 900 python -m pytest tests/test_encoders_gpu.py tests/test_optim_gpu.py tests/test_dropin_gpu.py tests/test_backward_gpu.py -q -m gpu --no-header -p no:cacheprovider --maxfail=40 -k "not attention_backward and not groupnorm_bwd and not layernorm_bwd" 2>&1 | tee $O/gpu_tests.log | tail -n 6
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cut -c1-330 $O/bench.json
timeout 200 python tools/profile_encoders.py 2>/dev/null | tail -n 1 | tee $O/encoders.json
for opt in none adamw8bit; do timeout 300 python bench.py --train-step --optimizer $opt --steps 8 --warmup 2 2>/dev/null | tail -n 1 > $O/train_$opt.json; cut -c1-200 $O/train_$opt.json; done
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o p -- python $GRAFT_REPO_ROOT/tools/profile_encoders.py > $O/enc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/traffic_from_pmc.py $O/f/p_counter_collection.csv $O/w/p_counter_collection.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; round 2 closing; $(date -u +%F)" > $O/traffic.json; head -4 $O/traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
