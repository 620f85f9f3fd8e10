#!/bin/bash
# re-run of the four tests fixed after call 17, encoder timings + rocprof table, training step incl. optimizer
O=$GRAFT_REPO_ROOT/gpurun_out/r2c18; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_encoders_gpu.py tests/test_optim_gpu.py -q -m gpu --no-header -p no:cacheprovider --maxfail=40 -rP -k "pipeline_with_hip or adamw8bit_kernel or stage2_trainer" 2>&1 | tee $O/tests.log | grep -v "^$" | tail -n 60
timeout 200 python tools/profile_encoders.py 2>&1 | tail -n 3 | tee $O/encoders.json
for opt in none adamw8bit adamw; do timeout 300 python bench.py --train-step --optimizer $opt --steps 8 --warmup 2 2>&1 | tail -n 1 | tee $O/train_$opt.json | cut -c1-400; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o p -- python $GRAFT_REPO_ROOT/tools/profile_encoders.py > $O/enc.log 2>&1
find $O -name "*kernel_trace.csv" -delete
head -25 $O/enc/p_kernel_stats.csv | cut -c1-200
