#!/bin/bash
# round 6, call 35: training step after: split-K scratch, no-SLP backward attention, batched transposes, in-graph finite flag; 20-step timings
O=$GRAFT_REPO_ROOT/gpurun_out/r6be; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_optim_gpu.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log
for i in 1 2; do
timeout 600 python bench.py --train-step --steps 20 --warmup 3 > $O/train_ws_$i.json 2>$O/err.txt; cut -c100-260 $O/train_ws_$i.json
timeout 600 python bench.py --train-step --steps 20 --warmup 3 --train-no-splitk-workspace > $O/train_nows_$i.json 2>$O/err.txt; cut -c100-260 $O/train_nows_$i.json
done
timeout 600 python bench.py --train-step --steps 20 --warmup 3 --optimizer adamw8bit > $O/train_adamw8bit.json 2>$O/err.txt; cut -c100-260 $O/train_adamw8bit.json
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 5 --warmup 2 > $O/train_prof.json 2>$O/prof_err.txt
cp $(find $O/prof -name '*kernel_stats.csv' | head -n 1) $O/train_kernel_stats.csv; rm -rf $O/prof
