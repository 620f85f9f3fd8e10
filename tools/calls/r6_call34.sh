#!/bin/bash
# round 6, call 34: training step with / without split-K scratch for the training classes (same box, same library), kernel stats of both
O=$GRAFT_REPO_ROOT/gpurun_out/r6bd; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_optim_gpu.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log
for i in 1 2; do
timeout 600 python bench.py --train-step --steps 5 --warmup 2 > $O/train_ws_$i.json 2>$O/err.txt; cut -c100-260 $O/train_ws_$i.json
timeout 600 python bench.py --train-step --steps 5 --warmup 2 --train-no-splitk-workspace > $O/train_nows_$i.json 2>$O/err.txt; cut -c100-260 $O/train_nows_$i.json
done
export TMPDIR=/tmp; cd /tmp
for v in ws nows; do
  X=""; [ $v = nows ] && X="--train-no-splitk-workspace"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o train -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 5 --warmup 2 $X > $O/train_prof_$v.json 2>$O/prof_err.txt
  cp $(find $O/prof_$v -name '*kernel_stats.csv' | head -n 1) $O/train_kernel_stats_$v.csv; rm -rf $O/prof_$v
done
