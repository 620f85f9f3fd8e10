#!/bin/bash
# round 6, call 32: where the training step's time is: per-shape table + kernel stats of the whole step
O=$GRAFT_REPO_ROOT/gpurun_out/r6bb; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/profile_train_step.py > $O/train_per_shape.txt 2>&1; head -n 60 $O/train_per_shape.txt
timeout 300 python tools/bench_attn_bwd.py > $O/bwd_new.txt 2>&1; cat $O/bwd_new.txt
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o train -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 5 --warmup 2 > $O/train_prof.json 2>$O/prof_err.txt
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name '*kernel_stats.csv' | head -n 1); cp $f $O/train_kernel_stats.csv; head -n 40 $O/train_kernel_stats.csv | cut -c1-200
rm -rf $O/prof
