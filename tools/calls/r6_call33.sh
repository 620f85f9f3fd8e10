#!/bin/bash
# round 6, call 33 (second run; the first compared the backward-attention variants _occ2 / _noslp / _occ2noslp): split-K scratch for the training classes, no-SLP backward attention
O=$GRAFT_REPO_ROOT/gpurun_out/r6bc; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_optim_gpu.py -x -q -m gpu 2>&1 | tail -n 3 | tee $O/tests.txt
for v in ""; do
  timeout 300 python tools/bench_attn_bwd.py storygen_amd/lib/libstorygen_hip$v.so > $O/bwd$v.txt 2>&1; head -n 6 $O/bwd$v.txt | tail -n 5
done
timeout 600 python bench.py --train-step --steps 5 --warmup 2 > $O/train_ws.json 2>$O/err2.txt; cut -c1-330 $O/train_ws.json
timeout 600 python tools/profile_train_step.py > $O/train_per_shape.txt 2>&1; head -n 40 $O/train_per_shape.txt
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 5 --warmup 2 > $O/train_prof.json 2>$O/prof_err.txt
cd $GRAFT_REPO_ROOT
cp $(find $O/prof -name '*kernel_stats.csv' | head -n 1) $O/train_kernel_stats.csv; head -n 45 $O/train_kernel_stats.csv | cut -c1-180
rm -rf $O/prof
