#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
timeout 100 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/pytest_k.log 2>&1; echo "kernels rc=$?"
timeout 150 python -m pytest tests/test_unet_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "golden or graph_replay" > $O/pytest_u.log 2>&1; echo "unet golden rc=$?"
timeout 150 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $O/prof/s_kernel_trace.csv
tail -n 1 $O/pytest_k.log $O/pytest_u.log
cat $O/bench.log | tail -n 1 | cut -c1-600
ls $O/prof
