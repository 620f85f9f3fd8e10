#!/bin/bash
# round 6, call 36: -fno-slp-vectorize on gemm_conv.hip / attention.hip (the contract step): same-box A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r6bf; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/base_$i.json 2>$O/err.txt; cut -c100-200 $O/base_$i.json
  for v in gcnoslp atnoslp; do
    timeout 300 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_$v.so --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/${v}_$i.json 2>$O/err.txt; cut -c100-200 $O/${v}_$i.json
  done
done
