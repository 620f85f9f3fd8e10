#!/bin/bash
# round 3, call 14: single-pass division-free GroupNorm statistics merge + wider second-pass reduce: kernel tests, then a same-box
# A/B against the previous commit's library (storygen_amd/lib/libstorygen_hip_s1.so, built from 5935e67)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c14; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_kernels_gpu.py -q --no-header -p no:cacheprovider --maxfail=10 -k "statistics or groupnorm" 2>&1 | tee $O/kernel_tests.log | tail -n 8
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline > $O/bench_new_$i.json 2>/dev/null; cut -c100-190 $O/bench_new_$i.json
  timeout 200 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_s1.so --no-cpu-baseline > $O/bench_s1_$i.json 2>/dev/null; cut -c100-190 $O/bench_s1_$i.json
done
