#!/bin/bash
# round 6, call 37: backward attention on a 3-stage ring with counted waits (variant _s3): parity, per-shape A/B; contract step on this box with the morning's library
O=$GRAFT_REPO_ROOT/gpurun_out/r6bg; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/pytest_with_lib.py storygen_amd/lib/libstorygen_hip_s3.so tests/test_backward_gpu.py -x -q -m gpu -k "attention or transformer_block or training_step" > $O/tests_s3.log 2>&1; grep -E "passed|failed" $O/tests_s3.log
for v in "" _s3 "" _s3; do
  timeout 300 python tools/bench_attn_bwd.py storygen_amd/lib/libstorygen_hip$v.so > $O/bwd$v.txt 2>&1; head -n 8 $O/bwd$v.txt | tail -n 7
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/base.json 2>$O/err.txt; cut -c100-200 $O/base.json
timeout 300 python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_prev.so --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/prev.json 2>$O/err.txt; cut -c100-200 $O/prev.json
