#!/bin/bash
# round 3, call 1: first hardware run of the register epilogue (gemm_conv.hip rewrite): kernel parity, step parity, per-shape table, bench
O=$GRAFT_REPO_ROOT/gpurun_out/r3c01; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "mfma or gemm or conv3x3 or geglu or register or tile_waves or groupnorm_statistics or split_k" 2>&1 | tee $O/kernel_tests.log | tail -n 15
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; head -n 12 $O/per_shape.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cut -c1-400 $O/bench.json
