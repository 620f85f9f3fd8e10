#!/bin/bash
# round 3, call 4: LayerNorm fold (kernel test, step parity, full depth), config-4 golden at SD-1.5 size, per-shape table, bench
O=$GRAFT_REPO_ROOT/gpurun_out/r3c04; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "layernorm_fold or gemm or geglu or register or groupnorm_statistics or split_k or conv3x3" 2>&1 | tee $O/kernel_tests.log | tail -n 12
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1
timeout 600 python -m pytest tests/test_unet_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -s -k "full_depth or single_pass or passes_vs_oracle" 2>&1 | tee $O/unet_tests.log | grep -E "rel-L2|passed|failed|Error" | tail -n 12
timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -s -k "config4_size" 2>&1 | tee $O/config4.log | grep -E "config 4|worst|passed|failed|Error" | tail -n 8
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; head -n 6 $O/per_shape.txt; grep -A4 "bandwidth-class" $O/per_shape.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cut -c1-200 $O/bench.json
